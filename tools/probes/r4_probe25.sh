#!/bin/bash
# the genome's bases cross PCIe once (index build keeps the image, fga_dgenome_adopt): 3 Gbp open first (fresh box), then parity
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
FGA_TIMING=1 timeout 400 python tools/config4_check.py --mbp 3000 --div 0.01 2>&1 | grep "upload +\|session_run\|lines_md5\|hipMalloc 38\|index build: uploads" | cut -c1-200
timeout 900 python -m pytest tests/test_gix_device_gpu.py tests/test_end_to_end_gpu.py tests/test_parts_gpu.py tests/test_mask_files_gpu.py -x -q 2>&1 | tail -2
