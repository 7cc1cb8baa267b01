#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
( timeout 900 python -m pytest tests/test_gix_device_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py tests/test_mask_files_gpu.py "tests/test_full_size_gpu.py::test_config2_100mbp_pair_is_identical_to_the_reference" -x -q -m gpu 2>&1 | tail -4 )
FGA_TIMING=1 timeout 300 python tools/gix_scan_probe.py --mbp 1000 2>&1 | grep -v "hipMalloc\|hipFree" | tail -12
