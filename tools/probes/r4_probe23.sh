#!/bin/bash
# extension: unit work order on the device; parity, then extend_s against its kernel at 10^6 and 2 x 2 x 10^6 units
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 900 python -m pytest tests/test_extend_gpu.py tests/test_end_to_end_gpu.py tests/test_shims_gpu.py -x -q 2>&1 | tail -2
timeout 300 python tools/scale_check.py --mbp 150 --self 2>&1 | grep "run 1\|stages" | tail -2 | cut -c1-260
timeout 400 python tools/config4_check.py --mbp 3000 --div 0.01 2>&1 | grep "session_run\|stages\|lines_md5" | cut -c1-300
