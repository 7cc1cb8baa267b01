#!/bin/bash
# filter: final order by merge, discovery order from the device; parity, then the phases at 10^6 and 4 x 10^6 records
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 900 python -m pytest tests/test_extend_gpu.py tests/test_end_to_end_gpu.py tests/test_edge_cases_gpu.py tests/test_parts_gpu.py -x -q 2>&1 | tail -2
FGA_FILTER_TIMING=1 timeout 300 python tools/scale_check.py --mbp 150 --self 2>&1 | grep "filter timing\|run 1\|stages" | tail -3 | cut -c1-260
FGA_FILTER_TIMING=1 timeout 400 python tools/config4_check.py --mbp 3000 --div 0.01 2>&1 | grep "filter timing\|session_run\|stages\|digest" | cut -c1-260
