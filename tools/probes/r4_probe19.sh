#!/bin/bash
# host tails after the last kernel: parity, then 3 Gbp with the stage clocks (the run first: a fresh box allocates fast)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
FGA_FILTER_TIMING=1 timeout 400 python tools/config4_check.py --mbp 3000 --div 0.01 2>&1 | grep "filter timing\|session_run\|stages\|digest\|upload" | cut -c1-260
timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_parts_gpu.py -x -q 2>&1 | tail -2
