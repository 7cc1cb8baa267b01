#!/bin/bash
# chain scan with units ordered and hits re-laid on the device: parity, the throughput shape, the bench pair, 3 Gbp
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_end_to_end_gpu.py tests/test_edge_cases_gpu.py tests/test_parts_gpu.py -x -q 2>&1 | tail -2
FGA_HOST_TIMING=1 timeout 300 python tools/scale_check.py --mbp 150 --self 2>&1 | grep "chain timing\|run 1\|stages" | tail -3 | cut -c1-230
FGA_HOST_TIMING=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-cold --no-human-scale 2>&1 | grep "chain timing" | tail -1 | cut -c1-200
FGA_HOST_TIMING=1 timeout 400 python tools/config4_check.py --mbp 3000 --div 0.01 --no-digest 2>&1 | grep "chain timing\|session_run\|stages" | cut -c1-230
