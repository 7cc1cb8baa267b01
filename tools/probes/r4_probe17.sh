#!/bin/bash
# chain_small_kernel / chain_sparse_kernel: parity, then the stage's kernel time on the throughput shape and the bench pair
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 600 python -m pytest tests/test_chain_gpu.py tests/test_end_to_end_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -3
FGA_HOST_TIMING=1 timeout 300 python tools/scale_check.py --mbp 150 --self 2>&1 | grep "chain timing" | tail -2 | cut -c1-100
FGA_HOST_TIMING=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu --no-cold --no-human-scale 2>&1 | grep "chain timing\|ms_per_step" | tail -3 | cut -c1-300
