#!/bin/bash
# chain scan with the two hot counters on separate cache lines: parity, the throughput shape, and 3 Gbp per workgroup size
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 600 python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -2
for b in 512 1024; do
  FGA_CHAIN_BLOCK=$b FGA_HOST_TIMING=1 timeout 300 python tools/scale_check.py --mbp 150 --self 2>&1 | grep "chain timing" | tail -1 | cut -c1-60
done
python -c "
import sys; sys.path.insert(0,'.')
from fastga_amd import workload
import tempfile,os
d='/tmp/c4'; os.makedirs(d,exist_ok=True)
" 
for b in 512 1024; do
  echo "3 Gbp, block $b"
  FGA_CHAIN_BLOCK=$b FGA_HOST_TIMING=1 timeout 400 python tools/config4_check.py --mbp 3000 --div 0.01 --no-digest --workdir /tmp/c4 --keep 2>&1 | grep "chain timing\|session_run\|stages" | cut -c1-200
done
