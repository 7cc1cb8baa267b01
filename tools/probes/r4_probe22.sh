#!/bin/bash
# chain_segment_kernel: persistent wavefronts per CU (bench pair: its units are contig-long)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for w in 8 16 24 32; do
  FGA_CHAIN_SEG_WAVES=$w FGA_HOST_TIMING=1 timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu --no-cold --no-human-scale 2>&1 | grep "chain timing" | tail -1 | cut -c1-60 | sed "s/^/waves per CU $w: /"
done
