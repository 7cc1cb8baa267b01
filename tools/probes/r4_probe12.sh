#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for n in 48600000 550000000; do
  for w in 8 16; do
    echo "wide $w"; FGA_SORT_WIDE=$w timeout 60 fastga_amd/bin/sort_bench $n 53 12 uniform 3; FGA_SORT_WIDE=$w timeout 60 fastga_amd/bin/sort_bench $n 53 12 seeds 3
  done
done 2>&1 | cut -c1-150
FGA_SORT_WIDE=16 timeout 60 fastga_amd/bin/sort_bench 100000 61 0 uniform 2 | cut -c1-150
FGA_SORT_WIDE=16 timeout 60 fastga_amd/bin/sort_bench 1000003 128 0 uniform 2 | cut -c1-150
( FGA_SORT_WIDE=16 timeout 900 python -m pytest tests/test_seed_sort_gpu.py tests/test_shims_gpu.py tests/test_end_to_end_gpu.py tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -3 )
