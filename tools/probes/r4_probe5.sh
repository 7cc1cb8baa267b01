#!/bin/bash
# round 4, GPU call 5: wide-tile LDS-staged sort passes
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4e; mkdir -p $o
for n in 48600000 550000000; do
  for w in 0 8 16; do
    FGA_SORT_WIDE=$w timeout 120 fastga_amd/bin/sort_bench $n 53 12 uniform 3
    FGA_SORT_WIDE=$w timeout 120 fastga_amd/bin/sort_bench $n 53 12 seeds 3
  done
done > $o/sort.log 2>&1
FGA_SORT_WIDE=8 timeout 120 fastga_amd/bin/sort_bench 100000 61 0 uniform 2 >> $o/sort.log 2>&1
FGA_SORT_WIDE=16 timeout 120 fastga_amd/bin/sort_bench 1000003 128 0 uniform 2 >> $o/sort.log 2>&1
cut -c1-200 $o/sort.log
for w in 8 16; do
( FGA_SORT_WIDE=$w timeout 900 python -m pytest tests/test_seed_sort_gpu.py tests/test_shims_gpu.py tests/test_end_to_end_gpu.py tests/test_chain_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/tests_$w.log 2>&1
tail -3 $o/tests_$w.log
done
