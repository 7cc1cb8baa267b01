#!/bin/bash
# round 4, GPU call 1: health of the tree (quick GPU tests), the sort's look-back batch, wave-width histograms
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4a; mkdir -p $o
( timeout 1200 python -m pytest tests/test_parts_gpu.py tests/test_end_to_end_gpu.py tests/test_gix_device_gpu.py tests/test_seed_sort_gpu.py tests/test_shims_gpu.py -x -q -m gpu 2>&1 | tail -15 ) > $o/tests.log 2>&1
for n in 48600000 550000000; do
  for lb in 1 4 8 16; do
    FGA_SORT_LB=$lb timeout 120 fastga_amd/bin/sort_bench $n 53 12 uniform 3
    FGA_SORT_LB=$lb timeout 120 fastga_amd/bin/sort_bench $n 53 12 seeds 3
  done
done > $o/sort.log 2>&1
SORT_BENCH_COPY=1 timeout 120 fastga_amd/bin/sort_bench 550000000 53 12 uniform 1 >> $o/sort.log 2>&1
FGA_LIBRARY=$root/fastga_amd/variants/lib_widthhist.so timeout 300 python tools/scale_check.py --mbp 150 --self > $o/width150.log 2>&1
FGA_LIBRARY=$root/fastga_amd/variants/lib_widthhist.so FGA_TIMING=1 timeout 600 python tools/config4_check.py --mbp 3000 --div 0.01 > $o/width_c4.log 2>&1
tail -5 $o/tests.log; cat $o/sort.log; grep -i "width\|run 1\|stages" $o/width150.log | tail -8; grep -i "width\|session_run\|stages\|upload" $o/width_c4.log | tail -12
