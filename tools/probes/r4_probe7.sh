#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4g; mkdir -p $o
for n in 48600000 550000000; do
  for c in 1 4 8 32; do
    echo "chunk $c"; FGA_SORT_CHUNK=$c timeout 60 fastga_amd/bin/sort_bench $n 53 12 uniform 3; FGA_SORT_CHUNK=$c timeout 60 fastga_amd/bin/sort_bench $n 53 12 seeds 3
  done
done > $o/sort.log 2>&1
timeout 60 fastga_amd/bin/sort_bench 100000 61 0 uniform 2 >> $o/sort.log 2>&1
timeout 60 fastga_amd/bin/sort_bench 1000003 128 0 uniform 2 >> $o/sort.log 2>&1
cut -c1-175 $o/sort.log
( timeout 900 python -m pytest tests/test_seed_sort_gpu.py tests/test_shims_gpu.py tests/test_end_to_end_gpu.py tests/test_chain_gpu.py tests/test_gix_device_gpu.py -x -q -m gpu 2>&1 | tail -3 )
