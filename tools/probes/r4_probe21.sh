#!/bin/bash
# unit scan in three launches: parity, the stage clock, and the throughput shape's kernel table again
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_end_to_end_gpu.py tests/test_parts_gpu.py -x -q 2>&1 | tail -2
out=$root/gpurun_out; mkdir -p $out/prof_r04
cd /tmp && export TMPDIR=/tmp
FGA_HOST_TIMING=1 timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_r04/thr -o kt --output-format csv -- python $root/tools/scale_check.py --mbp 150 --self --repeats 0.30 > $out/prof_r04/thr.log 2>&1
cp $out/prof_r04/thr/kt_kernel_stats.csv $out/r04_throughput_kernel_stats.csv
grep "chain timing\|run 1" $out/prof_r04/thr.log | tail -2 | cut -c1-200
grep "unit_\|chain_small" $out/r04_throughput_kernel_stats.csv | cut -c1-120
