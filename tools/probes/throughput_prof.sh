#!/bin/bash
# tools/throughput_prof.sh -- rocprofv3 kernel summary of the throughput regime (150 Mbp repeat-heavy unmasked self comparison,
# 10^6 units; run via gpurun): gpurun_out/r02_throughput_kernel_stats.csv, to be copied into profiles/
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $root/gpurun_out/thr_prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/thr_prof -o tp -- python $root/tools/scale_check.py --mbp 150 --self < /dev/null > $root/gpurun_out/thr_prof/run.log 2>&1
f=$(find $root/gpurun_out/thr_prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp "$f" $root/gpurun_out/r02_throughput_kernel_stats.csv; head -8 "$f" | cut -c1-150; else echo none; fi
grep -E "run 1|stages" $root/gpurun_out/thr_prof/run.log | tail -2 | cut -c1-330
