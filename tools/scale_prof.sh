#!/bin/bash
# tools/scale_prof.sh -- rocprofv3 kernel summary of the trace stage on the 1 M-alignment workload (run via gpurun)
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/scale_prof
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/scale_prof -o sp -- python /root/repo/tools/scale_check.py --mbp 150 --self --repeats 0.3 --pafx < /dev/null > /root/repo/gpurun_out/scale_prof/run.log 2>&1
f=$(find /root/repo/gpurun_out/scale_prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then grep -i "trace_\|Name" "$f" | cut -c1-200; else echo none; tail -5 /root/repo/gpurun_out/scale_prof/run.log; fi
