#!/bin/bash
# tools/trace_prof.sh -- trace stage: parity tests, timing beside the CPU oracle, rocprofv3 kernel summary (run via gpurun)
cd /root/repo
timeout 300 python -m pytest tests/test_trace_gpu.py -x -q < /dev/null 2>&1 | tail -5
timeout 300 python tools/trace_bench.py < /dev/null 2>&1 | tail -8
mkdir -p gpurun_out/trace_prof
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/trace_prof -o tr -- \
    python /root/repo/tools/trace_bench.py --cpu-sample 0 --reps 2 < /dev/null > /root/repo/gpurun_out/trace_prof/run.log 2>&1
cd /root/repo
f=$(find gpurun_out/trace_prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -14 "$f"; else echo "no kernel_stats.csv"; ls -R gpurun_out/trace_prof | head; fi
