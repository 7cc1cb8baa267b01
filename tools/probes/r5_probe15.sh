#!/bin/bash
# round 5, GPU call 15: which clock does the latency-bound extension run at?  clock64() cycles of the longest wavefront over the
# kernel's time, and rocm-smi's view of sclk while the bench pair runs
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5p; mkdir -p $o
export TMPDIR=/tmp
rocm-smi --showclocks --showperflevel > $o/smi_idle.log 2>&1
( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $o/smi_run.log 2>&1 &
smi=$!
FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 40 --warmup 3 --no-human-scale --batch 0 --no-cold --no-cpu > $o/b.log 2>&1
wait $smi
grep "ms/step\|kernel_ms" $o/b.log | cut -c1-200
grep "extend profile" $o/b.log | tail -3 | cut -c1-400
cat $o/smi_idle.log | grep -i "clk\|perf" | head -12
sort $o/smi_run.log | uniq -c | sort -rn | head -8
