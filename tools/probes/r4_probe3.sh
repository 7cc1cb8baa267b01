#!/bin/bash
# round 4, GPU call 3: the bench line with the new legs; kernel trace of a 3 Gbp open + comparison
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4c; mkdir -p $o
timeout 900 python bench.py > $o/bench.json 2> $o/bench.err
tail -c 6000 $o/bench.json; tail -5 $o/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_c4 -o c4 -- python $root/tools/config4_check.py --mbp 3000 --div 0.01 > $o/prof_c4.log 2>&1
f=$(find $o/prof_c4 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $o/c4_kernel_stats.csv && head -30 "$f" | cut -c1-200
find $o/prof_c4 -name '*kernel_trace.csv' -size +60M -delete
