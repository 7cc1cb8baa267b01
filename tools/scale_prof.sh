#!/bin/bash
# tools/scale_prof.sh <tag> -- rocprofv3 --kernel-trace --stats of the other shapes (run ON THE GPU BOX through gpurun):
#   gpurun_out/<tag>_config4_kernel_stats.csv     3 Gbp x 3 Gbp, 1 % (BASELINE configs[3]) on one GPU: tools/config4_check.py
#   gpurun_out/<tag>_config3_kernel_stats.csv     1 Gbp repeat-heavy self comparison with -M (configs[2]): tools/config3_check.py
#   gpurun_out/<tag>_throughput_kernel_stats.csv  150 Mbp repeat-heavy self comparison, 10^6 units: tools/scale_check.py
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $out/prof_$tag/c4 -o kt --output-format csv -- python $root/tools/config4_check.py --mbp 3000 --div 0.01 > $out/prof_$tag/c4.log 2>&1
cp $out/prof_$tag/c4/kt_kernel_stats.csv $out/${tag}_config4_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag/c3 -o kt --output-format csv -- python $root/tools/config3_check.py --mbp 1000 > $out/prof_$tag/c3.log 2>&1
cp $out/prof_$tag/c3/kt_kernel_stats.csv $out/${tag}_config3_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag/thr -o kt --output-format csv -- python $root/tools/scale_check.py --mbp 150 --self --repeats 0.30 > $out/prof_$tag/thr.log 2>&1
cp $out/prof_$tag/thr/kt_kernel_stats.csv $out/${tag}_throughput_kernel_stats.csv
grep -h "fga_session_run\|stages\|run 1" $out/prof_$tag/c4.log $out/prof_$tag/c3.log $out/prof_$tag/thr.log
for f in config4 config3 throughput; do echo "== $f"; head -14 $out/${tag}_${f}_kernel_stats.csv | cut -c1-150; done
