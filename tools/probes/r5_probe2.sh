#!/bin/bash
# round 5, GPU call 2: where the throughput-regime extension spends its step cycles (register vs ring steps, by width)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5b; mkdir -p $o
export TMPDIR=/tmp
FGA_LIBRARY=$root/fastga_amd/variants/lib_modeprof.so FGA_EXTEND_PROFILE=1 timeout 300 python tools/scale_check.py --mbp 150 --self > $o/mode150.log 2>&1
grep -i "extend modes\|extend profile\|run 1\|stages" $o/mode150.log | tail -24
FGA_LIBRARY=$root/fastga_amd/variants/lib_modeprof.so FGA_EXTEND_PROFILE=1 timeout 600 python tools/config4_check.py --mbp 3000 --div 0.01 > $o/mode_c4.log 2>&1
grep -i "extend modes\|extend profile" $o/mode_c4.log | tail -24
