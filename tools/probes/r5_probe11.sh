#!/bin/bash
# round 5, GPU call 11: fga_run_multi at 3 Gbp with two and four virtual ranks on the one GPU, digest against the golden
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5k; mkdir -p $o
export TMPDIR=/tmp
FGA_TIMING=1 timeout 900 python tools/multi_3g_check.py --devices 0,0 > $o/m2.log 2>&1
grep -v "pool\|region\|hipMalloc\|index build\|forward view" $o/m2.log | tail -16
timeout 900 python tools/multi_3g_check.py --devices 0,0,0,0 > $o/m4.log 2>&1
tail -4 $o/m4.log
