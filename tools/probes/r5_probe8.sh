#!/bin/bash
# round 5, GPU call 8: the 3 Gbp comparison's timeline, cold and warm (FGA_TIMING=1), after the seed buffer / forward view reorder
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5h; mkdir -p $o
export TMPDIR=/tmp
FGA_TIMING=1 timeout 900 python tools/config4_check.py --mbp 3000 --div 0.01 --runs 2 --no-digest > $o/c4.log 2>&1
grep -v "pool\|region" $o/c4.log | grep -i "timing\|comparison\|session_run\|stages\|upload" | tail -80
