#!/bin/bash
# round 4, GPU call 4: where the host time of a 3 Gbp comparison goes
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r4d; mkdir -p $o
FGA_TIMING=1 FGA_HOST_TIMING=1 FGA_FILTER_TIMING=1 FGA_EXTEND_PROFILE=1 timeout 600 python tools/config4_check.py --mbp 3000 --div 0.01 > $o/c4.log 2>&1
grep -v "hipFree\|hipMalloc" $o/c4.log | grep -i "timing\|profile\|session_run\|stages\|upload" | cut -c1-400
