#!/bin/bash
# tools/merge_knockout.sh -- time the seed-merge stage with phases of the wave kernel knocked out (timing experiment,
# output of the knocked-out builds is wrong by construction).  Variants are built beforehand into fastga_amd/variants/
# with -DKNOCK_AFTER_LOAD / _AFTER_COMPACT / _AFTER_MATCH / KNOCK_STORE (see fga_merge.hip).
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== full"; timeout 120 python tools/merge_bench.py --reps 3 < /dev/null 2>&1 | grep "rep 2"
for v in fastga_amd/variants/lib_*.so; do
  echo "== $v"; FGA_LIBRARY=$PWD/$v timeout 120 python tools/merge_bench.py --reps 3 < /dev/null 2>&1 | tail -1
done
