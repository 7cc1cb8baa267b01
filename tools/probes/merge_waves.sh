#!/bin/bash
# resident wavefronts per CU of the wave merge kernel
cd $GRAFT_REPO_ROOT
for w in ${WAVES:-16 20 22 24}; do echo "== FGA_MERGE_WAVES=$w"; FGA_MERGE_WAVES=$w python tools/merge_bench.py --reps 3 2>&1 | grep "rep 2"; done
