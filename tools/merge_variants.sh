#!/bin/bash
# A/B variants of the wave merge kernel on the GPU box: each item is a string of extra -D flags
cd $GRAFT_REPO_ROOT
IFS='|' read -ra V <<< "${VARIANTS:-}"
for v in "${V[@]}"; do
  rm -f build/obj/fga_merge.hip.o
  make -C fastga_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $v" > /dev/null 2>&1
  echo "== $v"
  python tools/merge_bench.py --reps 3 2>&1 | grep "rep 2"
done
