#!/bin/bash
# rebuild fga_merge.hip with different tile constants on the GPU box and time the seed-merge stage
cd $GRAFT_REPO_ROOT
for cfg in "1024 512 4" "512 256 8" "512 512 6" "2048 512 2" "2048 1024 2" "1024 256 4"; do
  set -- $cfg
  rm -f build/obj/fga_merge.hip.o
  make -C fastga_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DTILE_COST=$1 -DSTAGE_CAP=$2" > /dev/null 2>&1
  echo "== TILE_COST=$1 STAGE_CAP=$2 WGS=$3"
  FGA_MERGE_WGS=$3 python tools/merge_bench.py --reps 2 2>&1 | grep "rep 1"
done
