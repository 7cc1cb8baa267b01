#!/bin/bash
# rebuild fga_merge.hip with different tile constants / occupancy targets on the GPU box and time the seed-merge stage
# each config: "TILE_COST STAGE_CAP WGS_PER_CU WAVES_PER_EU(0 = compiler's choice)"
cd $GRAFT_REPO_ROOT
for cfg in ${SWEEP:-"1024 512 4 0"}; do
  IFS=: read t s w e nt <<< "$cfg"
  nt=${nt:-256}
  rm -f build/obj/fga_merge.hip.o
  extra=""
  if [ "$e" != "0" ]; then extra="-DMERGE_WAVES_PER_EU=$e"; fi
  make -C fastga_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DTILE_COST=$t -DSTAGE_CAP=$s -DNT=$nt $extra" > /dev/null 2>&1
  echo "== TILE_COST=$t STAGE_CAP=$s WGS=$w WAVES_PER_EU=$e NT=$nt"
  FGA_MERGE_WGS=$w python tools/merge_bench.py --reps 3 2>&1 | grep "rep 2"
done
