#!/bin/bash
# rebuild fga_merge.hip with per-phase cycle accounting on the GPU box and run the merge stage
cd $GRAFT_REPO_ROOT
rm -f build/obj/fga_merge.hip.o
make -C fastga_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DMERGE_PROF $1" > /dev/null 2>&1
python tools/merge_bench.py --reps 3 2>&1 | tail -6
