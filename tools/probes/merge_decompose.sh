#!/bin/bash
# tools/merge_decompose.sh -- where the seed merge's time goes on the 100 Mbp bench pair: builds of
# fga_merge.hip that stop after the tile fill (-DMERGE_FILL_ONLY), after the match (-DMERGE_NO_EMIT), before the seed stores
# (-DMERGE_NO_STORE), and the per-phase cycle accounting + wavefront timeline (-DMERGE_PROF), each at several occupancies.
# Run here: builds the variants, then ONE gpurun call.
set -e
cd "$(dirname "$0")/.."
bash tools/build_variant.sh mfill fga_merge.hip -DMERGE_FILL_ONLY
bash tools/build_variant.sh mnoemit fga_merge.hip -DMERGE_NO_EMIT
bash tools/build_variant.sh mnostore fga_merge.hip -DMERGE_NO_STORE
bash tools/build_variant.sh mprof fga_merge.hip -DMERGE_PROF
gpurun --timeout 600 -- 'mkdir -p gpurun_out/merge_decompose; (
for w in 2 4 6 8 12; do echo "whole kernel, $w wavefronts per CU"; FGA_MERGE_WAVES=$w timeout 100 python tools/merge_bench.py --mbp 100 --reps 3 2>&1 | tail -1; done
for v in mfill mnoemit mnostore; do for w in 4 8 12; do echo "$v, $w wavefronts per CU"; FGA_MERGE_WAVES=$w FGA_LIBRARY=fastga_amd/variants/lib_$v.so timeout 100 python tools/merge_bench.py --mbp 100 --reps 3 2>&1 | tail -1; done; done
echo "phase accounting (the instrumentation itself costs ~20 %)"; FGA_LIBRARY=fastga_amd/variants/lib_mprof.so timeout 100 python tools/merge_bench.py --mbp 100 --reps 2 2>&1 | tail -6
) > gpurun_out/merge_decompose/log.txt 2>&1; cat gpurun_out/merge_decompose/log.txt'
