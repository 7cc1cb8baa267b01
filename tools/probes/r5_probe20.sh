#!/bin/bash
# round 5, GPU call 20: seed merge, emission: a run without dropped members (mask byte / strand) indexes its partner directly:
# parity of the merge suites, then the 1 Gbp self comparison with -M (configs[2]) and the bench pair before / after
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5x; mkdir -p $o
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_seed_merge_gpu.py tests/test_end_to_end_gpu.py tests/test_mask_files_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -4 ) > $o/t1.log 2>&1; tail -1 $o/t1.log
for v in before default; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  echo "== $v"
  FGA_LIBRARY=$lib timeout 400 python tools/config3_check.py --mbp 1000 > $o/c3_$v.log 2>&1; grep "run 1\|kernels ms\|merge " $o/c3_$v.log | tail -3 | cut -c1-260
  FGA_LIBRARY=$lib timeout 300 python tools/bench_brief.py --steps 10 --warmup 3 --no-human-scale --no-cold --batch 0 --no-cpu > $o/b_$v.log 2>&1; grep "ms/step\|kernel_ms" $o/b_$v.log | cut -c1-200
done
