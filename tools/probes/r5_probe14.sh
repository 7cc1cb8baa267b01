#!/bin/bash
# round 5, GPU call 14: the episode functions in the latency build too (spare cells of level 0 fixed): register path against
# the forced ring at 100 Mbp, the suites that pin the extension, A/B on the bench pair
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5o; mkdir -p $o
export TMPDIR=/tmp
timeout 300 python tools/extend_diff.py 100 > $o/diff.log 2>&1; tail -1 $o/diff.log
( timeout 900 python -m pytest tests/test_extend_gpu.py tests/test_end_to_end_gpu.py tests/test_shims_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -4 ) > $o/t1.log 2>&1; tail -1 $o/t1.log
( timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "not 3000 and not 3gbp" 2>&1 | tail -4 ) > $o/t2.log 2>&1; tail -1 $o/t2.log
for v in default now2full default now2full; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  echo "== $v"
  FGA_LIBRARY=$lib timeout 300 python tools/bench_brief.py --steps 10 --warmup 3 --no-human-scale --batch 0 --no-cold > $o/b_$v.log 2>&1
  grep "ms/step\|kernel_ms\|cpu" $o/b_$v.log | cut -c1-250
done
