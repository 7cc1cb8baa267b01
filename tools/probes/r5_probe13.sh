#!/bin/bash
# round 5, GPU call 13: the latency build (ext_full) with the episode functions (EXT_W2) and / or the lazy trace-point
# arithmetic (EXT_LAZY_CROSS): parity on the toy units and end to end, then the bench pair
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5m; mkdir -p $o
export TMPDIR=/tmp
for v in default fw2lc fw2 flc; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  echo "== $v"
  ( FGA_LIBRARY=$lib timeout 600 python -m pytest tests/test_extend_gpu.py tests/test_end_to_end_gpu.py -x -q -m gpu 2>&1 | tail -4 ) > $o/t_$v.log 2>&1; tail -1 $o/t_$v.log
  FGA_LIBRARY=$lib timeout 300 python tools/bench_brief.py --steps 10 --warmup 3 --no-human-scale --batch 0 --no-cold > $o/b_$v.log 2>&1
  grep "ms/step\|kernel_ms\|cpu" $o/b_$v.log | cut -c1-250
done
