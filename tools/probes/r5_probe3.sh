#!/bin/bash
# round 5, GPU call 3: the throughput build's register step with the trace-point arithmetic / the trim tables only in the
# steps that need them: parity (extension + end-to-end tests) and A/B timing on the 150 Mbp self shape
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5c; mkdir -p $o
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_extend_gpu.py tests/test_end_to_end_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -8 ) > $o/tests.log 2>&1
tail -4 $o/tests.log
for v in nolazy lazycross lazytrim default; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 300 python tools/scale_check.py --mbp 150 --self > $o/self150_$v.log 2>&1
  echo "== $v"; grep "run 1\|kernels ms" $o/self150_$v.log | tail -2
done
FGA_TIMING=1 timeout 600 python tools/config4_check.py --mbp 3000 --div 0.01 > $o/c4_default.log 2>&1
grep -i "extend\|digest\|stages\|session_run" $o/c4_default.log | tail -12
