#!/bin/bash
# round 5, GPU call 21: throughput build, lane masks of uniform origin made on the scalar side (inverse ballot): parity, A/B
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5y; mkdir -p $o
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_extend_gpu.py -x -q -m gpu 2>&1 | tail -3 ) > $o/t1.log 2>&1; tail -1 $o/t1.log
( FGA_EXTEND_NARROW=1 timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -3 ) > $o/t2.log 2>&1; tail -1 $o/t2.log
for v in before default before default; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 300 python tools/scale_check.py --mbp 150 --self > $o/self150_$v.log 2>&1
  echo "== $v $(grep 'kernels ms' $o/self150_$v.log | tail -1 | sed 's/.*kernels ms/kernels ms/' | cut -c1-120)"
done
