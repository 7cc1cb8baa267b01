#!/bin/bash
# round 6, GPU call 2: both episode functions on the new step; A/B: round-5 library, scalar trim (default), vector trim
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r6b; mkdir -p $o
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_extend_gpu.py tests/test_shims_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/t1.log 2>&1; tail -2 $o/t1.log
( timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/t2.log 2>&1; tail -2 $o/t2.log
( FGA_EXTEND_NARROW=1 timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/t3.log 2>&1; tail -2 $o/t3.log
( FGA_LIBRARY=$root/fastga_amd/variants/lib_vtrim.so FGA_EXTEND_NARROW=1 timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_extend_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/t4.log 2>&1; tail -2 $o/t4.log
for v in before default vtrim before default vtrim; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 300 python tools/scale_check.py --mbp 150 --self > $o/self150_$v.log 2>&1
  echo "== $v $(grep 'kernels ms' $o/self150_$v.log | tail -1 | sed 's/.*kernels ms/kernels ms/' | cut -c1-160)"
done
for v in before default; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 12 --warmup 2 --no-human-scale --no-cold --batch 0 --no-cpu > $o/b_$v.log 2>&1
  echo "== $v: $(grep 'ms/step' $o/b_$v.log | cut -c1-60) kernel: $(grep 'extend profile' $o/b_$v.log | sed 's/.*kernel \([0-9.]*\) ms.*/\1/' | sort -n | head -6 | tr '\n' ' ')"
done
