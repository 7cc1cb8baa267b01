#!/bin/bash
# round 5, GPU call 12: waves of 61 .. 124 diagonals in two register blocks (EXT_W2): parity on the toy units through ext_mid,
# end-to-end tests with the throughput build forced, then A/B on the 150 Mbp self shape
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5l; mkdir -p $o
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_extend_gpu.py -x -q -m gpu 2>&1 | tail -12 ) > $o/t1.log 2>&1; tail -5 $o/t1.log
( FGA_EXTEND_NARROW=1 timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -12 ) > $o/t2.log 2>&1; tail -5 $o/t2.log
for v in now2 default; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 300 python tools/scale_check.py --mbp 150 --self > $o/self150_$v.log 2>&1
  echo "== $v"; grep "run 1\|kernels ms" $o/self150_$v.log | tail -2
done
FGA_LIBRARY=$root/fastga_amd/variants/lib_w2prof.so FGA_EXTEND_PROFILE=1 timeout 300 python tools/scale_check.py --mbp 150 --self > $o/mode150.log 2>&1
grep -i "extend modes" $o/mode150.log | tail -6
