#!/bin/bash
# round 5, GPU call 18: the forward unwind window by window (one v_readlane per link, pairs by the lanes): parity, then the bench pair
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5s; mkdir -p $o
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_extend_gpu.py tests/test_end_to_end_gpu.py tests/test_shims_gpu.py tests/test_golden_gpu.py tests/test_edge_cases_gpu.py -x -q -m gpu 2>&1 | tail -6 ) > $o/t1.log 2>&1; tail -2 $o/t1.log
FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 20 --warmup 3 --no-human-scale --no-cold --batch 0 > $o/b.log 2>&1
grep "ms/step\|kernel_ms\|cpu" $o/b.log | cut -c1-200
grep "extend profile" $o/b.log | tail -3 | cut -c1-260
grep "extend profile" $o/b.log | sed 's/.*kernel \([0-9.]*\) ms.*/\1/' | sort -n | tr '\n' ' '; echo
timeout 300 python tools/scale_check.py --mbp 150 --self > $o/self150.log 2>&1; grep "run 1\|kernels ms" $o/self150.log | tail -2 | cut -c1-250
