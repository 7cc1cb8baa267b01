#!/bin/bash
# round 5, GPU call 17: wavefronts per CU of the latency-bound extension chosen from the hit-box bases: bench pair, the shapes around it
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5r; mkdir -p $o
export TMPDIR=/tmp
FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 20 --warmup 3 --no-human-scale --no-cold > $o/b.log 2>&1
grep "ms/step\|kernel_ms\|cpu" $o/b.log | cut -c1-200
grep "extend profile" $o/b.log | tail -2 | cut -c150-500
for args in "--mbp 20" "--mbp 50 --div 0.05" "--mbp 100 --div 0.10" "--mbp 100 --contigs 400"; do
  for w in 0 1024; do
    FGA_EXTEND_WGS=$w FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 6 --warmup 2 --no-human-scale --no-cold --no-cpu --batch 0 $args > $o/x.log 2>&1
    echo "== $args wgs=$w: $(grep 'ms/step' $o/x.log | cut -c1-30) $(grep 'extend profile' $o/x.log | tail -1 | sed 's/.*kernel \([0-9.]* ms, [0-9]* workgroups, [0-9]* units\).*bases \(.*\)/\1 \2/')"
  done
done
( timeout 600 python -m pytest tests/test_extend_gpu.py tests/test_end_to_end_gpu.py tests/test_parts_gpu.py tests/test_multi_gpu.py -x -q -m gpu 2>&1 | tail -3 ) > $o/t.log 2>&1; tail -1 $o/t.log
