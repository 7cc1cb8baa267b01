#!/bin/bash
# round 5, GPU call 19: code placement of the latency-bound extension (loop / block alignment), 12 launches each, twice
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5u; mkdir -p $o
export TMPDIR=/tmp
for v in default al64 al32 nofall nofall6 default al64 al32 nofall nofall6; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 12 --warmup 2 --no-human-scale --no-cold --batch 0 --no-cpu > $o/b_$v.log 2>&1
  echo "== $v: $(grep 'ms/step' $o/b_$v.log | cut -c1-16) kernel: $(grep 'extend profile' $o/b_$v.log | sed 's/.*kernel \([0-9.]*\) ms.*/\1/' | sort -n | head -8 | tr '\n' ' ')"
done
