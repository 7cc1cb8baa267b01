#!/bin/bash
# round 5, GPU call 16: the latency-bound extension with 1 / 2 / 3 / 4 resident wavefronts per CU: distribution of the kernel time
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5q; mkdir -p $o
export TMPDIR=/tmp
for w in 1024 256 512 768 1024 256; do
  FGA_EXTEND_WGS=$w FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 20 --warmup 3 --no-human-scale --batch 0 --no-cold --no-cpu > $o/b_$w.log 2>&1
  echo "== $w: $(grep 'ms/step' $o/b_$w.log | cut -c1-60)"
  grep "extend profile" $o/b_$w.log | sed 's/.*kernel \([0-9.]*\) ms.*/\1/' | sort -n | tr '\n' ' '; echo
done
