#!/bin/bash
# round 6: seed merge A/B (round-5 library vs current): parity suite, bench pair, self mode, flip
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r6g; mkdir -p $o
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_seed_merge_gpu.py -x -q -m gpu 2>&1 | tail -4 ) > $o/t.log 2>&1; tail -2 $o/t.log
for args in "" "--self --mask --mbp 300" "--flip" ; do
  for v in before default before default; do
    if [ $v = default ]; then unset FGA_LIBRARY; else export FGA_LIBRARY=$root/fastga_amd/variants/lib_$v.so; fi
    echo "== $v [$args] $(timeout 200 python tools/merge_bench.py --reps 6 --check $args 2>&1 | grep "^rep" | sort -t' ' -k6 -n | head -2 | tail -1 | cut -c1-150)"
  done
done
unset FGA_LIBRARY
