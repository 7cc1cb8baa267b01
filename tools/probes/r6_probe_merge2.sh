#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
export TMPDIR=/tmp
for v in oldmerge default oldmerge default; do
  if [ $v = default ]; then unset FGA_LIBRARY; else export FGA_LIBRARY=$root/fastga_amd/variants/lib_$v.so; fi
  echo "== $v $(timeout 200 python tools/merge_bench.py --reps 8 --check 2>&1 | grep "^rep" | sort -t' ' -k6 -n | head -3 | tail -1 | cut -c1-150)"
done
export FGA_LIBRARY=$root/fastga_amd/variants/lib_mprof.so
timeout 200 python tools/merge_bench.py --reps 3 2>&1 | tail -25
