#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_seed_merge_gpu.py tests/test_mask_files_gpu.py -x -q -m gpu 2>&1 | tail -3 )
for args in "--self --mask --mbp 300" "--mask" ""; do
  for v in oldmerge default; do
    if [ $v = default ]; then unset FGA_LIBRARY; else export FGA_LIBRARY=$root/fastga_amd/variants/lib_$v.so; fi
    echo "== $v [$args] $(timeout 200 python tools/merge_bench.py --reps 8 --check $args 2>&1 | grep "^rep" | sort -t' ' -k6 -n | head -3 | tail -1 | cut -c1-170)"
  done
done
