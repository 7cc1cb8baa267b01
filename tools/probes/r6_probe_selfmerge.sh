#!/bin/bash
# round 6: where the seed merge's wavefront cycles go in SELF mode (-DMERGE_PROF build), beside the pair
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; export TMPDIR=/tmp
for args in "--self --mask --mbp 300" "--mbp 100"; do
  echo "== default [$args] $(timeout 200 python tools/merge_bench.py --reps 6 $args 2>&1 | grep "^rep" | sort -t' ' -k6 -n | head -3 | tail -1 | cut -c1-170)"
  echo "== mprof [$args]"; FGA_LIBRARY=$root/fastga_amd/variants/lib_mprof.so timeout 200 python tools/merge_bench.py --reps 2 $args 2>&1 | tail -8
done
