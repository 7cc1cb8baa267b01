#!/bin/bash
# which build first loses alignments under concurrent launches (fga_run_multi with virtual ranks on one GPU)?
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5s4; mkdir -p $o
export TMPDIR=/tmp
for v in f40d152 4f8ebbe 2477db9 default; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 300 python tools/multi_stress.py 50 > $o/$v.log 2>&1
  echo "== $v: $(tail -1 $o/$v.log)"; grep "^iteration" $o/$v.log | head -3 | cut -c1-200
done
