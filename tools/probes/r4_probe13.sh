#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for v in base ext_maxilp ext_a ext_b ext_c ext_d; do
  lib=$root/fastga_amd/libfastga_amd.so; [ $v != base ] && lib=$root/fastga_amd/variants/lib_$v.so
  [ -f $lib ] || { echo "$v: no library"; continue; }
  FGA_LIBRARY=$lib timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-cold --no-human-scale 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$v', 'bench ms/step', round(j['ms_per_step'],2), 'extend kernel', round(j['config']['kernel_ms']['extend'],2), 'waves', j['config']['waves'], 'records', j['config']['records'])"
  FGA_LIBRARY=$lib timeout 200 python tools/scale_check.py --mbp 150 --self 2>/dev/null | grep "run 1\|kernels ms" | tail -2 | cut -c1-220
done
