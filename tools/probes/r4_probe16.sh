#!/bin/bash
# compiler options for fga_extend.hip on top of max-ilp / no SLP: which ones change the ISA was checked on the build host
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for v in base wprio trk iter phi8 taildup; do
  lib=$root/fastga_amd/libfastga_amd.so; [ $v != base ] && lib=$root/fastga_amd/variants/lib_$v.so
  [ -f $lib ] || { echo "$v: no library"; continue; }
  FGA_LIBRARY=$lib timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu --no-cold --no-human-scale 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$v', 'bench ms/step', round(j['ms_per_step'],2), 'extend kernel', round(j['config']['kernel_ms']['extend'],2), 'waves', j['config']['waves'], 'records', j['config']['records'])"
done
