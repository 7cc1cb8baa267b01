#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
for v in base ext_b all_ilp; do
  lib=$root/fastga_amd/libfastga_amd.so; [ $v != base ] && lib=$root/fastga_amd/variants/lib_$v.so
  [ -f $lib ] || { echo "$v: no library"; continue; }
  FGA_LIBRARY=$lib timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu --no-cold --no-human-scale 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); c=j['config']; print('$v', 'bench ms/step', round(j['ms_per_step'],2), 'kernels', c['kernel_ms'], 'stages', c['stage_ms'], 'gix', c['gix_build_on_device_ms'], 'frac', round(j['roofline']['frac'],3))"
  FGA_LIBRARY=$lib timeout 200 python tools/scale_check.py --mbp 150 --self 2>/dev/null | grep "kernels ms" | tail -1 | sed 's/.*kernels ms/  150self kernels ms/' | cut -c1-150
done
FGA_LIBRARY=$root/fastga_amd/variants/lib_all_ilp.so timeout 300 python tools/config4_check.py --mbp 3000 --div 0.01 --no-digest 2>/dev/null | grep "session_run\|stages\|upload" | cut -c1-330
