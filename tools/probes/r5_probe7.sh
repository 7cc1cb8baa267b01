#!/bin/bash
# round 5, GPU call 7: cache policy of the seed merge's streams (nt loads of the table tiles, nt stores of the seeds), A/B on
# the bench pair and on the 3 Gbp pair's merge; the seed tests on the winner come with the next full suite
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5g; mkdir -p $o
export TMPDIR=/tmp
for v in default merge_ntl merge_nts merge_ntls default merge_ntl merge_nts merge_ntls; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 200 python bench.py --steps 20 --warmup 5 --no-human-scale --no-cpu --no-cold --batch 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'launch ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'walk', round(r['walk_kernel_ms'],4), round(r['walk_kernel_frac'],4), 'seeds', d['config']['seeds'], 'records', d['config']['records'])"
done
