#!/bin/bash
# round 5, GPU call 22: the pair kernel's descriptor without the flag again: parity of the merge suites (short ones), A/B on the bench pair
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r5aa; mkdir -p $o
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_seed_merge_gpu.py tests/test_end_to_end_gpu.py tests/test_mask_files_gpu.py -x -q -m gpu -k "not dense and not 450 and not 300" 2>&1 | tail -2 ) > $o/t1.log 2>&1; tail -1 $o/t1.log
for v in before default before default; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 200 python tools/bench_brief.py --steps 20 --warmup 3 --no-human-scale --no-cold --batch 0 --no-cpu > $o/b_$v.log 2>&1; echo "== $v $(grep 'kernel_ms' $o/b_$v.log | cut -c1-80)"
done
