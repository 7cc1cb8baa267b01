#!/bin/bash
# round 6, GPU call 3: scalar-side bookkeeping in integer arithmetic; parity, A/B, instruction counts per step
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
o=$root/gpurun_out/r6c; mkdir -p $o
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_extend_gpu.py tests/test_shims_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/t1.log 2>&1; tail -1 $o/t1.log
( timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/t2.log 2>&1; tail -1 $o/t2.log
( FGA_EXTEND_NARROW=1 timeout 900 python -m pytest tests/test_end_to_end_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -5 ) > $o/t3.log 2>&1; tail -1 $o/t3.log
for v in ${VARIANTS:-before default before default}; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib timeout 300 python tools/scale_check.py --mbp 150 --self > $o/self150_$v.log 2>&1
  echo "== $v $(grep 'kernels ms' $o/self150_$v.log | tail -1 | sed 's/.*kernels ms/kernels ms/' | cut -c1-160)"
done
for v in ${BVARIANTS:-before default}; do
  lib=$root/fastga_amd/variants/lib_$v.so; [ $v = default ] && lib=$root/fastga_amd/libfastga_amd.so
  FGA_LIBRARY=$lib FGA_EXTEND_PROFILE=1 timeout 300 python tools/bench_brief.py --steps 12 --warmup 2 --no-human-scale --no-cold --batch 0 --no-cpu > $o/b_$v.log 2>&1
  echo "== $v: $(grep 'ms/step' $o/b_$v.log | cut -c1-60) kernel: $(grep 'extend profile' $o/b_$v.log | sed 's/.*kernel \([0-9.]*\) ms.*/\1/' | sort -n | head -6 | tr '\n' ' ')"
done
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $o/pmc -o pmc --output-format csv -- python $root/tools/scale_check.py --mbp 150 --self > $o/pmc.log 2>&1
python - "$o/pmc" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ext_mid" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
steps = 2 * 564733138
print("per wave step:", {k: round(v / steps, 1) for k, v in sorted(agg.items())})
PY
