#!/bin/bash
# round 6: PMC passes over the throughput shape (150 Mbp self), extension kernel only; usage: r6_pmc_thr.sh <tag> [lib]
tag=${1:-r6}; lib=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export FGA_LIBRARY=$lib
for grp in "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "sq3 SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_MISC"; do
  set -- $grp; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $out/$name -o pmc --output-format csv -- python $root/tools/scale_check.py --mbp 150 --self > $out/$name.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "extend_kernel" not in k: continue
        agg[(k, r["Counter_Name"])][0] += float(r["Counter_Value"]); agg[(k, r["Counter_Name"])][1].add(r["Dispatch_Id"])
with open(out + "/summary.csv", "w") as f:
    f.write("kernel,counter,launches,total\n")
    for (k, c), (v, d) in sorted(agg.items()):
        f.write(f"{k},{c},{len(d)},{v:.6g}\n"); print(f"{k:28s} {c:26s} n={len(d)} total={v:.5g}")
PY
grep -h "waves\|wave steps\|kernels ms" $out/sq.log | tail -3
