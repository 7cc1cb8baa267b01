#!/bin/bash
# tools/sort_pmc.sh <n keys> [env assignments...] -- PMC passes (one rocprofv3 run per counter group, no trace domains) over
# tools/ubench/sort_bench; per-kernel totals.  Run ON the GPU box (through gpurun).
n=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/sort_pmc
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift
  timeout 200 env "${ENVS[@]}" rocprofv3 --pmc "$@" -d $out/$name -o pmc --output-format csv -- $root/fastga_amd/bin/sort_bench $n 61 12 uniform 1 > $out/$name.log 2>&1
}
ENVS=("$@" FGA_LIBRARY=$root/fastga_amd/libfastga_amd.so)
pass fetch FETCH_SIZE WRITE_SIZE
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
pass lds SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
pass lds2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    if "pass" in k: print(f"{k:40s} {c:24s} per launch={v/n:.4g} launches={n}")
PY
