#!/bin/bash
# one rocprofv3 PMC pass over a single bench step; usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]
# (counters in their own run, no trace domains -- see the task notes on gpurun + --pmc)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" -d $out -o pmc --output-format csv -- python $root/bench.py --steps 1 --warmup 0 --no-cpu > $out/bench.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
disp = collections.defaultdict(set)
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k:60s} {c:24s} total={v:.4g} rows={n}")
PY
tail -1 $out/bench.log | cut -c1-600
