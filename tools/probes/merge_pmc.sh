#!/bin/bash
# tools/merge_pmc.sh <tag> <counter> [<counter> ...] -- one rocprofv3 PMC pass over tools/merge_bench.py (merge stage only);
# counters in their own run, no trace domains.  Prints per-kernel averages per launch for the merge kernels.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=/tmp/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc "$@" -d $out -o pmc --output-format csv -- python $root/tools/merge_bench.py --reps 3 $MB_ARGS > $out/bench.log 2>&1
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1].add(r["Dispatch_Id"])
for (k, c), (v, d) in sorted(agg.items()):
    if "merge" in k or "range_cut" in k or "hole" in k:
        print(f"{k:50s} {c:28s} per launch = {v/max(1,len(d)):.5g}")
PY
tail -1 $out/bench.log | cut -c1-200
