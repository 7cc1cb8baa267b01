#!/bin/bash
# tools/profile_round.sh <tag> -- rocprofv3 evidence for one round, run ON THE GPU BOX (through gpurun):
#   gpurun_out/<tag>_kernel_stats.csv   --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --no-cpu`
#   gpurun_out/<tag>_pmc_summary.csv    per-kernel averages of the PMC passes (each in its own run, no trace domains):
#                                       pmc_fetch = FETCH_SIZE, pmc_write = WRITE_SIZE (KiB; gfx950: FETCH_SIZE counts
#                                       half of a wide coalesced stream, see MI355X_MICROARCH.md), pmc_sq = SQ issue/wait
# Copy the two files into profiles/ afterwards.
tag=${1:-r03}
commit=${2:-unknown}          # the commit the snapshot was taken at (gpurun ships no .git): pass `git rev-parse --short HEAD`
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out/prof_$tag
cd /tmp && export TMPDIR=/tmp
BENCH="python $root/bench.py --steps 5 --warmup 2 --no-cpu --no-cold --no-human-scale --batch 0"
timeout 200 rocprofv3 --kernel-trace --stats -d $out/prof_$tag/kt -o kt --output-format csv -- $BENCH > $out/prof_$tag/kt.log 2>&1
cp $out/prof_$tag/kt/kt_kernel_stats.csv $out/${tag}_kernel_stats.csv
grep -a '"metric"' $out/prof_$tag/kt.log | tail -1 > $out/${tag}_bench_under_rocprof.json
pass() { name=$1; shift; echo "pass $name" >&2
  timeout 120 rocprofv3 --pmc "$@" -d $out/prof_$tag/$name -o pmc --output-format csv -- $BENCH > $out/prof_$tag/$name.log 2>&1
}
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES
pass pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
pass pmc_lds SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
python - "$out/prof_$tag" "$out/${tag}_pmc_summary.csv" "$commit" <<'PY'
import csv, glob, sys, collections
src, dst = sys.argv[1], sys.argv[2]
rows = []
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_lds"):
    agg = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(f"{src}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1].add(r["Dispatch_Id"])
    for (k, c), (v, d) in sorted(agg.items()):
        rows.append((name, k, c, len(d), v / max(1, len(d))))
with open(dst, "w") as f:
    f.write("# commit: %s\n" % (sys.argv[3] if len(sys.argv) > 3 else "unknown"))
    f.write("# command: bench.py --steps 5 --warmup 2 --no-cpu --no-cold --no-human-scale under rocprofv3 --pmc <counters> (one pass per counter group)\n")
    f.write("pass,kernel,counter,launches,avg_per_launch\n")
    w = csv.writer(f, lineterminator="\n")               # (a template kernel's name holds commas: quoted)
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], "%.6g" % r[4]])
print(open(dst).read())
PY
