#!/bin/bash
# tools/scale_prof.sh <tag> -- rocprofv3 --kernel-trace --stats of the other shapes (run ON THE GPU BOX through gpurun):
#   gpurun_out/<tag>_config4_kernel_stats.csv     3 Gbp x 3 Gbp, 1 % (BASELINE configs[3]) on one GPU: tools/config4_check.py
#   gpurun_out/<tag>_config3_kernel_stats.csv     1 Gbp repeat-heavy self comparison with -M (configs[2]): tools/config3_check.py
#   gpurun_out/<tag>_throughput_kernel_stats.csv  150 Mbp repeat-heavy self comparison, 10^6 units: tools/scale_check.py
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $out/prof_$tag/c4 -o kt --output-format csv -- python $root/tools/config4_check.py --mbp 3000 --div 0.01 --no-digest > $out/prof_$tag/c4.log 2>&1
cp $out/prof_$tag/c4/kt_kernel_stats.csv $out/${tag}_config4_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag/c3 -o kt --output-format csv -- python $root/tools/config3_check.py --mbp 1000 > $out/prof_$tag/c3.log 2>&1
cp $out/prof_$tag/c3/kt_kernel_stats.csv $out/${tag}_config3_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag/thr -o kt --output-format csv -- python $root/tools/scale_check.py --mbp 150 --self --repeats 0.30 > $out/prof_$tag/thr.log 2>&1
cp $out/prof_$tag/thr/kt_kernel_stats.csv $out/${tag}_throughput_kernel_stats.csv
# PMC passes over the same shapes (each counter group in its own run, no trace domains): the kernels that dominate them --
# the sort passes, ext_mid, the seed merge at 3 Gbp and in self mode, the index build -- with their counters per launch
# PMC_GROUPS="fetch write sq" limits the counter groups of a shape (GPU minutes: a 3 Gbp pass is a minute);
# SCALE_PROF_FULL=1 also takes the 1 Gbp self comparison's passes
pmc_shape() { shape=$1; shift
  for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" \
             "sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "lds SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    set -- $grp; name=$1; shift
    case " ${PMC_GROUPS:-fetch write sq sq2 lds} " in *" $name "*) ;; *) continue ;; esac
    timeout 400 rocprofv3 --pmc "$@" -d $out/prof_$tag/pmc_${shape}_$name -o pmc --output-format csv -- $CMD > $out/prof_$tag/pmc_${shape}_$name.log 2>&1
  done
  python - "$out/prof_$tag" "$shape" "$out/${tag}_${shape}_pmc_summary.csv" "$CMD" <<'PY'
import csv, glob, sys, collections
src, shape, dst, cmd = sys.argv[1:5]
rows = []
for name in ("fetch", "write", "sq", "sq2", "lds"):
    agg = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(f"{src}/pmc_{shape}_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1].add(r["Dispatch_Id"])
    for (k, c), (v, d) in sorted(agg.items()):
        if any(x in k for x in ("pass_kernel", "extend_kernel", "seed_merge", "gix_", "chain_small", "chain_segment")):
            rows.append(("pmc_" + name, k, c, len(d), v / max(1, len(d))))
with open(dst, "w") as f:
    f.write("# command: %s under rocprofv3 --pmc <counters> (one pass per counter group); FETCH_SIZE / WRITE_SIZE in KiB\n" % cmd)
    f.write("pass,kernel,counter,launches,avg_per_launch\n")
    w = csv.writer(f, lineterminator="\n")
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], "%.6g" % r[4]])
PY
}
CMD="python $root/tools/config4_check.py --mbp 3000 --div 0.01 --no-digest"; PMC_GROUPS="${PMC_GROUPS_3G:-fetch write sq}" pmc_shape config4
CMD="python $root/tools/scale_check.py --mbp 150 --self --repeats 0.30"; pmc_shape throughput
if [ "${SCALE_PROF_FULL:-0}" = 1 ]; then CMD="python $root/tools/config3_check.py --mbp 1000"; pmc_shape config3; fi
grep -h "fga_session_run\|stages\|run 1" $out/prof_$tag/c4.log $out/prof_$tag/c3.log $out/prof_$tag/thr.log
for f in config4 config3 throughput; do echo "== $f"; head -14 $out/${tag}_${f}_kernel_stats.csv | cut -c1-150; done
