#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
o=$root/gpurun_out/r4i; mkdir -p $o
cd /tmp && export TMPDIR=/tmp
for v in base gix_NO_SAMPLE gix_NO_COUNT gix_NO_PLACE_ATOMIC gix_NO_PLACE_STORE; do
  lib=$root/fastga_amd/libfastga_amd.so; [ $v != base ] && lib=$root/fastga_amd/variants/lib_$v.so
  FGA_LIBRARY=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/$v -o kt -- python $root/tools/gix_scan_probe.py --mbp 1000 > $o/$v.log 2>&1
  echo "== $v: $(grep 'rep 1' $o/$v.log | cut -c1-120)"
  python - $o/$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if r["Name"].startswith("gix_"):
            print(f"   {r['Name'][:40]:42s} calls {r['Calls']:>3} avg {float(r['AverageNs'])/1e6:8.3f} ms  min {float(r['MinNs'])/1e6:8.3f} max {float(r['MaxNs'])/1e6:8.3f}")
PY
done
