#!/usr/bin/env python3
"""tools/pass_times.py <kernel_trace.csv> [name substring ...] -- the dispatches of the named kernels in launch order:
duration and grid size each (the per-pass times of the radix sort, which the --stats summary averages away)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:] or ["os_pass", "os2_pass", "first_hist"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    nm = r["Kernel_Name"]
    if any(p in nm for p in pats):
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        gx = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        wx = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
        print(f"{dur:9.3f} ms  grid {gx:>12} wg {wx:>5}  {nm[:60]}")
