#!/usr/bin/env python3
"""tools/multi_stress.py [iterations] -- fga_run_multi with virtual ranks on the toy pair, again and again: every run must give
the single-session run's records.  Prints the runs that do not and what they lack."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fastga_amd import workload, device as D
from oracle import harness as H
it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
d = tempfile.mkdtemp(prefix="fga_ms_")
ra, rb = workload.build_pair(d, seed=11, ncontig=12, total=600_000, divergence=0.03, repeat_frac=0.05, inv_frac=0.05, swap_frac=0.05)
keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]
one = D.run(ra, rb, os.path.join(d, "one.1aln"), nthreads=8, reference_threads=8)
ref = keep(H.oneview(os.path.join(d, "one.1aln")))
recs = lambda lines: [ln for ln in lines if ln.startswith("A ")]
print("single run: nalns", one["nalns"], "records", len(recs(ref)), flush=True)
bad = 0
for k in range(it):
    for devices in ((0, 0, 0), (0, 0), (0, 0, 0, 0)):
        out = os.path.join(d, "m.1aln")
        st = D.run_multi(ra, rb, out, devices=devices, nthreads=8, reference_threads=8)
        got = keep(H.oneview(out))
        if got != ref:
            bad += 1
            miss = [ln for ln in recs(ref) if ln not in set(recs(got))]
            print(f"iteration {k} devices {devices}: nseeds {st['nseeds']} (single {one['nseeds']}) nhits {st['nhits']} ({one['nhits']}) "
                  f"nalns {st['nalns']} ({one['nalns']}) records {len(recs(got))} ({len(recs(ref))}); missing: {miss[:6]}", flush=True)
print("runs that differ:", bad, "of", 3 * it)
