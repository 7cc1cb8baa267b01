#!/usr/bin/env python3
"""round 6: fga_session_run in several passes (the .1aln streamed pass by pass), again and again on one session: every run must
give the one-pass run's file (record lines in sequence)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fastga_amd import workload, device as D
from oracle import harness as H
it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = tempfile.mkdtemp(prefix="fga_sp_")
ra, rb = workload.build_pair(d, seed=12, ncontig=16, total=900_000, divergence=0.03, repeat_frac=0.05, inv_frac=0.05, swap_frac=0.05)
keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]
ses = D.Session(ra, rb, nthreads=8)
one = ses.run(out_path=os.path.join(d, "one.1aln"), nthreads=8, reference_threads=8)
ref = keep(H.oneview(os.path.join(d, "one.1aln")))
bad = 0
for k in range(it):
    for div in (3, 5, 9):
        out = os.path.join(d, "m.1aln")
        st = ses.run(out_path=out, nthreads=8, reference_threads=8, pass_seeds=max(1, one["nseeds"] // div))
        got = keep(H.oneview(out))
        if got != ref or st["nlive"] != one["nlive"]:
            bad += 1
            print(f"iteration {k} parts {st['nparts']} streamed {st['streamed_parts']}: records {st['nlive']} ({one['nlive']})", flush=True)
print("parts of the last run:", st["nparts"], "streamed", st["streamed_parts"], "| runs that differ:", bad, "of", 3 * it)
ses.close()
