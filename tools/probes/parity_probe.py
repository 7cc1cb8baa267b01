#!/usr/bin/env python3
"""tools/parity_probe.py -- ours vs the reference FastGA on one synthetic pair; prints where the .1aln texts differ"""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import workload, device as D
from oracle import harness as H
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=5.0)
ap.add_argument("--div", type=float, default=0.001)
ap.add_argument("--contigs", type=int, default=8)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_pp_")
ra, rb = workload.build_pair(d, seed=a.seed, ncontig=a.contigs, total=int(a.mbp * 1e6), divergence=a.div,
                             repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02, threads=16)
st = D.run(ra, rb, os.path.join(d, "ours.1aln"), nthreads=8)
H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=8)
x = [l for l in H.oneview(os.path.join(d, "ours.1aln")) if l[0] not in "!<"]
y = [l for l in H.oneview(os.path.join(d, "ref.1aln")) if l[0] not in "!<"]
print("seeds", st["nseeds"], "hits", st["nhits"], "alns", st["nalns"], "live", st["nlive"], "lines", len(x), len(y), "equal", x == y)
if x != y:
    na = sum(1 for l in x if l.startswith("A ")); nb = sum(1 for l in y if l.startswith("A "))
    print("records ours/ref", na, nb)
    shown = 0
    for i, (p, q) in enumerate(zip(x, y)):
        if p != q:
            print(i, "OURS", p[:160]); print(i, "REF ", q[:160])
            shown += 1
            if shown >= 4: break
    ra_ = set(l for l in x if l.startswith("A ")); rb_ = set(l for l in y if l.startswith("A "))
    print("A lines only ours:", sorted(ra_ - rb_)[:5]); print("A lines only ref:", sorted(rb_ - ra_)[:5])
