#!/usr/bin/env python3
"""tools/merge_bench.py -- run only the seed-merge stage a few times on a synthetic pair (profiling driver)."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import workload, device as D
from fastga_amd.gixio import Gix

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=100.0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--self", action="store_true", dest="self_")
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_mb_")
ra, rb = workload.build_pair(d, seed=1, ncontig=40, total=int(a.mbp * 1e6), divergence=0.02,
                             repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02, threads=32)
A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
dev = D.Device(0)
dA, dB = dev.upload(A), dev.upload(B)
for r in range(a.reps):
    s = D.seed_merge(dev, dA, None if a.self_ else dB)
    alg = A.nents * A.ebytes + (0 if a.self_ else B.nents * B.ebytes) + s.count * (1 + A.pbyte + B.pbyte)
    ms = dev.stage_ms(D.STAGE_MERGE)
    print(f"rep {r}: seeds {s.count} merge {ms:.3f} ms  partition {dev.stage_ms(D.STAGE_MERGE_PARTITION):.3f} ms  "
          f"{alg/ms/1e6:.0f} GB/s ({alg/ms/1e6/80:.1f}% of 8 TB/s)", flush=True)
    s.free()
