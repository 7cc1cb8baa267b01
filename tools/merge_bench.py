#!/usr/bin/env python3
"""tools/merge_bench.py -- run only the seed-merge stage a few times (profiling / A-B driver): synthetic bench pair or the
repeat-heavy self genome, index built on the device; prints the stage time, the algorithmic GB/s and an order-independent
checksum of the seed set (so that two kernels can be compared without the oracle)."""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastga_amd import workload, device as D
from fastga_amd.gixio import Gdb

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=100.0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--self", action="store_true", dest="self_", help="repeat-heavy genome (config 3) against itself")
ap.add_argument("--mask", action="store_true")
ap.add_argument("--flip", action="store_true")
ap.add_argument("--freq", type=int, default=10)
ap.add_argument("--check", action="store_true", help="download the seeds and print a checksum")
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_mb_")
dev = D.Device(0)
if a.self_:
    roots = [workload.build_config3(d, mbp=a.mbp, threads=32)]
else:
    roots = list(workload.build_config2(d, mbp=a.mbp, threads=32))
tabs = []
for r in roots:
    g = Gdb(r + ".gdb")
    dg, x = D.build_gix_device(dev, g, 8, use_mask=a.mask)
    tabs.append((dg, x))
dA, A = tabs[0]
dB, B = (None, A) if a.self_ else tabs[1]
if a.flip:
    dA, A, dB, B = dB, B, dA, A
for r in range(a.reps):
    s = D.seed_merge(dev, dA, dB, freq=a.freq, soft_mask=a.mask, flip=a.flip)
    alg = A.nents * A.ebytes + (0 if a.self_ else B.nents * B.ebytes) + s.count * (1 + A.pbyte + B.pbyte)
    ms = dev.stage_ms(D.STAGE_MERGE)
    line = (f"rep {r}: seeds {s.count} merge {ms:.3f} ms (cuts/partition {dev.stage_ms(D.STAGE_MERGE_PARTITION):.3f} ms)  "
            f"{alg/ms/1e6:.0f} GB/s ({alg/ms/1e6/80:.1f}% of 8 TB/s)")
    if a.check and r == a.reps - 1:
        h = s.download()
        v = (h["apos"].astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (h["bpos"].astype(np.uint64) << np.uint64(17)) \
            ^ (h["actg"].astype(np.uint64) << np.uint64(34)) ^ (h["bctg"].astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F))
        v = (v ^ (v >> np.uint64(29))) * np.uint64(0xBF58476D1CE4E5B9)
        line += f"  checksum {int(v.sum()) & 0xffffffffffffffff:016x}/{int(np.bitwise_xor.reduce(v)):016x}"
    print(line, flush=True)
    s.free()
