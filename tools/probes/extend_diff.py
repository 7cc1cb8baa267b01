#!/usr/bin/env python3
"""debug: run merge+sort+chain once, then the extension twice (register path vs forced LDS ring) and diff."""
import os, sys, tempfile, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import workload, device as D
from fastga_amd.gixio import Gix, Gdb
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
d = tempfile.mkdtemp(prefix="fga_xd_")
ra, rb = workload.build_pair(d, seed=1, ncontig=40, total=int(mbp*1e6), divergence=0.02, repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02, threads=32)
A, B = Gix(ra+".gix"), Gix(rb+".gix"); ga, gb = Gdb(ra+".gdb"), Gdb(rb+".gdb")
amx, bmx = int(ga.maxctg), int(gb.maxctg)
dev = D.Device(0); dA, dB = dev.upload(A), dev.upload(B)
seeds = D.seed_merge(dev, dA, dB); keys = D.seed_sort(dev, seeds, amx, bmx, A.nctg, B.nctg); kh = keys.download()
hl = D.chain_scan(kh, (keys.wa, keys.wb, keys.wd, keys.wt), 2000, 170, amx, bmx, ga.clen[A.perm], nthreads=16)
f4 = (C.c_float*4)(); ga.L.fga_gdb_freq(ga.h, f4)
pa, table, score = D.align_spec(0.7, 100, list(f4))
dga = D.DeviceGenome(dev, ga, A.perm, True); dgb = D.DeviceGenome(dev, gb, B.perm, False)
def run():
    alns, tb, st = D.extend(dev, dga, dgb, hl, pa, table, score)
    order = np.lexsort((alns["seq"], alns["unit"]))
    return alns[order], tb, st
sys.stdout.flush(); print("=== REG"); sys.stdout.flush(); a1, t1, s1 = run()
os.environ["FGA_EXTEND_FORCE_LDS"] = "1"
sys.stdout.flush(); print("=== LDS"); sys.stdout.flush(); a2, t2, s2 = run()
print("reg:", len(a1), s1, " lds:", len(a2), s2)
units = hl.units; hits = hl.hits
bad = 0
for i in range(min(len(a1), len(a2))):
    x, y = a1[i], a2[i]
    same = all(int(x[f]) == int(y[f]) for f in ("unit","seq","abpos","bbpos","aepos","bepos","diffs","tlen")) and \
           np.array_equal(t1[int(x["toff"]):int(x["toff"])+int(x["tlen"])], t2[int(y["toff"]):int(y["toff"])+int(y["tlen"])])
    if not same:
        bad += 1
        if bad <= 5:
            U = units[int(x["unit"])]
            print("DIFF unit", int(x["unit"]), dict(actg=int(U["actg"]), bctg=int(U["bctg"]), comp=int(U["comp"]), nhits=int(U["nhits"])))
            print("  reg", [int(x[f]) for f in ("seq","abpos","bbpos","aepos","bepos","diffs","tlen")])
            print("  lds", [int(y[f]) for f in ("seq","abpos","bbpos","aepos","bepos","diffs","tlen")])
            print("  regT", list(t1[int(x["toff"]):int(x["toff"])+int(x["tlen"])]))
            print("  ldsT", list(t2[int(y["toff"]):int(y["toff"])+int(y["tlen"])]))
            print("  alen", int(ga.clen[A.perm[U["actg"]]]), "blen", int(gb.clen[B.perm[U["bctg"]]]))
            for h in hits[U["first_hit"]:U["first_hit"]+U["nhits"]][:3]:
                print("   hit", [int(h[f]) for f in ("dgmin","dgmax","alow","ahgh")])
print("mismatching alignments:", bad)
