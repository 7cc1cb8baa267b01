#!/usr/bin/env python3
"""tools/gixdev_bench.py -- time the device GIX build on a synthetic genome (GDB made by the host producer)"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import synth, device as D
from fastga_amd.gixio import Gdb, fasta_to_gdb
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=100.0)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_gb_")
lens = synth.contig_lengths(3, 40, int(a.mbp * 1e6))
A, _, _, _ = synth.make_pair(3, lens, 0.0, repeat_frac=0.05, self_only=True)
fa = os.path.join(d, "A.fa")
synth.write_fasta(fa, A, prefix="a")
t = time.time(); fasta_to_gdb(fa, os.path.join(d, "A")); print(f"FASTA->GDB {time.time()-t:.2f} s")
g = Gdb(os.path.join(d, "A.gdb"))
dev = D.Device(0)
for rep in range(3):
    t = time.time()
    dg, x = D.build_gix_device(dev, g, 8)
    print(f"rep {rep}: device GIX build {1000*(time.time()-t):.1f} ms wall, kernels {dev.stage_ms(5):.2f} ms, "
          f"{x.nents} entries x {x.ebytes} B", flush=True)
    dg.free(); x.close()
