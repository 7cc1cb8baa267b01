#!/usr/bin/env python3
"""tools/trace_bench.py -- time the device trace stage (fga_trace_pts) on the alignments of a synthetic pair, with the
CPU oracle (oracle/trace_oracle.c, one core) timed on a sample of the same alignments beside it"""
import argparse, ctypes as C, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import workload, synth, device as D
from fastga_amd.gixio import Gdb

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=100.0)
ap.add_argument("--div", type=float, default=0.02)
ap.add_argument("--contigs", type=int, default=40)
ap.add_argument("--cpu-sample", type=int, default=2000)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_tb_")
ra, rb = workload.build_pair(d, seed=1, ncontig=a.contigs, total=int(a.mbp * 1e6), divergence=a.div,
                             repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02, threads=16, gix=False)
ga, gb = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
dev = D.Device(0)
dA, xa = D.build_gix_device(dev, ga, 8, host_copy=True)
dB, xb = D.build_gix_device(dev, gb, 8, host_copy=True)
amx, bmx = int(ga.maxctg), int(gb.maxctg)
seeds = D.seed_merge(dev, dA, dB)
keys = D.seed_sort(dev, seeds, amx, bmx, xa.nctg, xb.nctg)
clen = np.zeros(xa.nctg, dtype=np.int64); clen[:len(ga.clen)] = ga.clen
hl = D.chain_scan_device(dev, keys, 2000, 170, amx, bmx, clen[xa.perm])
f4 = (C.c_float * 4)(); ga.L.fga_gdb_freq(ga.h, f4)
pa, table, score = D.align_spec(0.7, 100, list(f4))
dga = D.DeviceGenome(dev, ga, xa.perm, True)
dgb = D.DeviceGenome(dev, gb, xb.perm, True)
alns, tb, _ = D.extend(dev, dga, dgb, hl, pa, table, score, aln_min=50, aln_rate=0.35)
keys.free(); seeds.free()
bases = int((alns["aepos"] - alns["abpos"]).sum())
print(f"{len(alns)} alignments, {bases/1e6:.1f} Mbp aligned, {int(alns['tlen'].sum())//2} panels, "
      f"{int(alns['diffs'].sum())} diffs", flush=True)
for rep in range(a.reps):
    t = time.time()
    toff, tlen, diffs, ints, st = D.trace_pts(dev, dga, dgb, alns, tb)
    w = time.time() - t
    print(f"rep {rep}: trace_pts {1000*w:.1f} ms wall, device {dev.stage_ms(6):.2f} ms, {len(ints)} indels, "
          f"{st['panels']/dev.stage_ms(6)/1e3:.2f} M panels/s on the device", flush=True)
if a.cpu_sample > 0:
    from oracle import harness as H
    rng = np.random.default_rng(1)
    pick = rng.choice(len(alns), size=min(a.cpu_sample, len(alns)), replace=False)
    cache = {}
    jobs = []
    for i in pick:
        x = alns[i]
        c1, c2, comp = int(x["aread"]), int(x["bread"]), int(x["flags"]) & 1
        if ("a", c1) not in cache:
            cache[("a", c1)] = H.pad_seq(ga.contig(c1))
        if ("b", c2, comp) not in cache:
            s = gb.contig(c2)
            cache[("b", c2, comp)] = H.pad_seq(synth.revcomp(s) if comp else s)
        t16 = tb[int(x["toff"]):int(x["toff"]) + int(x["tlen"])].astype(np.uint16)
        jobs.append((i, cache[("a", c1)], cache[("b", c2, comp)],
                     (int(x["abpos"]), int(x["bbpos"]), int(x["aepos"]), int(x["bepos"]), int(x["diffs"]), t16)))
    t = time.time()
    npan = 0
    for i, sa, sb, path in jobs:
        od, ot = H.oracle_trace_pts(sa, sb, path)
        npan += len(path[5]) // 2
        assert od == int(diffs[i]) and np.array_equal(ot, ints[int(toff[i]):int(toff[i + 1])])
    w = time.time() - t
    print(f"cpu oracle (1 core): {len(jobs)} alignments, {npan} panels in {w:.2f} s = {npan/w/1e6:.3f} M panels/s "
          f"(all equal to the device result)", flush=True)
