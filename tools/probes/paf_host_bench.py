#!/usr/bin/env python3
"""tools/paf_host_bench.py -- where the host side of `-pafx` spends its time (no GPU needed): a repeat-heavy self comparison
made by the REAL reference on the CPU (oracle/_ref), its edit scripts from the CPU oracle of the trace stage, then
fga_gap_improve alone and fga_write_paf (-x) with FGA_PAF_TIMING=1 (threads' compute span / file write)."""
import argparse, ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastga_amd import workload, synth
from fastga_amd.lib import load_library, Alns, Traces
from fastga_amd.gixio import Gdb
from oracle import harness as H
from tests.test_aln_reader import read_1aln

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=12.0)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--repeats", type=float, default=0.30)
ap.add_argument("--keep", default=None, help="directory to keep / reuse the set in")
a = ap.parse_args()
d = a.keep or tempfile.mkdtemp(prefix="fga_pafb_")
os.makedirs(d, exist_ok=True)
L = load_library()
ra = os.path.join(d, "A")
if not os.path.exists(os.path.join(d, "ref.1aln")):
    lens = synth.contig_lengths(2, 40, int(a.mbp * 1e6))
    A, mA, _, _ = synth.make_pair(2, lens, 0.02, repeat_frac=a.repeats, inv_frac=0.02, swap_frac=0.02, self_only=True)
    ra = workload.build_genome(d, "A", A, threads=a.threads, gix=True)
    t = time.time()
    H.ref_fastga(ra, None, d, os.path.join(d, "ref"), threads=a.threads)
    print(f"reference FastGA self: {time.time()-t:.1f} s", flush=True)
g = Gdb(ra + ".gdb")
recs, tb, _, _, _ = read_1aln(L, os.path.join(d, "ref.1aln"))
print(f"{len(recs)} alignments, {len(tb)} trace bytes", flush=True)
cache = {}
toff, tlen, diffs, ints = [0], [], [], []
t = time.time()
for r in recs:
    c1, c2, comp = int(r["aread"]), int(r["bread"]), int(r["flags"]) & 1
    if ("a", c1) not in cache:
        cache[("a", c1)] = H.pad_seq(g.contig(c1))
    if ("b", c2, comp) not in cache:
        s = g.contig(c2)
        cache[("b", c2, comp)] = H.pad_seq(synth.revcomp(s) if comp else s)
    tr = tb[int(r["toff"]):int(r["toff"]) + int(r["tlen"])].astype(np.uint16)
    dd, e = H.oracle_trace_pts(cache[("a", c1)], cache[("b", c2, comp)],
                               (int(r["abpos"]), int(r["bbpos"]), int(r["aepos"]), int(r["bepos"]), int(r["diffs"]), tr))
    tlen.append(len(e)); diffs.append(dd); ints.append(e); toff.append(toff[-1] + len(e))
print(f"oracle edit scripts: {time.time()-t:.1f} s, {toff[-1]} indels", flush=True)
arrs = [np.array(toff, np.int64), np.array(tlen, np.int32), np.array(diffs, np.int32), np.concatenate(ints).astype(np.int32)]
A_ = Alns(len(recs), len(tb), 0, 0, recs.ctypes.data, tb.ctypes.data)
for rep in range(2):
    cp = [x.copy() for x in arrs]
    T = Traces(len(recs), int(toff[-1]), 0, *(x.ctypes.data for x in cp))
    t = time.time()
    assert L.fga_gap_improve(g.h, None, C.byref(A_), C.byref(T)) == 0
    print(f"fga_gap_improve (1 thread): {1000*(time.time()-t):.1f} ms, {int((cp[3] != arrs[3]).sum())} entries rewritten", flush=True)
os.environ["FGA_PAF_TIMING"] = "1"
out = os.path.join(d, "out.paf")
for rep in range(3):
    T = Traces(len(recs), int(toff[-1]), 0, *(x.ctypes.data for x in arrs))
    t = time.time()
    assert L.fga_write_paf(out.encode(), g.h, None, C.byref(A_), C.byref(T), 2, a.threads) == 0
    print(f"fga_write_paf -x, {a.threads} threads: {1000*(time.time()-t):.1f} ms, {os.path.getsize(out)/1e6:.0f} MB", flush=True)
# the formatter alone: scripts regrouped beforehand (what fga_trace_pts_regrouped hands over; here by the host instantiation
# of the device routine), resume = -1 everywhere
cp = [x.copy() for x in arrs]
T = Traces(len(recs), int(toff[-1]), 0, *(x.ctypes.data for x in cp), None)
assert L.fga_gap_core_check(g.h, None, C.byref(A_), C.byref(T), 1024, 1 << 20) == 0
for rep in range(3):
    t = time.time()
    assert L.fga_write_paf(out.encode(), g.h, None, C.byref(A_), C.byref(T), 2, a.threads) == 0
    print(f"fga_write_paf -x on regrouped scripts, {a.threads} threads: {1000*(time.time()-t):.1f} ms", flush=True)
