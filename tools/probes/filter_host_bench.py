#!/usr/bin/env python3
"""tools/filter_host_bench.py -- the host filter's O(n) passes at the record counts of a 3 Gbp part, without a GPU: a
synthetic raw set (units in key order, records of a unit in sequence, alignments that do not touch one another, so that
nearly all survive) through fga_filter_alignments_mt with FGA_FILTER_TIMING=1.  The elimination itself depends on the
data; the ordering and copying passes do not."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastga_amd.lib import load_library, Alns
from fastga_amd.device import ALN_DTYPE

ap = argparse.ArgumentParser()
ap.add_argument("--records", type=int, default=2_300_000)
ap.add_argument("--contigs", type=int, default=32)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--tlen", type=int, default=40)
a = ap.parse_args()
rng = np.random.default_rng(1)
n, nc = a.records, a.contigs
L = load_library()
# units: one per (comp, aread, bread, bucket) in key order; ~1.1 records per unit
seg = rng.integers(0, 2 * nc * nc, n)
seg.sort()
comp, aread, bread = seg // (nc * nc), (seg // nc) % nc, seg % nc
recs = np.zeros(n, dtype=ALN_DTYPE)
recs["aread"], recs["bread"], recs["flags"] = aread, bread, comp
recs["unit"] = np.arange(n) // 2
recs["seq"] = np.arange(n) % 2
ab = rng.integers(0, 90_000_000, n)
recs["abpos"], recs["aepos"] = ab, ab + 1500
bb = rng.integers(0, 90_000_000, n)
recs["bbpos"], recs["bepos"] = bb, bb + 1500
recs["tlen"] = a.tlen
recs["diffs"] = 30
recs["toff"] = np.arange(n, dtype=np.int64) * a.tlen
tb = np.full(n * a.tlen, 50, dtype=np.uint8)
tb[0::2] = 1
perm = rng.permutation(n)                     # the kernel's output order is arbitrary
recs = recs[perm].copy()
A = Alns(n, len(tb), 0, 0, recs.ctypes.data, tb.ctypes.data)
os.environ["FGA_FILTER_TIMING"] = "1"
for rep in range(3):
    out = C.POINTER(Alns)()
    t = time.time()
    assert L.fga_filter_alignments_mt(C.byref(A), a.threads, C.byref(out)) == 0, L.fga_last_error()
    print(f"fga_filter_alignments_mt, {a.threads} threads: {1000*(time.time()-t):.1f} ms, {out.contents.naln} of {n} survive", flush=True)
    L.fga_alns_free(out)
