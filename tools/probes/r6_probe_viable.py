#!/usr/bin/env python3
"""round 6 probe: what fraction of a comparison's seeds lie in diagonal buckets that can reach the chain threshold at all
(2 x sum of lcp over the bucket and its better neighbour >= chain_min)?  The seeds of the others could be dropped before the
sort without changing a hit."""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=100.0)
ap.add_argument("--self", dest="self_", action="store_true")
ap.add_argument("--config4", action="store_true", help="the 3 Gbp pair's recipe (1 % divergence, 45 % repeats) at --mbp")
ap.add_argument("--div", type=float, default=0.01)
ap.add_argument("--cmin", type=int, default=85, help="chain_min: sum of lcp a chain must cover / 2 (FastGA -c)")
a = ap.parse_args()
from fastga_amd import device as D, workload
from fastga_amd.gixio import Gdb
dev = D.Device(0)
d = tempfile.mkdtemp(prefix="fga_viable_")
if a.self_:
    roots = [workload.build_config3(d, mbp=a.mbp, threads=32)]
elif a.config4:
    roots = list(workload.build_config4(d, mbp=a.mbp, divergence=a.div, ncontig=max(2, int(a.mbp / 94)), threads=32))
else:
    roots = list(workload.build_config2(d, mbp=a.mbp, threads=32))
tabs = []
for r in roots:
    g = Gdb(r + ".gdb")
    tabs.append((g,) + tuple(D.build_gix_device(dev, g, 8, use_mask=a.self_)))
gA, dA, A = tabs[0]
gB, dB, B = (gA, None, A) if a.self_ else tabs[1]
s = D.seed_merge(dev, dA, dB, freq=10, soft_mask=a.self_)
h = s.download()
n = len(h)
amx, bmx = int(max(gA.clen)), int(max(gB.clen))
i, j = h["apos"].astype(np.int64), h["bpos"].astype(np.int64)
comp = (h["bctg"] >> 31).astype(np.int64)
actg = (h["actg"] >> 8).astype(np.int64)
bctg = (h["bctg"] & 0x3fffffff).astype(np.int64)
lcp = (h["actg"] & 63).astype(np.int64)
diag = np.where(comp == 1, (amx + bmx) - (i + j), bmx + (i - j))
buck = diag >> 6
pair = (comp * (actg.max() + 1) + actg) * (bctg.max() + 1) + bctg
gid = pair * (1 << 24) + buck              # buckets < 2^24
t = time.time()
order = np.argsort(gid, kind="stable")
g = gid[order]; l = lcp[order]
heads = np.flatnonzero(np.r_[True, g[1:] != g[:-1]])
sums = np.add.reduceat(l, heads)
ug = g[heads]
cnt = np.diff(np.r_[heads, n])
prev_adj = np.r_[False, (ug[1:] == ug[:-1] + 1) & ((ug[1:] & ((1 << 24) - 1)) != 0)]
next_adj = np.r_[prev_adj[1:], False]
with_prev = sums + np.where(prev_adj, np.r_[0, sums[:-1]], 0)
with_next = sums + np.where(next_adj, np.r_[sums[1:], 0], 0)
for cm in (a.cmin, 2 * a.cmin):
    keep = (with_prev >= cm) | (with_next >= cm)
    print(f"sum-of-lcp threshold {cm}: {n} seeds in {len(ug)} buckets; buckets that can reach it (alone or with a neighbour): {keep.sum()} "
          f"({100*keep.mean():.2f} %), their seeds {cnt[keep].sum()} ({100*cnt[keep].sum()/n:.2f} %)", flush=True)
print(f"   ({time.time()-t:.1f} s of numpy)")
