"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed; nccl = RCCL over xGMI, gloo on CPU).

Phase 1 shards by contiguous ranges of the 2^24 12-mer prefixes, balanced on cost(p) = idx1[p] + idx2[p] exactly like
the reference splits its merge threads on prefix quantiles (FastGA.c:2291-2321).  Panels are independent, so the
union of the per-shard seed sets is the full seed set and no data-path collective is needed; only counts (and, for
phase 2, the seed records keyed by A-contig part -- SURVEY.md 8e) ever cross ranks.
"""
import numpy as np

NPREFIX = 1 << 24


def prefix_shards(idx1, idx2, nshards):
    """[(begin,end)] * nshards covering [0, 2^24), equal cumulative entry count of both tables per shard."""
    total = int(idx1[-1]) + (int(idx2[-1]) if idx2 is not None else 0)
    cuts = [0]
    for s in range(1, nshards):
        target = (total * s) // nshards
        if idx2 is None:
            p = int(np.searchsorted(idx1, target, side="left"))
        else:
            lo, hi = 0, NPREFIX
            while lo < hi:
                m = (lo + hi) >> 1
                if int(idx1[m]) + int(idx2[m]) >= target:
                    hi = m
                else:
                    lo = m + 1
            p = lo
        cuts.append(max(cuts[-1], min(NPREFIX, p + 1)))
    cuts.append(NPREFIX)
    return [(cuts[i], cuts[i + 1]) for i in range(nshards)]


def gather_counts(dist, value, device=None):
    """all-gather one integer per rank (the only cross-rank traffic of bench.py's weak-scaling run)."""
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]
