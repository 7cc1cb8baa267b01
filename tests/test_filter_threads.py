"""Host filter (fga_filter_alignments_mt): the result does not depend on the thread count or on the order the
alignments arrive in -- random sets of overlapping alignments over many contig pairs, with consistent trace bytes."""
import ctypes as C

import numpy as np


def _random_set(rng, npairs, per_pair, nctg=40):
    from fastga_amd.device import ALN_DTYPE
    recs, tb, off, unit = [], [], 0, 0
    for pair in range(npairs):
        a, b, comp = int(rng.integers(0, nctg)), int(rng.integers(0, 40)), int(rng.integers(0, 2))
        seq = 0
        for _ in range(int(rng.integers(1, per_pair + 1))):
            ab = int(rng.integers(0, 200_000))
            ln = int(rng.integers(150, 5000))
            if recs and rng.random() < 0.3 and recs[-1][7] == a:       # near-duplicates and overlaps of the previous one
                ab = recs[-1][2] + int(rng.integers(0, 3)) * 100
            bb = ab + int(rng.integers(-50, 50)) + 1000
            ae = ab + ln
            npan = ae // 100 - ab // 100 + (1 if ae % 100 else 0)
            npan = max(npan, 1)
            t = np.empty(2 * npan, dtype=np.uint8)
            t[0::2] = rng.integers(0, 9, npan)
            t[1::2] = 100
            first = 100 - ab % 100 if npan > 1 else ln
            last = ae - (ab // 100 + npan - 1) * 100 if npan > 1 else ln
            t[1] = first
            t[-1] = last
            be = bb + int(t[1::2].sum())
            recs.append((2 * npan, int(t[0::2].sum()), ab, bb, ae, be, comp, a, b, unit, seq, 0, off))
            tb.append(t)
            off += 2 * npan
            seq += 1
        unit += 1
    arr = np.array(recs, dtype=ALN_DTYPE)
    return arr, np.concatenate(tb)


def _run(L, alns, tb, nthreads):
    from fastga_amd.lib import Alns
    from fastga_amd.device import ALN_DTYPE
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    out = C.POINTER(Alns)()
    assert L.fga_filter_alignments_mt(C.byref(A), nthreads, C.byref(out)) == 0
    o = out.contents
    n, nt = o.naln, o.ntrace
    a = np.frombuffer((C.c_char * (n * ALN_DTYPE.itemsize)).from_address(o.alns), dtype=ALN_DTYPE).copy()
    t = np.frombuffer((C.c_char * max(nt, 1)).from_address(o.tbytes), dtype=np.uint8)[:nt].copy()
    L.fga_alns_free(out)
    return a, t


def test_filter_is_thread_and_arrival_order_independent(built_library):
    L = built_library
    rng = np.random.default_rng(77)
    alns, tb = _random_set(rng, 16000, 6)      # > 50 k alignments: the threaded path
    assert len(alns) > 50000
    a1, t1 = _run(L, alns, tb, 1)
    assert 0 < len(a1) < len(alns)
    for nt in (2, 8, 32):
        a, t = _run(L, alns, tb, nt)
        assert np.array_equal(a, a1) and np.array_equal(t, t1)
    perm = rng.permutation(len(alns))
    a, t = _run(L, alns[perm], tb, 8)
    assert np.array_equal(a, a1) and np.array_equal(t, t1)
    key = list(zip(a1["aread"].tolist(), a1["abpos"].tolist(), a1["bread"].tolist(), (a1["flags"] & 1).tolist()))
    assert key == sorted(key)
    small, tbs = _random_set(rng, 5, 3)                        # tiny sets take the same code path
    s1 = _run(L, small, tbs, 1)
    s8 = _run(L, small, tbs, 8)
    assert np.array_equal(s1[0], s8[0]) and np.array_equal(s1[1], s8[1])


def test_final_order_merge_does_not_depend_on_its_task_cuts(built_library):
    """the final order is a merge of every A contig's segment lists, cut by abpos splitters into tasks: two A contigs
    with ~30 k survivors each, many records on the same (aread, abpos) in different segments -- one task per contig,
    tasks of 64 records, the default cut and several thread counts must give the same bytes, in sorted order"""
    import os
    L = built_library
    rng = np.random.default_rng(5)
    alns, tb = _random_set(rng, 12000, 8, nctg=2)
    alns["abpos"] -= alns["abpos"] % 100                         # ties on abpos across segments
    alns["aepos"] = np.maximum(alns["aepos"], alns["abpos"] + 100)
    assert len(alns) > 50000
    res = []
    for chunk, nt in ((None, 8), (10**9, 1), (64, 8), (64, 3), (1000, 32)):
        if chunk is None:
            os.environ.pop("FGA_FILTER_CHUNK", None)
        else:
            os.environ["FGA_FILTER_CHUNK"] = str(chunk)
        try:
            res.append(_run(L, alns, tb, nt))
        finally:
            os.environ.pop("FGA_FILTER_CHUNK", None)
    a1, t1 = res[0]
    assert len(a1) > 20000
    for a, t in res[1:]:
        assert np.array_equal(a, a1) and np.array_equal(t, t1)
    key = list(zip(a1["aread"].tolist(), a1["abpos"].tolist(), a1["bread"].tolist(), (a1["flags"] & 1).tolist()))
    assert key == sorted(key)
    ties = sum(1 for x, y in zip(key, key[1:]) if x[:2] == y[:2] and x[2:] != y[2:])
    assert ties > 1000
