"""The reference's order for records that tie on (aread, abpos) (fga_order.c; la_sort + la_merge, FastGA.c:3800-3835,
3906-3918): pinned on the CPU against the REAL reference.
  * fga_rmsd_ranges against the Range[] the real rmsd_sort fills (libalign_ref.so),
  * fga_reference_slots + fga_alns_reference_order against the real FastGA's own .1aln: the reference's seed temp files
    (kept alive by the unlink shim) give the per-strand seed counts of the A contigs, the reference's records are put
    into the filter's order (aread, abpos, bread, comp, survival) and must come back in the file's order -- for several
    -T, on a pair whose repeat families put records of both strands on the same (aread, abpos)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import harness as H
from tests.test_aln_reader import read_1aln

needs_ref = pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref (real reference build) not present")


class _Range(C.Structure):
    _fields_ = [("beg", C.c_int), ("end", C.c_int), ("off", C.c_int64)]


def _ranges(L, part, nthreads):
    part = np.ascontiguousarray(part, dtype=np.int64)
    beg, end = np.zeros(nthreads, np.int32), np.zeros(nthreads, np.int32)
    off = np.zeros(nthreads, np.int64)
    n = L.fga_rmsd_ranges(part.ctypes.data_as(C.POINTER(C.c_int64)), len(part), int(part.sum()), nthreads,
                          beg.ctypes.data_as(C.POINTER(C.c_int)), end.ctypes.data_as(C.POINTER(C.c_int)),
                          off.ctypes.data_as(C.POINTER(C.c_int64)))
    return n, beg, end, off


@needs_ref
@pytest.mark.parametrize("seed", range(6))
def test_rmsd_ranges_equal_the_reference_s(built_library, seed):
    R = C.CDLL(os.path.join(H.REF, "libalign_ref.so"))
    R.rmsd_sort.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(seed)
    rsize = int(rng.choice([9, 11, 12]))
    nparts = int(rng.integers(1, 60))
    nthreads = int(rng.integers(1, 17))
    cnt = rng.integers(0, 300, nparts)
    if seed % 2:
        cnt[rng.integers(0, nparts)] = 20000                 # one panel heavier than a thread's share
    cnt[rng.integers(0, nparts, 3)] = 0
    nelem = int(cnt.sum())
    if nelem == 0:
        cnt[0] = 5; nelem = 5
    part = (cnt * rsize).astype(np.int64)
    recs = rng.integers(0, 256, (nelem + 1, rsize), dtype=np.uint8)
    ra = (_Range * nthreads)()
    na = R.rmsd_sort(recs.ctypes.data, nelem, rsize, rsize, nparts, part.ctypes.data, nthreads, C.byref(ra))
    n, beg, end, off = _ranges(built_library, part, nthreads)
    assert n == na
    for t in range(na):
        assert (ra[t].beg, ra[t].end, ra[t].off) == (beg[t], end[t], off[t])


def _slots_py(counts, clen, nthreads, swide):
    """plain restatement of FastGA.c:5057-5086 + RSDsort.c:318-343 + FastGA.c:4336-4345"""
    nctg = len(clen)
    npost = int(sum(clen))
    split = [0]
    r, t, cum = nthreads, npost // nthreads, int(clen[0])
    for x in range(1, nctg):
        if cum >= t and x >= r:
            split.append(x)
            t = (npost * len(split)) // nthreads
            r += nthreads
        cum += int(clen[x])
    split.append(nctg)
    slot = -np.ones((2, nctg), dtype=np.int32)
    for u in range(2):
        for i in range(len(split) - 1):
            c = [int(counts[u][x]) * swide if split[i] <= x < split[i + 1] else 0 for x in range(nctg)]
            asize = sum(c)
            n, thr, s, b = 0, asize // nthreads, 0, None
            for x in range(nctg):
                if c[x] > 0:
                    if b is None:
                        b = x
                    s += c[x]
                    if s >= thr and n < nthreads:
                        for y in range(b, x + 1):
                            if c[y] > 0:
                                slot[u][y] = n
                        n += 1
                        thr = (asize * (n + 1)) // nthreads
                        b = x + 1
    return slot


def _slots(L, counts, clen, nthreads, swide):
    nctg = len(clen)
    cnt = np.ascontiguousarray(np.asarray(counts, dtype=np.int64).reshape(-1))
    cl = np.ascontiguousarray(clen, dtype=np.int64)
    slot = np.zeros(2 * nctg, dtype=np.int32)
    assert L.fga_reference_slots(cnt.ctypes.data_as(C.POINTER(C.c_int64)), cl.ctypes.data_as(C.POINTER(C.c_int64)), nctg,
                                 nthreads, swide, slot.ctypes.data_as(C.POINTER(C.c_int))) == 0, L.fga_last_error()
    return slot.reshape(2, nctg)


@pytest.mark.parametrize("seed", range(8))
def test_reference_slots_follow_the_restatement(built_library, seed):
    rng = np.random.default_rng(100 + seed)
    nctg = int(rng.integers(1, 90))
    nthreads = int(rng.integers(1, 12))
    clen = np.sort(rng.integers(40, 100000, nctg))[::-1]
    counts = rng.integers(0, 5000, (2, nctg))
    counts[:, rng.integers(0, nctg, 4)] = 0
    got = _slots(built_library, counts, clen, nthreads, 13)
    assert np.array_equal(got, _slots_py(counts, clen, nthreads, 13))
    assert ((got >= 0) == (counts > 0)).all() and got.max() < nthreads


def _order_with(L, recs, tb, slot, invp, nctg):
    from fastga_amd.lib import Alns
    recs = np.ascontiguousarray(recs).copy()
    tb = np.ascontiguousarray(tb, dtype=np.uint8).copy()
    # the function re-lays the trace bytes with malloc/free: hand it C-owned memory
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    pa, pt = libc.malloc(recs.nbytes + 64), libc.malloc(len(tb) + 64)
    C.memmove(pa, recs.ctypes.data, recs.nbytes)
    C.memmove(pt, tb.ctypes.data, len(tb))
    A = Alns(len(recs), len(tb), 0, 0, pa, pt)
    s = np.ascontiguousarray(slot.reshape(-1), dtype=np.int32)
    iv = np.ascontiguousarray(invp, dtype=np.int32)
    assert L.fga_alns_reference_order(C.byref(A), s.ctypes.data_as(C.POINTER(C.c_int)),
                                      iv.ctypes.data_as(C.POINTER(C.c_int)), nctg) == 0, L.fga_last_error()
    out = np.frombuffer((C.c_char * recs.nbytes).from_address(A.alns), dtype=recs.dtype).copy()
    otb = np.frombuffer((C.c_char * max(A.ntrace, 1)).from_address(A.tbytes), dtype=np.uint8)[:A.ntrace].copy()
    libc.free.argtypes = [C.c_void_p]
    libc.free(C.c_void_p(A.alns)); libc.free(C.c_void_p(A.tbytes))
    return out, otb


def _traces(recs, tb):
    return [bytes(tb[o:o + n]) for o, n in zip(recs["toff"], recs["tlen"])]


@needs_ref
@pytest.mark.parametrize("threads", [3, 8, 13])
def test_tie_order_of_the_real_reference(tmp_path_factory, built_library, threads):
    from fastga_amd import workload
    from fastga_amd.gixio import Gix, Gdb
    L = built_library
    d = str(tmp_path_factory.mktemp(f"tie{threads}"))
    # repeat families with inverted copies: records of both strands start on the same A position
    ra, rb = workload.build_pair(d, seed=11, ncontig=24, total=1_500_000, divergence=0.02, repeat_frac=0.30, inv_frac=0.20,
                                 swap_frac=0.05, threads=threads)
    _, seeds = H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=threads, capture_seeds=True)
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    ga, gb = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    ipost, icont, jpost, jcont = A.postbytes, A.contbytes, B.postbytes, B.contbytes
    w = 1 + ipost + icont + jpost + jcont
    nctg = len(A.perm)
    counts = np.zeros((2, nctg), dtype=np.int64)
    for u, buf in enumerate(seeds):
        a = np.frombuffer(buf, dtype=np.uint8).reshape(-1, w).astype(np.int64)
        actg = np.zeros(len(a), dtype=np.int64)
        for k in range(icont):
            actg |= a[:, 1 + ipost + k] << (8 * k)
        counts[u] = np.bincount(actg, minlength=nctg)[:nctg]
    assert counts.sum() > 10000
    clen = np.array([int(ga.clen[int(p)]) if p < ga.ncontig else 40 for p in A.perm], dtype=np.int64)
    blen = np.array([int(gb.clen[int(p)]) if p < gb.ncontig else 40 for p in B.perm], dtype=np.int64)
    dbyte, cum = 0, 1
    while cum < int(clen.max()) + int(blen.max()):
        cum *= 256; dbyte += 1
    slot = _slots(L, counts, clen, threads, 2 * dbyte + jcont + 2)
    invp = np.zeros(nctg, dtype=np.int32)
    for j, p in enumerate(A.perm):
        if p < ga.ncontig:
            invp[p] = j

    recs, tb, _, _, _ = read_1aln(L, os.path.join(d, "ref.1aln"))
    comp = (recs["flags"] & 1).astype(np.int64)
    # the filter's order: (aread, abpos, bread, comp), records that tie on all four in the reference's own relative order
    own = np.lexsort((np.arange(len(recs)), comp, recs["bread"], recs["abpos"], recs["aread"]))
    mixed = recs[own]
    key = np.stack([recs["aread"], recs["abpos"]], axis=1)
    tied = (key[1:] == key[:-1]).all(axis=1)
    both = tied & (comp[1:] != comp[:-1])
    assert both.sum() > 5, "the test genome makes no (aread, abpos) ties across strands"
    got, gtb = _order_with(L, mixed, tb, slot, invp, nctg)
    fields = ("tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "flags", "aread", "bread")
    for f in fields:
        assert np.array_equal(got[f], recs[f]), (f, threads)
    assert _traces(got, gtb) == _traces(recs, tb)
    # the same with the trace bytes in record order, as the filter hands them over: permuted inside every moved run's bytes
    packed = mixed.copy()
    pieces = _traces(mixed, tb)
    packed["toff"] = np.concatenate([[0], np.cumsum([len(x) for x in pieces])[:-1]])
    ptb = np.frombuffer(b"".join(pieces), dtype=np.uint8)
    got2, gtb2 = _order_with(L, packed, ptb, slot, invp, nctg)
    for f in fields:
        assert np.array_equal(got2[f], recs[f]), (f, threads)
    assert _traces(got2, gtb2) == _traces(recs, tb)
    assert np.array_equal(got2["toff"], np.concatenate([[0], np.cumsum(got2["tlen"])[:-1]]))
    if threads > 3:
        assert not all(np.array_equal(mixed[f], recs[f]) for f in fields), "every tie was in the filter's order already"
    # idempotent, and a no-op with equal slots
    again, _ = _order_with(L, got, gtb, slot, invp, nctg)
    assert all(np.array_equal(again[f], got[f]) for f in fields)
    same, _ = _order_with(L, mixed, tb, np.zeros_like(slot), invp, nctg)
    assert all(np.array_equal(same[f], mixed[f]) for f in fields)
    A.close(); B.close(); ga.close(); gb.close()
