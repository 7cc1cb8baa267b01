"""GPU: the exact-signature shims of the reference's module seams (include/fastga_amd.h), called through
the C-ABI beside the REAL reference functions (oracle/_ref/libalign_ref.so = align.c + RSDsort.c compiled as they are) on
the same inputs: fga_shim_Local_Alignment vs Local_Alignment (align.h:235-236) call by call -- Path fields and trace --
and fga_shim_rmsd_sort vs rmsd_sort (RSDsort.c:292) -- sorted bytes, thread ranges, return value."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import harness as H

pytestmark = pytest.mark.gpu

needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(H.REF, "libalign_ref.so")),
                               reason="oracle/_ref/libalign_ref.so did not travel")


class ShimAligner:
    """fga_shim_* with the calling sequence of the reference (cf. oracle.harness.RefAligner)"""

    def __init__(self, L, ave_corr=0.7, tspace=100, freq=(0.25, 0.25, 0.25, 0.25), reach=0):
        self.L = L
        f = (C.c_float * 4)(*freq)
        self.spec = L.fga_shim_New_Align_Spec(ave_corr, tspace, f, reach)
        self.work = L.fga_shim_New_Work_Data()
        assert self.spec and self.work, L.fga_last_error()

    def align(self, abuf, bbuf, low, hgh, anti, lbord=-1, hbord=-1, acomp=False, selfie=False):
        path = H._RPath()
        al = H._RAlign()
        al.path = C.pointer(path)
        al.flags = 2 if acomp else 0
        al.aseq = abuf.ctypes.data + 1
        al.bseq = (abuf.ctypes.data + 1) if selfie else (bbuf.ctypes.data + 1)
        al.alen = len(abuf) - 2
        al.blen = len(bbuf) - 2
        st = self.L.fga_shim_Local_Alignment(C.byref(al), self.work, self.spec, low, hgh, anti, lbord, hbord)
        assert st == 0, self.L.fga_last_error()
        n = path.tlen
        tr = np.ctypeslib.as_array(C.cast(path.trace, C.POINTER(C.c_uint16)), shape=(max(n, 1),))[:n].copy()
        return (path.abpos, path.bbpos, path.aepos, path.bepos, path.diffs, tr)

    def close(self):
        self.L.fga_shim_Free_Work_Data(self.work)
        self.L.fga_shim_Free_Align_Spec(self.spec)


@needs_ref
def test_local_alignment_shim_equals_reference_call_by_call(built_library):
    from fastga_amd import synth
    from tests.test_oracle_vs_reference import _random_case
    rng = np.random.default_rng(20260926)
    ref, ours = H.RefAligner(), ShimAligner(built_library)
    found = 0
    for _ in range(200):
        A, B, acomp, low, hgh, anti, lb, hb = _random_case(rng)
        abuf, bbuf = H.pad_seq(A), H.pad_seq(B)
        r = ref.align(abuf, bbuf, low, hgh, anti, lb, hb, acomp)
        o = ours.align(abuf, bbuf, low, hgh, anti, lb, hb, acomp)
        assert r[:5] == o[:5], (r[:5], o[:5], acomp, low, hgh, anti, lb, hb)
        assert np.array_equal(r[5], o[5])
        found += r[2] > r[0]
    assert found > 100
    # a long alignment (thousands of wave steps, arena levels beyond the first) and the aseq == bseq rule
    A = rng.integers(0, 4, 400_000, dtype=np.uint8)
    B = synth.mutate(rng, A, 0.03)
    abuf, bbuf = H.pad_seq(A), H.pad_seq(B)
    r = ref.align(abuf, bbuf, -40, 40, 2 * 200_000)
    o = ours.align(abuf, bbuf, -40, 40, 2 * 200_000)
    assert r[:5] == o[:5] and np.array_equal(r[5], o[5]) and r[2] - r[0] > 300_000
    S = rng.integers(0, 4, 40_000, dtype=np.uint8)
    S[21_000:26_000] = synth.mutate(rng, S[1000:6000], 0.05)[:5000]
    sbuf = H.pad_seq(S)
    for low, hgh in ((-20_040, -19_960), (19_960, 20_040)):
        r = ref.align(sbuf, sbuf, low, hgh, 2 * 3500 + (20_000 if low < 0 else -20_000) + 2 * 20_000 * (low > 0), selfie=True)
        o = ours.align(sbuf, sbuf, low, hgh, 2 * 3500 + (20_000 if low < 0 else -20_000) + 2 * 20_000 * (low > 0), selfie=True)
        assert r[:5] == o[:5] and np.array_equal(r[5], o[5])
    ref.close(); ours.close()


class _Range(C.Structure):
    _fields_ = [("beg", C.c_int), ("end", C.c_int), ("off", C.c_int64)]


@needs_ref
@pytest.mark.parametrize("rsize,ksize", [(9, 9), (12, 12), (11, 9)])
def test_rmsd_sort_shim_equals_reference(built_library, rsize, ksize):
    R = C.CDLL(os.path.join(H.REF, "libalign_ref.so"))
    rng = np.random.default_rng(rsize * 100 + ksize)
    nparts, nthreads = 37, 6
    cnt = rng.integers(0, 4000, nparts)
    cnt[rng.integers(0, nparts, 5)] = 0                       # empty panels
    nelem = int(cnt.sum())
    part = (cnt * rsize).astype(np.int64)
    recs = rng.integers(0, 256, (nelem + 1, rsize), dtype=np.uint8)       # + 1 record of slack (FastGA.c:4190)
    recs[:, rsize - 1] = rng.integers(0, 3, nelem + 1)                    # few distinct top bytes: deep radix levels
    recs[:nelem // 3, rsize - 2] = 7
    if ksize < rsize:                                                      # bytes outside the key: equal, so any order of ties is the same
        recs[:, :rsize - ksize] = 0
    a = np.ascontiguousarray(recs).copy()
    b = a.copy()
    ra, rb = (_Range * nthreads)(), (_Range * nthreads)()
    R.rmsd_sort.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    na = R.rmsd_sort(a.ctypes.data, nelem, rsize, ksize, nparts, part.ctypes.data, nthreads, C.byref(ra))
    nb = built_library.fga_shim_rmsd_sort(b.ctypes.data, nelem, rsize, ksize, nparts,
                                          part.ctypes.data_as(C.POINTER(C.c_int64)), nthreads, C.byref(rb))
    assert nb == na > 0, built_library.fga_last_error()
    assert np.array_equal(a[:nelem], b[:nelem])
    for t in range(na):
        assert (ra[t].beg, ra[t].end, ra[t].off) == (rb[t].beg, rb[t].end, rb[t].off)
    # really sorted, panel by panel, as little-endian integers of the key bytes
    off = 0
    for p in range(nparts):
        blk = b[off:off + int(cnt[p])]
        keys = [int.from_bytes(bytes(r[rsize - ksize:]), "little") for r in blk]
        assert keys == sorted(keys)
        off += int(cnt[p])


@needs_ref
def test_compute_trace_pts_and_gap_improver_shims_equal_the_reference_call_by_call(built_library):
    """fga_shim_Compute_Trace_PTS / fga_shim_Gap_Improver (align.h:266-267, 399) beside the real functions of
    libalign_ref.so on the alignments the real Local_Alignment finds: the int edit script, its length and the difference
    count after each of the two calls, as the readers of a .1aln make them (ALNtoPAF.c:278-280)."""
    from fastga_amd import synth
    from tests.test_oracle_vs_reference import _random_case
    L = built_library
    rng = np.random.default_rng(20260927)
    ref = H.RefAligner()
    work = L.fga_shim_New_Work_Data()
    assert work, L.fga_last_error()

    def ours(abuf, bbuf, path):
        abpos, bbpos, aepos, bepos, diffs, tr = path
        pts = np.ascontiguousarray(tr, dtype=np.uint16).copy()
        rp = H._RPath()
        rp.trace = pts.ctypes.data
        rp.tlen, rp.diffs = len(pts), diffs
        rp.abpos, rp.bbpos, rp.aepos, rp.bepos = abpos, bbpos, aepos, bepos
        al = H._RAlign()
        al.path = C.pointer(rp)
        al.flags = 0
        al.aseq, al.bseq = abuf.ctypes.data + 1, bbuf.ctypes.data + 1
        al.alen, al.blen = len(abuf) - 2, len(bbuf) - 2
        assert L.fga_shim_Compute_Trace_PTS(C.byref(al), work, 100, 0, 1, -1) == 0, L.fga_last_error()
        n = rp.tlen
        t1 = np.ctypeslib.as_array(C.cast(rp.trace, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
        d1 = rp.diffs
        assert L.fga_shim_Gap_Improver(C.byref(al), work) == 0, L.fga_last_error()
        assert rp.tlen == n
        t2 = np.ctypeslib.as_array(C.cast(rp.trace, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
        return (d1, t1), (rp.diffs, t2)

    done = 0
    cases = [_random_case(rng) for _ in range(120)]
    A = rng.integers(0, 4, 60_000, dtype=np.uint8)                 # a long one: hundreds of trace panels, many indels
    cases.append((A, synth.mutate(rng, A, 0.06), False, -30, 30, 2 * 30_000, -1, -1))
    for A, B, acomp, low, hgh, anti, lb, hb in cases:
        if acomp:
            continue                                               # the readers complement B themselves; A is never complemented
        abuf, bbuf = H.pad_seq(A), H.pad_seq(B)
        path = ref.align(abuf, bbuf, low, hgh, anti, lb, hb, False)
        if path[2] - path[0] < 50 or len(path[5]) == 0 or int(path[5].max()) > 255:
            continue
        r1 = ref.trace_pts(abuf, bbuf, path)
        r2 = ref.trace_pts(abuf, bbuf, path, improve=True)
        o1, o2 = ours(abuf, bbuf, path)
        assert r1[0] == o1[0] and np.array_equal(r1[1], o1[1]), (path[:5], r1[0], o1[0])
        assert r2[0] == o2[0] and np.array_equal(r2[1], o2[1]), (path[:5], r2[0], o2[0])
        done += 1
    assert done > 40
    # what the shim refuses, it refuses loudly: another spacing, another mode, a band
    rp = H._RPath(); al = H._RAlign(); al.path = C.pointer(rp)
    buf = H.pad_seq(rng.integers(0, 4, 500, dtype=np.uint8))
    al.aseq = al.bseq = buf.ctypes.data + 1
    al.alen = al.blen = 500
    for args in ((50, 0, 1, -1), (100, 1, 1, -1), (100, 0, -5, 5)):
        assert L.fga_shim_Compute_Trace_PTS(C.byref(al), work, *args) == 1 and b"only trace spacing 100" in L.fga_last_error()
    L.fga_shim_Free_Work_Data(work)
    ref.close()
