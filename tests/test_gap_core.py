"""fga_gapcore.inc -- the per-alignment routine of the device's Gap_Improver (fga_trace_pts_regrouped) -- instantiated for
the host (fga_gap_core_check) and compared with the pinned restatements, no GPU needed:
  * against oracle/gap_oracle.c (itself pinned call by call against the reference's Gap_Improver) and against the host
    regrouping of fga_paf.c, on the scripts of a reference-made .1aln (pair with both strands, self comparison, and a
    15 %-diverged pair whose boxes hold many gaps);
  * with a scratch too small for most boxes: what the routine hands back (resume index) is finished by the host
    regrouping from there and the result is the same."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import harness as H
from tests.test_aln_writer import _parse_records
from tests.test_paf_writer import _oracle_traces

needs_ref = pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref (real reference build) not present")


def _traces(arrs, n, resume=None):
    from fastga_amd.lib import Traces
    cp = [x.copy() for x in arrs]
    T = Traces(n, len(cp[3]), 0, *(x.ctypes.data for x in cp), resume.ctypes.data if resume is not None else None)
    return T, cp


def _check(L, g1, g2, alns, tb, self_cmp):
    from fastga_amd.lib import Alns
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    h2 = None if self_cmp else g2.h
    _, plain = _oracle_traces(g1, g2, alns, tb)                      # Compute_Trace_PTS
    _, want = _oracle_traces(g1, g2, alns, tb, improve=True)         # ... + Gap_Improver (oracle)
    n = len(alns)
    # the host regrouping (what the formatter threads run)
    T, host = _traces(plain, n)
    assert L.fga_gap_improve(g1.h, h2, C.byref(A), C.byref(T)) == 0, L.fga_last_error()
    assert np.array_equal(host[3], want[3]) and np.array_equal(host[2], want[2])
    changed = int((plain[3] != want[3]).sum())

    # the device routine on the host, roomy scratch: every alignment finished
    T, core = _traces(plain, n)
    assert L.fga_gap_core_check(g1.h, h2, C.byref(A), C.byref(T), 4096, 1 << 20) == 0, L.fga_last_error()
    resume = np.frombuffer((C.c_char * (4 * n)).from_address(T.resume), dtype=np.int32).copy()
    assert (resume == -1).all()
    assert np.array_equal(core[3], want[3]) and np.array_equal(core[2], want[2])

    # a scratch most boxes do not fit: handed back, finished by the host from the resume index
    T2, part = _traces(plain, n)
    assert L.fga_gap_core_check(g1.h, h2, C.byref(A), C.byref(T2), 3, 12) == 0, L.fga_last_error()
    res2 = np.frombuffer((C.c_char * (4 * n)).from_address(T2.resume), dtype=np.int32).copy()
    back = int((res2 >= 0).sum())
    T3, fin = _traces(part, n, resume=res2)
    assert L.fga_gap_improve(g1.h, h2, C.byref(A), C.byref(T3)) == 0, L.fga_last_error()
    assert np.array_equal(fin[3], want[3]) and np.array_equal(fin[2], want[2]) and (res2 == -1).all()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for t in (T, T2):
        libc.free(C.c_void_p(t.resume))
    return changed, back


@needs_ref
@pytest.mark.parametrize("self_cmp", [False, True])
def test_core_equals_the_oracle_on_reference_alignments(toy_pair, tmp_path, built_library, self_cmp):
    from fastga_amd.gixio import Gdb
    d, ra, rb = toy_pair
    w = str(tmp_path)
    H.ref_fastga(ra, None if self_cmp else rb, w, os.path.join(w, "ref"), threads=4)
    alns, tb = _parse_records(H.oneview(os.path.join(w, "ref.1aln")))
    assert len(alns) > 10 and (alns["flags"] & 1).any()
    g1 = Gdb(ra + ".gdb")
    g2 = g1 if self_cmp else Gdb(rb + ".gdb")
    changed, back = _check(built_library, g1, g2, alns, tb, self_cmp)
    assert changed > 0 and back > 0


@needs_ref
def test_core_on_a_diverged_pair_with_crowded_boxes(tmp_path, built_library):
    """15 % divergence: a third of the differences are indels, most of them closer than 50 bases to the next one -- boxes
    of ten and more gaps, tie-breaks between equally far moves, boxes touching each other and the alignment's ends"""
    from fastga_amd import workload
    from fastga_amd.gixio import Gdb
    w = str(tmp_path)
    ra, rb = workload.build_pair(w, seed=5, ncontig=6, total=600_000, divergence=0.15, repeat_frac=0.10, inv_frac=0.10,
                                 swap_frac=0.05)
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=4)
    alns, tb = _parse_records(H.oneview(os.path.join(w, "ref.1aln")))
    assert len(alns) > 5 and (alns["flags"] & 1).any()
    g1, g2 = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    changed, back = _check(built_library, g1, g2, alns, tb, False)
    assert changed > 800 and back > 0
