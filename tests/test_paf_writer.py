"""PAF output (host C, no GPU needed): fga_write_paf / fga_gap_improve against the reference's ALNtoPAF on a
reference-produced .1aln.  The edit scripts the writer needs come from the device stage in the product; here they come
from the CPU oracle of that stage (oracle/trace_oracle.c, pinned against Compute_Trace_PTS), so that this test checks
the host side alone: gap regrouping (Gap_Improver), operations, tags, coordinates, every option of ALNtoPAF."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import harness as H
from tests.test_aln_writer import _parse_records

needs_ref = pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref (real reference build) not present")

FLAGS = {"m": 1, "x": 2, "s": 4, "S": 8, "w": 16}


def _oracle_traces(g1, g2, alns, tb, improve=False):
    from fastga_amd import synth
    from fastga_amd.lib import Traces
    toff, tlen, diffs, ints = [0], [], [], []
    cache = {}
    for a in alns:
        c1, c2, comp = int(a["aread"]), int(a["bread"]), int(a["flags"]) & 1
        if ("a", c1) not in cache:
            cache[("a", c1)] = H.pad_seq(g1.contig(c1))
        if ("b", c2, comp) not in cache:
            s = g2.contig(c2)
            cache[("b", c2, comp)] = H.pad_seq(synth.revcomp(s) if comp else s)
        t = tb[int(a["toff"]):int(a["toff"]) + int(a["tlen"])].astype(np.uint16)
        path = (int(a["abpos"]), int(a["bbpos"]), int(a["aepos"]), int(a["bepos"]), int(a["diffs"]), t)
        d, tr = H.oracle_trace_pts(cache[("a", c1)], cache[("b", c2, comp)], path)
        if improve:                       # the reference hands Gap_Improver only the aligned piece of B (ALNtoPAF.c:270)
            piece = H.pad_seq(cache[("b", c2, comp)][1 + path[1]:1 + path[3]])
            shifted = (path[0], 0, path[2], path[3] - path[1], path[4], t)
            rel = np.where(tr > 0, tr - path[1], tr)
            d, rel = H.oracle_gap_improver(cache[("a", c1)], piece, shifted, rel, d)
            tr = np.where(rel > 0, rel + path[1], rel).astype(np.int32)
        tlen.append(len(tr)); diffs.append(d); ints.append(tr)
        toff.append(toff[-1] + len(tr))
    arrs = (np.array(toff, np.int64), np.array(tlen, np.int32), np.array(diffs, np.int32),
            np.concatenate(ints).astype(np.int32) if ints else np.zeros(0, np.int32))
    T = Traces(len(alns), int(toff[-1]), 0, *(x.ctypes.data for x in arrs))
    return T, arrs


@needs_ref
@pytest.mark.parametrize("self_cmp", [False, True])
def test_paf_matches_alntopaf_for_every_option(toy_pair, tmp_path, built_library, self_cmp):
    from fastga_amd.lib import load_library, Alns
    from fastga_amd.gixio import Gdb
    d, ra, rb = toy_pair
    w = str(tmp_path)
    H.ref_fastga(ra, None if self_cmp else rb, w, os.path.join(w, "ref"), threads=4)
    ref = os.path.join(w, "ref.1aln")
    alns, tb = _parse_records(H.oneview(ref))
    assert len(alns) > 10 and (alns["flags"] & 1).any()
    L = load_library()
    g1 = Gdb(ra + ".gdb")
    g2 = g1 if self_cmp else Gdb(rb + ".gdb")
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    T, keep = _oracle_traces(g1, g2, alns, tb)
    out = os.path.join(w, "ours.paf")
    for opts in ("", "m", "x", "s", "S", "ms", "xS", "mS", "xs", "w", "wx", "wm", "wS", "wxs"):
        flags = sum(FLAGS[c] for c in opts)
        rc = L.fga_write_paf(out.encode(), g1.h, None if self_cmp else g2.h, C.byref(A),
                             C.byref(T) if flags & 15 else None, flags, 3)
        assert rc == 0, L.fga_last_error()
        exp = H.run([H.ref_bin("ALNtoPAF"), "-T2"] + (["-" + opts] if opts else []) + [ref], cwd=w).stdout
        got = open(out).read()
        assert got.count("\n") == len(alns)
        assert got == exp, opts
    # PSL (the reference's ALNtoPSL): counts, ranges and block lists
    rc = L.fga_write_psl(out.encode(), g1.h, None if self_cmp else g2.h, C.byref(A), C.byref(T), 3)
    assert rc == 0, L.fga_last_error()
    exp = H.run([H.ref_bin("ALNtoPSL"), "-T2", ref], cwd=w).stdout
    got = open(out).read()
    assert got.count("\n") == len(alns) and got == exp
    assert L.fga_write_psl(out.encode(), g1.h, None, C.byref(A), None, 1) != 0
    # thread count does not change the file; conflicting or incomplete requests are refused
    L.fga_write_paf(out.encode(), g1.h, None if self_cmp else g2.h, C.byref(A), C.byref(T), 2 | 8, 1)
    one = open(out).read()
    L.fga_write_paf(out.encode(), g1.h, None if self_cmp else g2.h, C.byref(A), C.byref(T), 2 | 8, 7)
    assert open(out).read() == one
    assert L.fga_write_paf(out.encode(), g1.h, None, C.byref(A), C.byref(T), 1 | 2, 1) != 0
    assert L.fga_write_paf(out.encode(), g1.h, None, C.byref(A), None, 2, 1) != 0
    assert b"edit scripts" in L.fga_last_error()
    g1.close()
    if not self_cmp:
        g2.close()


@needs_ref
def test_gap_improve_matches_oracle(toy_pair, tmp_path, built_library):
    from fastga_amd.lib import load_library, Alns
    from fastga_amd.gixio import Gdb
    d, ra, rb = toy_pair
    w = str(tmp_path)
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=4)
    alns, tb = _parse_records(H.oneview(os.path.join(w, "ref.1aln")))
    L = load_library()
    g1, g2 = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    T, (toff, tlen, diffs, ints) = _oracle_traces(g1, g2, alns, tb)
    before = ints.copy()
    _, (toff2, tlen2, diffs2, ints2) = _oracle_traces(g1, g2, alns, tb, improve=True)
    assert L.fga_gap_improve(g1.h, g2.h, C.byref(A), C.byref(T)) == 0
    assert np.array_equal(ints, ints2) and np.array_equal(diffs, diffs2)
    assert not np.array_equal(before, ints)            # boxes were found and rewritten
    g1.close(); g2.close()
