"""The chain scan pinned to the reference (CPU): hit boxes printed by a DEBUG_HIT build of the real reference
(oracle/_ref/FastGA_hits, FastGA.c:3165-3225) against oracle/chain_oracle.c run on the records the pinned seed oracle
gives for the same index files -- and the product's host scan (fga_chain_scan, closed form) against the oracle, hit
for hit.  tests/test_chain_gpu.py then compares the device scan with this same oracle."""
import os

import numpy as np
import pytest

from oracle import harness as H

needs_hits = pytest.mark.skipif(not os.path.exists(H.ref_bin("FastGA_hits")),
                                reason="oracle/_ref/FastGA_hits (DEBUG_HIT reference build) not present")


def _records(A, B, ga, gb, self_cmp=False, freq=10, symmetric=False):
    """sorted record fields of the run from the pinned seed oracle (tests/test_oracle_vs_reference.py)"""
    amx, bmx = int(ga.maxctg), int(gb.maxctg)
    if self_cmp:
        n, c, _, _ = H.oracle_self_seed_merge(A.table, A.index, A.pbyte, freq=freq)
    else:
        n, c, _, _ = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte, freq=freq)
        if symmetric:
            n2, c2, _, _ = H.oracle_seed_merge(B.table, B.index, B.pbyte, A.table, A.index, A.pbyte, freq=freq,
                                               flip=True)
            n, c = n + n2, c + c2
    return H.records_from_seed_bytes(n, c, A.postbytes, A.contbytes, B.postbytes, B.contbytes, amx, bmx), amx, bmx


def _check(ra, rb, workdir, built_library, flags=(), **kw):
    from fastga_amd.gixio import Gix, Gdb
    from fastga_amd import device as D
    self_cmp = rb is None
    A, ga = Gix(ra + ".gix"), Gdb(ra + ".gdb")
    B, gb = (A, ga) if self_cmp else (Gix(rb + ".gix"), Gdb(rb + ".gdb"))
    f, amx, bmx = _records(A, B, ga, gb, self_cmp=self_cmp, **kw)
    alen_sorted = ga.clen[A.perm]
    cmin = 2 * 85
    rows = H.oracle_chain_scan(f, 2000, cmin, amx, bmx, alen_sorted)
    # -- oracle vs the reference's own boxes (original contig indices there; multiset: the reference prints part by part)
    ref = H.ref_hit_boxes(ra, rb, workdir, threads=4, flags=flags)
    ours = sorted((int(A.perm[r[1]]), int(B.perm[r[2]]), int(r[3]), int(r[4]), int(r[5]), int(r[6]), int(r[7]),
                   int(r[8]), int(r[9])) for r in rows)
    assert len(ref) > 10
    assert sorted(ref) == ours
    # -- the product's host scan (closed form) == the oracle, in order
    bits = lambda v: max(1, int(v).bit_length())       # noqa: E731
    wa, wb, wt = bits(A.nctg - 1), bits(B.nctg - 1), bits(amx + bmx)
    wd = bits((amx + bmx) >> 6)
    keys = H.pack_keys(f, wa, wb, wd, wt)
    hl = D.chain_scan(keys, (wa, wb, wd, wt), 2000, cmin, amx, bmx, alen_sorted, nthreads=3)
    u, h = hl.units, hl.hits
    got = []
    for x in u:
        for q in range(int(x["nhits"])):
            y = h[int(x["first_hit"]) + q]
            got.append((int(x["comp"]), int(x["actg"]), int(x["bctg"]), int(x["bucket"]), int(y["cov"]),
                        int(y["dgmin"]), int(y["dgmax"]), int(y["alow"]), int(y["ahgh"])))
    exp = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[5]), int(r[6]), int(r[7]), int(r[8]), int(r[9]))
           for r in rows]
    assert got == exp
    hl.free()
    return len(rows)


@needs_hits
def test_chain_oracle_equals_reference_hit_boxes_pair(toy_pair, tmp_path, built_library):
    d, ra, rb = toy_pair
    assert _check(ra, rb, str(tmp_path), built_library) > 20


@needs_hits
def test_chain_oracle_equals_reference_hit_boxes_self_and_symmetric(toy_pair, tmp_path, built_library):
    d, ra, rb = toy_pair
    _check(ra, None, str(tmp_path), built_library)
    _check(ra, rb, str(tmp_path), built_library, flags=("-S", "-f6"), freq=6, symmetric=True)
