"""The pre-v1.3 genome index layout (separate .post.N files): fga_gix_open turns it into today's
in-memory table.  The old-layout files are written by tests/legacy_gix.py from an index in today's layout; that writer
is itself checked against the reference, whose FastGA reads such files through its old merge threads."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.legacy_gix import write_legacy_index


def _table(L, root):
    X = C.c_void_p()
    assert L.fga_gix_open((root + ".gix").encode(), C.byref(X)) == 0, L.fga_last_error()
    n, eb = L.fga_gix_nents(X), L.fga_gix_ebytes(X)
    tab = np.ctypeslib.as_array(C.cast(L.fga_gix_table(X), C.POINTER(C.c_uint8)), shape=(n * eb,)).copy()
    idx = np.ctypeslib.as_array(C.cast(L.fga_gix_index(X), C.POINTER(C.c_int64)), shape=(1 << 24,)).copy()
    meta = (n, eb, L.fga_gix_postbytes(X), L.fga_gix_contbytes(X), L.fga_gix_nctg(X), L.fga_gix_nparts(X), L.fga_gix_maxpre(X),
            [L.fga_gix_part_begin(X, p) for p in range(L.fga_gix_nparts(X) + 1)],
            np.ctypeslib.as_array(C.cast(L.fga_gix_perm(X), C.POINTER(C.c_int32)), shape=(L.fga_gix_nctg(X),)).tolist())
    L.fga_gix_close(X)
    return tab, idx, meta


def test_legacy_layout_opens_as_the_same_table(toy_pair, family_pair, tmp_path, built_library):
    L = built_library
    for k, (d, ra, rb) in enumerate((toy_pair, family_pair)):
        for r in (ra, rb):
            old = write_legacy_index(r, str(tmp_path / f"old{k}" / os.path.basename(r)))
            t0, i0, m0 = _table(L, r)
            t1, i1, m1 = _table(L, old)
            assert m0 == m1
            assert np.array_equal(i0, i1)
            assert np.array_equal(t0, t1)             # toy genomes are not soft-masked: byte 7 is 0 in both


def test_legacy_layout_errors(toy_pair, tmp_path, built_library):
    L = built_library
    d, ra, rb = toy_pair
    old = write_legacy_index(ra, str(tmp_path / "bad" / "A"))
    os.remove(os.path.join(os.path.dirname(old), ".A.post.1"))
    X = C.c_void_p()
    assert L.fga_gix_open((old + ".gix").encode(), C.byref(X)) != 0
    assert b"position list part" in L.fga_last_error()
    old = write_legacy_index(rb, str(tmp_path / "bad2" / "B"))
    p = os.path.join(os.path.dirname(old), ".B.post.1")
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-4])                      # a truncated position list
    assert L.fga_gix_open((old + ".gix").encode(), C.byref(X)) != 0


def test_reference_reads_the_legacy_files_like_the_new_ones(toy_pair, tmp_path):
    """pins the test writer: the reference's old merge threads on the old-layout files give the alignments its new merge
    threads give on today's files"""
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("reference not built")
    d, ra, rb = toy_pair
    oa = write_legacy_index(ra, str(tmp_path / "old" / "A"))
    ob = write_legacy_index(rb, str(tmp_path / "old" / "B"))
    H.ref_fastga(ra, rb, d, str(tmp_path / "new_out"), threads=4)
    H.ref_fastga(oa, ob, str(tmp_path / "old"), str(tmp_path / "old_out"), threads=4)
    a = H.oneview(str(tmp_path / "new_out") + ".1aln")
    b = H.oneview(str(tmp_path / "old_out") + ".1aln")
    assert len(a) > 100
    assert sorted(a) == sorted(b)
    # a genome against itself: old_self_merge_thread against new_self_merge_thread
    H.ref_fastga(ra, None, d, str(tmp_path / "new_self"), threads=4)
    H.ref_fastga(oa, None, str(tmp_path / "old"), str(tmp_path / "old_self"), threads=4)
    a = H.oneview(str(tmp_path / "new_self") + ".1aln")
    b = H.oneview(str(tmp_path / "old_self") + ".1aln")
    assert len(a) > 20
    assert sorted(a) == sorted(b)


@pytest.mark.gpu
def test_pipeline_on_legacy_index_files_gives_the_same_alignments(toy_pair, tmp_path, built_library):
    """the hot path itself over indices read from the old layout: the .1aln of the run on today's files"""
    from fastga_amd import device as D
    from oracle import harness as H
    d, ra, rb = toy_pair
    oa = write_legacy_index(ra, str(tmp_path / "old" / "A"))
    ob = write_legacy_index(rb, str(tmp_path / "old" / "B"))
    new_out, old_out = str(tmp_path / "new.1aln"), str(tmp_path / "old.1aln")
    st0 = D.run(ra, rb, new_out, nthreads=4)
    st1 = D.run(oa, ob, old_out, nthreads=4)
    assert st0["nseeds"] == st1["nseeds"] and st0["nlive"] == st1["nlive"] > 20
    if H.have_reference():
        assert H.oneview(new_out) == H.oneview(old_out)
    # and a genome against itself (the reference: old_self_merge_thread)
    self_new, self_old = str(tmp_path / "self_new.1aln"), str(tmp_path / "self_old.1aln")
    st2 = D.run(ra, None, self_new, nthreads=4)
    st3 = D.run(oa, None, self_old, nthreads=4)
    assert st2["nseeds"] == st3["nseeds"] > 0 and st2["nlive"] == st3["nlive"]
    if H.have_reference():
        assert H.oneview(self_new) == H.oneview(self_old)
    X = C.c_void_p()
    L = built_library
    assert L.fga_gix_open((oa + ".gix").encode(), C.byref(X)) == 0
    assert L.fga_gix_legacy_cutoff(X) == 255
    L.fga_gix_close(X)
    with pytest.raises(Exception):                       # -f above the cutoff the old index was built with
        D.run(write_legacy_index(ra, str(tmp_path / "old8" / "A"), freq=8), ob, old_out, nthreads=4)


def test_legacy_stub_with_a_bad_prefix_index_is_refused(toy_pair, tmp_path, built_library):
    """the old layout's table is expanded along the stub's 2^24 prefix index: an index that is not a cumulative count
    over the stub's k-mers (a damaged or foreign stub) must be refused, not followed past the table"""
    import numpy as np
    L = built_library
    d, ra, rb = toy_pair
    X = C.c_void_p()
    for k, edit in enumerate((lambda ix: ix.__setitem__(slice(1000, 1010), ix[1000:1010] + 10**7),      # runs past the k-mers
                              lambda ix: ix.__setitem__(5000, ix[4999] - 1 if ix[4999] > 0 else ix[5001] + 1),   # not monotone
                              lambda ix: np.minimum(ix, ix[-1] - 1, out=ix))):                               # ends short of them
        old = write_legacy_index(ra, str(tmp_path / f"stub{k}" / "A"))
        raw = bytearray(open(old + ".gix", "rb").read())
        ix = np.frombuffer(raw, dtype=np.int64, count=1 << 24, offset=16).copy()
        edit(ix)
        raw[16:16 + 8 * (1 << 24)] = ix.tobytes()
        open(old + ".gix", "wb").write(raw)
        assert L.fga_gix_open((old + ".gix").encode(), C.byref(X)) != 0
        msg = L.fga_last_error()        # the last entry is also the k-mer count the table parts are checked against
        assert b"prefix index of the stub" in msg or (k == 2 and b"does not match its stub" in msg), msg
