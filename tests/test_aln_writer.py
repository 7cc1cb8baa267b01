"""The .1aln writers (host C, no GPU needed): a reference-produced .1aln is parsed from its ONEview text, written
again by fga_write_1aln_binary / fga_write_1aln, and the reference's own tools must see the same file:
ONEview text identical (minus provenance/path lines), and ALNtoPAF -- which seeks through the binary object index
(oneGoto) -- prints the same PAF, plain and with -x (re-alignment between trace points through the GDB)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import harness as H

needs_ref = pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref (real reference build) not present")


def _parse_records(lines):
    """A/R/D/T/X lines of ONEview text -> (ALN_DTYPE array, trace bytes) in the interleaved trace layout of fga_aln"""
    from fastga_amd.device import ALN_DTYPE
    recs, tb = [], []
    cur = None
    off = 0
    for ln in lines:
        t = ln[0]
        f = ln.split()
        if t == "A":
            cur = dict(aread=int(f[1]), abpos=int(f[2]), aepos=int(f[3]), bread=int(f[4]), bbpos=int(f[5]),
                       bepos=int(f[6]), flags=0, diffs=0, T=[], X=[])
            recs.append(cur)
        elif t == "R":
            cur["flags"] = 1
        elif t == "D":
            cur["diffs"] = int(f[1])
        elif t == "T":
            cur["T"] = [int(x) for x in f[2:]]
        elif t == "X":
            cur["X"] = [int(x) for x in f[2:]]
    arr = np.zeros(len(recs), dtype=ALN_DTYPE)
    for i, r in enumerate(recs):
        n = len(r["T"])
        assert n == len(r["X"])
        inter = np.empty(2 * n, dtype=np.uint8)
        inter[0::2] = r["X"]
        inter[1::2] = r["T"]
        tb.append(inter)
        arr[i] = (2 * n, r["diffs"], r["abpos"], r["bbpos"], r["aepos"], r["bepos"], r["flags"], r["aread"],
                  r["bread"], -1, i, 0, off)
        off += 2 * n
    return arr, (np.concatenate(tb) if tb else np.zeros(0, np.uint8))


@needs_ref
@pytest.mark.parametrize("self_cmp", [False, True])
def test_writers_round_trip_through_reference_tools(toy_pair, tmp_path, built_library, self_cmp):
    from fastga_amd.lib import load_library, Alns
    from fastga_amd.gixio import Gdb
    d, ra, rb = toy_pair
    w = str(tmp_path)
    H.ref_fastga(ra, None if self_cmp else rb, w, os.path.join(w, "ref"), threads=4)
    ref = os.path.join(w, "ref.1aln")
    txt = H.oneview(ref)
    alns, tb = _parse_records(txt)
    assert len(alns) > 10
    L = load_library()
    g1 = Gdb(ra + ".gdb")
    g2 = None if self_cmp else Gdb(rb + ".gdb")
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]          # noqa: E731
    for name, fn in (("bin", L.fga_write_1aln_binary), ("txt", L.fga_write_1aln)):
        out = os.path.join(w, name + ".1aln")
        rc = fn(out.encode(), g1.h, g2.h if g2 is not None else None, C.byref(A), 100, (ra + ".gdb").encode(),
                None if self_cmp else (rb + ".gdb").encode(), b"FastGA test")
        assert rc == 0
        assert keep(H.oneview(out)) == keep(txt), name
    head = open(os.path.join(w, "bin.1aln"), "rb").read(4096)
    assert b"\n$ 0\n" in head                                                # really the binary container
    for flags in ((), ("-x",)):
        a = H.run([H.ref_bin("ALNtoPAF"), "-T3", *flags, os.path.join(w, "bin.1aln")], cwd=w).stdout
        b = H.run([H.ref_bin("ALNtoPAF"), "-T3", *flags, ref], cwd=w).stdout
        assert len(a.splitlines()) == len(alns) and a == b, flags
    g1.close()
    if g2 is not None:
        g2.close()


@needs_ref
def test_binary_writer_on_many_threads_writes_the_same_file(toy_pair, tmp_path, built_library):
    """the binary container's records are formatted by 1 / 8 / 32 / 64 threads (sets beyond 200 k trace points): the same
    bytes apart from the provenance line's time stamp, and the reference's ONEview reads all of it back -- 60 k synthetic
    records with traces of 2-40 panels inside the toy pair's contigs"""
    from fastga_amd.lib import Alns
    from fastga_amd.device import ALN_DTYPE
    from fastga_amd.gixio import Gdb
    from tests.test_aln_reader import read_1aln
    L = built_library
    d, ra, rb = toy_pair
    g1, g2 = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    rng = np.random.default_rng(12)
    n = 60_000
    recs = np.zeros(n, dtype=ALN_DTYPE)
    recs["aread"] = np.sort(rng.integers(0, g1.ncontig, n))
    recs["bread"] = rng.integers(0, g2.ncontig, n)
    recs["flags"] = rng.integers(0, 2, n)
    npan = rng.integers(1, 21, n)
    recs["tlen"] = 2 * npan
    pieces, off = [], 0
    for i in range(n):
        la, lb = int(g1.clen[recs["aread"][i]]), int(g2.clen[recs["bread"][i]])
        k = int(min(npan[i], la // 100 - 2, lb // 110 - 2))
        k = max(k, 1)
        recs["tlen"][i] = 2 * k
        ab = 100 * int(rng.integers(0, la // 100 - k))
        bb = int(rng.integers(0, lb - 110 * k))
        t = np.empty(2 * k, dtype=np.uint8)
        t[0::2] = rng.integers(0, 12, k)
        t[1::2] = rng.integers(95, 106, k)
        recs["abpos"][i], recs["aepos"][i] = ab, ab + 100 * k
        recs["bbpos"][i], recs["bepos"][i] = bb, bb + int(t[1::2].sum())
        recs["diffs"][i] = int(t[0::2].sum())
        recs["toff"][i] = off
        pieces.append(t)
        off += 2 * k
    recs["unit"], recs["seq"] = -1, np.arange(n)
    tb = np.concatenate(pieces)
    assert len(tb) // 2 > 200_000
    A = Alns(n, len(tb), 0, 0, recs.ctypes.data, tb.ctypes.data)
    files = []
    try:
        for nt in (1, 8, 32, 64):
            L.fga_aln_writer_threads(nt)
            p = str(tmp_path / f"w{nt}.1aln")
            assert L.fga_write_1aln_binary(p.encode(), g1.h, g2.h, C.byref(A), 100, ra.encode(), rb.encode(), b"test") == 0
            files.append(open(p, "rb").read())
    finally:
        L.fga_aln_writer_threads(8)
    import re
    strip = lambda b: re.sub(rb"\n! [^\n]*\n", b"\n!\n", b, count=1)        # noqa: E731  (the time stamp)
    assert all(strip(f) == strip(files[0]) for f in files[1:])
    a, t, ts, _, _ = read_1aln(L, str(tmp_path / "w32.1aln"))
    assert ts == 100 and np.array_equal(t, tb)
    for f in ("tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "aread", "bread"):
        assert np.array_equal(a[f], recs[f]), f
    assert np.array_equal(a["flags"] & 1, recs["flags"] & 1)
    txt = H.oneview(str(tmp_path / "w64.1aln"))
    assert sum(1 for ln in txt if ln[0] == "A") == n
    g1.close(); g2.close()


def _synthetic_set(g1, g2, n, seed):
    """n records with traces of 1-20 panels inside the two genomes' contigs, in final order (aread ascending)"""
    from fastga_amd.device import ALN_DTYPE
    rng = np.random.default_rng(seed)
    recs = np.zeros(n, dtype=ALN_DTYPE)
    recs["aread"] = np.sort(rng.integers(0, g1.ncontig, n))
    recs["bread"] = rng.integers(0, g2.ncontig, n)
    recs["flags"] = rng.integers(0, 2, n)
    pieces, off = [], 0
    for i in range(n):
        la, lb = int(g1.clen[recs["aread"][i]]), int(g2.clen[recs["bread"][i]])
        k = max(1, int(min(rng.integers(1, 21), la // 100 - 2, lb // 110 - 2)))
        ab = 100 * int(rng.integers(0, la // 100 - k))
        bb = int(rng.integers(0, lb - 110 * k))
        t = np.empty(2 * k, dtype=np.uint8)
        t[0::2] = rng.integers(0, 12, k)
        t[1::2] = rng.integers(95, 106, k)
        recs["tlen"][i] = 2 * k
        recs["abpos"][i], recs["aepos"][i] = ab, ab + 100 * k
        recs["bbpos"][i], recs["bepos"][i] = bb, bb + int(t[1::2].sum())
        recs["diffs"][i] = int(t[0::2].sum())
        recs["toff"][i] = off
        pieces.append(t)
        off += 2 * k
    recs["unit"], recs["seq"] = -1, np.arange(n)
    return recs, np.concatenate(pieces)


def test_stream_appended_in_stretches_is_the_file_written_at_once(toy_pair, tmp_path, built_library):
    """fga_aln_stream_*: the records of a file appended in stretches of A contigs (what fga_session_run does with its passes
    and fga_multi_run with its ranks), each stretch either appended in one call or formatted first -- by a thread that does
    not hold the stream's turn -- and committed later, give the bytes of fga_write_1aln_binary on the whole set; a stream
    closed without `keep` leaves no file"""
    import re
    from fastga_amd.lib import Alns
    from fastga_amd.gixio import Gdb
    L = built_library
    d, ra, rb = toy_pair
    g1, g2 = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    recs, tb = _synthetic_set(g1, g2, 30_000, 21)
    whole = Alns(len(recs), len(tb), 0, 0, recs.ctypes.data, tb.ctypes.data)
    p0 = str(tmp_path / "whole.1aln")
    assert L.fga_write_1aln_binary(p0.encode(), g1.h, g2.h, C.byref(whole), 100, ra.encode(), rb.encode(), b"test") == 0
    strip = lambda b: re.sub(rb"\n! [^\n]*\n", b"\n!\n", b, count=1)        # noqa: E731  (the time stamp)
    ref = strip(open(p0, "rb").read())
    # stretches: cut where the A contig changes, near 1/4, 1/2, 3/4 (and an empty one)
    cuts = [0]
    for f in (0.25, 0.5, 0.5, 0.75):
        i = int(f * len(recs))
        while 0 < i < len(recs) and recs["aread"][i] == recs["aread"][i - 1]:
            i += 1
        cuts.append(max(i, cuts[-1]))
    cuts.append(len(recs))
    sets = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r = recs[a:b].copy()
        t0 = int(recs["toff"][a]) if b > a else 0
        t1 = int(recs["toff"][b - 1] + recs["tlen"][b - 1]) if b > a else 0
        r["toff"] -= t0
        tt = np.ascontiguousarray(tb[t0:t1]) if t1 > t0 else np.zeros(1, np.uint8)
        sets.append((r, tt, Alns(len(r), t1 - t0, 0, 0, r.ctypes.data, tt.ctypes.data)))
    for how in ("append", "format-then-commit"):
        p = str(tmp_path / (how + ".1aln"))
        h = C.c_void_p()
        assert L.fga_aln_stream_open(p.encode(), g1.h, g2.h, 100, ra.encode(), rb.encode(), b"test", C.byref(h)) == 0
        if how == "append":
            for _, _, A in sets:
                assert L.fga_aln_stream_append(h, C.byref(A)) == 0
        else:
            assert L.fga_aln_stream_preformats(h) == 1
            blocks = []
            for _, _, A in reversed(sets):                      # formatted in any order ..
                b = C.c_void_p()
                if A.naln > 0:
                    assert L.fga_aln_stream_format(h, C.byref(A), C.byref(b)) == 0 and b.value
                blocks.append(b)
            for b in reversed(blocks):                          # .. committed in the file's
                if b.value:
                    assert L.fga_aln_stream_commit(h, b) == 0
        assert L.fga_aln_stream_records(h) == len(recs)
        assert L.fga_aln_stream_close(h, 1) == 0
        assert strip(open(p, "rb").read()) == ref, how
    p = str(tmp_path / "dropped.1aln")
    h = C.c_void_p()
    assert L.fga_aln_stream_open(p.encode(), g1.h, g2.h, 100, ra.encode(), rb.encode(), b"test", C.byref(h)) == 0
    assert L.fga_aln_stream_append(h, C.byref(sets[0][2])) == 0
    b = C.c_void_p()
    assert L.fga_aln_stream_format(h, C.byref(sets[-1][2]), C.byref(b)) == 0
    L.fga_aln_block_free(b)                                     # a block that is never committed
    assert L.fga_aln_stream_close(h, 0) == 0 and not os.path.exists(p)
    g1.close(); g2.close()
