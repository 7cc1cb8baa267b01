"""fga_read_1aln (host C): the reference's own binary .1aln files -- small ones with plain lists and a larger one whose
T / X lists are Huffman-coded (ONElib trains a list code after ~100 KB of list data) -- must come back as exactly the
records ONEview prints; our own binary and the db paths of the header likewise."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import harness as H
from tests.test_aln_writer import _parse_records

needs_ref = pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref (real reference build) not present")


def read_1aln(L, path):
    from fastga_amd.lib import Alns
    from fastga_amd.device import ALN_DTYPE
    out = C.POINTER(Alns)()
    ts = C.c_int()
    d1, d2 = C.c_char_p(), C.c_char_p()
    rc = L.fga_read_1aln(path.encode(), C.byref(out), C.byref(ts), C.byref(d1), C.byref(d2))
    assert rc == 0, L.fga_last_error()
    o = out.contents
    a = np.frombuffer((C.c_char * (o.naln * ALN_DTYPE.itemsize)).from_address(o.alns), dtype=ALN_DTYPE).copy() \
        if o.naln else np.zeros(0, ALN_DTYPE)
    t = np.frombuffer((C.c_char * max(o.ntrace, 1)).from_address(o.tbytes), dtype=np.uint8)[:o.ntrace].copy()
    L.fga_alns_free(out)
    return a, t, ts.value, d1.value, d2.value


def _same_records(a, t, alns, tb):
    assert len(a) == len(alns) and np.array_equal(t, tb)
    for f in ("tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "flags", "aread", "bread", "toff"):
        assert np.array_equal(a[f], alns[f]), f


@needs_ref
@pytest.mark.parametrize("self_cmp", [False, True])
def test_reads_reference_1aln(toy_pair, tmp_path, built_library, self_cmp):
    d, ra, rb = toy_pair
    w = str(tmp_path)
    H.ref_fastga(ra, None if self_cmp else rb, w, os.path.join(w, "ref"), threads=4)
    ref = os.path.join(w, "ref.1aln")
    alns, tb = _parse_records(H.oneview(ref))
    a, t, ts, d1, d2 = read_1aln(built_library, ref)
    _same_records(a, t, alns, tb)
    assert ts == 100 and os.path.basename(d1.decode()).startswith("A")
    assert (d2 is None) == self_cmp
    # our own binary writer's file reads back the same way
    from fastga_amd.lib import Alns
    from fastga_amd.gixio import Gdb
    g1 = Gdb(ra + ".gdb")
    g2 = None if self_cmp else Gdb(rb + ".gdb")
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    ours = os.path.join(w, "ours.1aln")
    assert built_library.fga_write_1aln_binary(ours.encode(), g1.h, g2.h if g2 else None, C.byref(A), 100,
                                               (ra + ".gdb").encode(), None if self_cmp else (rb + ".gdb").encode(),
                                               b"t") == 0
    a2, t2, _, e1, e2 = read_1aln(built_library, ours)
    _same_records(a2, t2, alns, tb)
    assert e1.decode().endswith("A.gdb")
    # the text forms: our ASCII writer's file and what ONEview prints for the reference's file
    txt = os.path.join(w, "ours.txt.1aln")
    assert built_library.fga_write_1aln(txt.encode(), g1.h, g2.h if g2 else None, C.byref(A), 100,
                                        (ra + ".gdb").encode(), None if self_cmp else (rb + ".gdb").encode(), b"t") == 0
    a3, t3, ts3, f1, _ = read_1aln(built_library, txt)
    _same_records(a3, t3, alns, tb)
    assert ts3 == 100 and f1.decode().endswith("A.gdb")
    view = os.path.join(w, "view.1aln")
    open(view, "w").write(H.run([H.ref_bin("ONEview"), ref]).stdout)
    a4, t4, _, _, _ = read_1aln(built_library, view)
    _same_records(a4, t4, alns, tb)


@needs_ref
def test_reads_compressed_lists(tmp_path, built_library):
    from fastga_amd import workload
    w = str(tmp_path)
    ra, rb = workload.build_pair(w, seed=3, ncontig=8, total=12_000_000, divergence=0.03, inv_frac=0.02, threads=8)
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=8)
    ref = os.path.join(w, "ref.1aln")
    raw = open(ref, "rb").read()
    tbyte = 0x80 | ((ord("T") - 65) << 1) | 1
    assert bytes([tbyte]) in raw                           # some T lines carry the compression flag
    alns, tb = _parse_records(H.oneview(ref))
    assert int(alns["tlen"].sum()) > 150_000
    a, t, ts, d1, d2 = read_1aln(built_library, ref)
    _same_records(a, t, alns, tb)
    # our writer can train list codes too (FGA_ALN_CODEC=1): the reference's tools must read such a file like their own
    from fastga_amd.lib import Alns
    from fastga_amd.gixio import Gdb
    g1, g2 = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    sizes = {}
    for name, env in (("coded", "1"), ("plain", None)):
        if env:
            os.environ["FGA_ALN_CODEC"] = env
        try:
            out = os.path.join(w, name + ".1aln")
            assert built_library.fga_write_1aln_binary(out.encode(), g1.h, g2.h, C.byref(A), 100,
                                                       (ra + ".gdb").encode(), (rb + ".gdb").encode(), b"t") == 0
        finally:
            os.environ.pop("FGA_ALN_CODEC", None)
        sizes[name] = os.path.getsize(out)
        keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]          # noqa: E731
        assert keep(H.oneview(out)) == keep(H.oneview(ref)), name
        a2, t2, _, _, _ = read_1aln(built_library, out)
        _same_records(a2, t2, alns, tb)
    coded = open(os.path.join(w, "coded.1aln"), "rb").read()
    assert bytes([tbyte]) in coded and sizes["coded"] < 0.8 * sizes["plain"]
    exp = H.run([H.ref_bin("ALNtoPAF"), "-T4", "-x", ref], cwd=w).stdout
    got = H.run([H.ref_bin("ALNtoPAF"), "-T4", "-x", os.path.join(w, "coded.1aln")], cwd=w).stdout
    assert got == exp
    print("sizes", sizes, "reference", os.path.getsize(ref))


def test_rejects_other_files(tmp_path, built_library, toy_pair):
    from fastga_amd.lib import Alns
    d, ra, rb = toy_pair
    out = C.POINTER(Alns)()
    L = built_library
    p = os.path.join(str(tmp_path), "x.1aln")
    open(p, "w").write("1 3 seq 2 1\n" + "S 4 acgt\n" * 8)                    # another ONEcode file type
    assert L.fga_read_1aln(p.encode(), C.byref(out), None, None, None) != 0
    assert b"neither" in L.fga_last_error()
    assert L.fga_read_1aln((ra + ".fa").encode(), C.byref(out), None, None, None) != 0
    assert L.fga_read_1aln(b"/nonexistent.1aln", C.byref(out), None, None, None) != 0


def _tool(name):
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastga_amd", "bin", name)


@needs_ref
def test_alntopaf_tool_plain_needs_no_gpu(toy_pair, tmp_path, built_library):
    """our ALNtoPAF on the reference's own .1aln: plain PAF is host-only and equals the reference's; base-level output
    asks for the device and, without one, fails loudly instead of falling back"""
    import subprocess
    d, ra, rb = toy_pair
    w = str(tmp_path)
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=4)
    exp = H.run([H.ref_bin("ALNtoPAF"), "-T2", os.path.join(w, "ref.1aln")], cwd=w).stdout
    r = subprocess.run([_tool("ALNtoPAF"), "-T3", "ref"], cwd=w, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == exp
    exp = H.run([H.ref_bin("ALNtoPAF"), "-w", os.path.join(w, "ref.1aln")], cwd=w).stdout
    assert subprocess.run([_tool("ALNtoPAF"), "-w", "ref.1aln"], cwd=w, capture_output=True, text=True).stdout == exp
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([_tool("ALNtoPAF"), "-x", "ref"], cwd=w, capture_output=True, text=True)
        assert r.returncode != 0 and "no CPU fallback" in r.stderr and r.stdout == ""
    r = subprocess.run([_tool("ALNtoPAF"), "-mx", "ref"], cwd=w, capture_output=True, text=True)
    assert r.returncode != 0 and "Only one of -m or -x" in r.stderr


def _try_read(L, path):
    from fastga_amd.lib import Alns
    out = C.POINTER(Alns)()
    ts = C.c_int()
    rc = L.fga_read_1aln(path.encode(), C.byref(out), C.byref(ts), None, None)
    if rc == 0:
        n = out.contents.naln
        L.fga_alns_free(out)
        return 0, n
    return rc, L.fga_last_error().decode()


def test_text_form_is_parsed_within_its_bounds(tmp_path, built_library):
    """the text reader never scans beyond a line (no trailing newline, short lists, duplicate or over-long T / X lines,
    16-bit trace forms are errors, not overruns) and its cost is linear in the file size"""
    import time
    head = "1 3 aln 2 1\nt 100\n"
    ok = head + "A 0 10 250 1 20 262\nD 3\nT 3 90 100 52\nX 3 1 1 1"           # no newline at the end
    p = str(tmp_path / "a.1aln")
    open(p, "w").write(ok)
    assert _try_read(built_library, p) == (0, 1)
    bad = {
        "short": head + "A 0 10 250 1 20 262\nT 5 90 100\nA 1 0 100 1 0 100\n",
        "twice": head + "A 0 10 250 1 20 262\nT 2 90 100\nT 4 1 2 3 4\n",
        "longer": head + "A 0 10 250 1 20 262\nX 2 1 1\nT 3 90 100 52\n",
        "wide": head + "A 0 10 250 1 20 262\nT 2 90 300\n",
        "tspace": "1 3 aln 2 1\nt 200\nA 0 10 250 1 20 262\n",
        "aline": head + "A 0 10 250 1\n",
    }
    for name, txt in bad.items():
        q = str(tmp_path / (name + ".1aln"))
        open(q, "w").write(txt)
        rc, msg = _try_read(built_library, q)
        assert rc != 0, name
    # linear time: 200k alignments (about 9 MB of text) in well under a second per MB
    big = [head]
    for i in range(200_000):
        big.append(f"A 0 {i} {i+300} 1 {i} {i+300}\nD 2\nT 3 100 100 100\nX 3 0 1 1\n")
    q = str(tmp_path / "big.1aln")
    open(q, "w").write("".join(big))
    t = time.time()
    assert _try_read(built_library, q) == (0, 200_000)
    assert time.time() - t < 5.0


def test_rejects_crafted_list_codes(built_library):
    """a footer list code with an escape length outside [0,16] or a code length > 16 must be refused by the parser, and
    a stream whose escape runs off its end by the decoder (reachable from .1aln and .1gdb files)"""
    L = built_library
    import struct
    if not hasattr(L, "fga_one_codec_parse"):
        pytest.skip("codec entry points are internal in this build")
    # struct fga_one_codec { int have; int esc, esclen; uint8_t len[256]; uint8_t *look; } -- opaque here: a zeroed blob
    codec = (C.c_uint8 * 4096)()
    def ser(esc, esclen, lens):
        b = bytes([0]) + struct.pack("<ii", esc, esclen)
        for i, l in enumerate(lens):
            b += bytes([l])
            if l > 0 or i == esc:
                b += struct.pack("<H", 0)
        return b
    lens = [0] * 256
    L.fga_one_codec_parse.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.fga_one_codec_parse.restype = C.c_int
    for esc, esclen, mod in ((3, -5, None), (3, 1 << 20, None), (300, 4, None), (3, 4, 40)):
        ll = list(lens)
        if mod is not None:
            ll[7] = mod
        blob = ser(esc, esclen, ll)
        assert L.fga_one_codec_parse(codec, blob, len(blob)) != 0
