"""Mask files (`#<mask>[.1ano]` arguments, FastGA.c:4568-4573 -> GIXmake.c:1821-1842): fga_gdb_apply_masks reads .1ano / .ano
files (binary and ASCII ONEcode, made here by the reference's BEDtoANO / ONEview), matches their scaffolds to the GDB's,
maps the intervals to contigs and unites them; the host producer's table built with that soft mask must be the table the
REAL `GIXmake -T1 <genome> #<mask> ...` writes, byte for byte.  (-T1: the reference's masked build races between its
threads, tests/test_oracle_vs_reference.py.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import harness as H

needs_ref = pytest.mark.skipif(not (H.have_reference() and os.path.exists(H.ref_bin("BEDtoANO"))),
                               reason="oracle/_ref (real reference build with BEDtoANO) not present")


def _apply(L, g, paths):
    arr = (C.c_char_p * max(len(paths), 1))(*[p.encode() for p in paths])
    return L.fga_gdb_apply_masks(g.h, arr, len(paths))


@pytest.fixture(scope="module")
def masked_genome(tmp_path_factory, built_library):
    """a 3-scaffold genome with gaps (several contigs per scaffold) and lower-case runs, its reference GDB, and two BED
    masks turned into .1ano files by the reference's BEDtoANO"""
    from fastga_amd import synth
    d = str(tmp_path_factory.mktemp("masks"))
    rng = np.random.default_rng(17)
    fa = os.path.join(d, "T.fa")
    with open(fa, "w") as f:
        for s, lens in enumerate(((30000, 12000), (20000,), (9000, 7000, 15000))):
            f.write(f">t{s}\n")
            seq = []
            for k, n in enumerate(lens):
                if k:
                    seq.append("N" * (50 + 10 * k))
                b = np.array(list("acgt"))[rng.integers(0, 4, n)]
                b[200:700] = np.char.lower(b[200:700])
                up = np.char.upper(b)
                up[200:700] = b[200:700]                      # one lower-case run per contig: the implicit mask
                seq.append("".join(up))
            s_ = "".join(seq)
            for i in range(0, len(s_), 80):
                f.write(s_[i:i + 80] + "\n")
    H.run([H.ref_bin("FAtoGDB"), fa], cwd=d)
    # scaffold coordinates; overlapping and out-of-order intervals, one in the second contig of t0, one reversed (BED has none)
    open(os.path.join(d, "m1.bed"), "w").write("t0\t100\t900\nt0\t5000\t5600\nt1\t0\t300\nt2\t9100\t9500\nt0\t850\t1200\n"
                                               "t0\t30100\t30900\nt2\t16200\t17000\n")
    open(os.path.join(d, "m2.bed"), "w").write("t0\t1100\t1300\nt1\t19000\t20000\nt2\t100\t150\n")
    for m in ("m1", "m2"):
        H.run([H.ref_bin("BEDtoANO"), m + ".bed", "T.1gdb"], cwd=d)
        assert os.path.exists(os.path.join(d, m + ".1ano"))
    return d


def _ours(L, d, masks, sub):
    from fastga_amd.gixio import Gdb, Gix, build_gix
    od = os.path.join(d, sub)
    os.makedirs(od, exist_ok=True)
    g = Gdb(os.path.join(d, "T.1gdb"))
    assert _apply(L, g, masks) == 0, L.fga_last_error()
    n = L.fga_gdb_nmask(g.h)
    build_gix(g, os.path.join(od, "T"), 1, use_mask=True)
    g.close()
    return Gix(os.path.join(od, "T.gix")), n


def _ref(d, args):
    from fastga_amd.gixio import Gix
    for f in os.listdir(d):
        if f == "T.gix" or f.startswith(".T.ktab."):
            os.remove(os.path.join(d, f))
    H.run([H.ref_bin("GIXmake"), "-T1", f"-P{d}", os.path.join(d, "T")] + args, cwd=d)
    return Gix(os.path.join(d, "T.gix"))


@needs_ref
@pytest.mark.parametrize("case", ["one", "two", "implicit+named", "ascii"])
def test_table_with_named_masks_is_gixmake_s(masked_genome, built_library, case):
    d, L = masked_genome, built_library
    m1, m2 = os.path.join(d, "m1"), os.path.join(d, "m2.1ano")
    if case == "one":
        ours, ref = _ours(L, d, [m1], "o1"), _ref(d, ["#" + m1])
    elif case == "two":
        ours, ref = _ours(L, d, [m1, m2], "o2"), _ref(d, ["#" + m1, "#" + m2])
    elif case == "implicit+named":
        ours, ref = _ours(L, d, ["", m2], "o3"), _ref(d, ["#", "#" + m2])
    else:                                              # the ASCII form of the same file (what ONEview prints), as <root>.ano
        txt = subprocess.run([H.ref_bin("ONEview"), m1 + ".1ano"], capture_output=True, text=True, check=True).stdout
        os.makedirs(os.path.join(d, "asc"), exist_ok=True)
        open(os.path.join(d, "asc", "m1.ano"), "w").write(txt)
        ours, ref = _ours(L, d, [os.path.join(d, "asc", "m1.ano")], "o4"), _ref(d, ["#" + m1])
    (x, nmask), y = ours, ref
    a, b = x.entries(), y.entries()
    assert nmask > 0 and a.shape == b.shape
    assert (b[:, 7] != 0).sum() > 100                    # the reference really masked entries
    assert np.array_equal(a, b)
    x.close(); y.close()


@needs_ref
def test_mask_file_errors(masked_genome, built_library, tmp_path):
    from fastga_amd.gixio import Gdb, fasta_to_gdb
    d, L = masked_genome, built_library
    g = Gdb(os.path.join(d, "T.1gdb"))
    assert _apply(L, g, [os.path.join(d, "nothere")]) != 0 and b"Cannot find/open ANO file" in L.fga_last_error()
    assert _apply(L, g, [os.path.join(d, "T.fa")]) != 0                     # not an ano file (no T.fa.1ano either)
    g.close()
    # a mask made for another genome: the skeletons differ
    from fastga_amd import synth
    rng = np.random.default_rng(3)
    fa = str(tmp_path / "U.fa")
    synth.write_fasta(fa, [rng.integers(0, 4, 5000, dtype=np.uint8) for _ in range(3)], prefix="t")
    fasta_to_gdb(fa, str(tmp_path / "U"))
    u = Gdb(str(tmp_path / "U.gdb"))
    assert _apply(L, u, [os.path.join(d, "m1")]) != 0 and b"not equivalent" in L.fga_last_error()
    # no masks named: the soft mask becomes empty; "" keeps the GDB's own
    assert _apply(L, u, []) == 0 and L.fga_gdb_nmask(u.h) == 0
    u.close()
