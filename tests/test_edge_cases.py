"""Producer parity on awkward inputs (CPU): our FASTA->GDB and GDB->GIX vs the reference's FAtoGDB + GIXmake."""
import os

import numpy as np
import pytest

from oracle import harness as H
from tests.edge_inputs import make_edge_scaffolds, write_edge_fasta

needs_ref = pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref (real reference build) not present")


@needs_ref
def test_edge_fasta_gdb_and_gix_match_reference(tmp_path, built_library):
    from fastga_amd.gixio import Gix, Gdb, fasta_to_gdb, build_gix
    d = str(tmp_path)
    od, rd = os.path.join(d, "ours"), os.path.join(d, "ref")
    os.makedirs(od); os.makedirs(rd)
    sc = make_edge_scaffolds(3)
    for w in (od, rd):
        write_edge_fasta(os.path.join(w, "E.fa"), sc, seed=1)
    # ours
    fasta_to_gdb(os.path.join(od, "E.fa"), os.path.join(od, "E"))
    g = Gdb(os.path.join(od, "E.gdb"))
    build_gix(g, os.path.join(od, "E"), 4)
    # reference
    root = H.ref_build_index(os.path.join(rd, "E.fa"), rd, threads=4)
    assert open(os.path.join(od, ".E.bps"), "rb").read() == open(os.path.join(rd, ".E.bps"), "rb").read()
    # skeleton: the reference prints its .1gdb, ours is the ASCII form of the same ONEcode file
    ours_txt = [ln for ln in H.oneview(os.path.join(od, "E.gdb")) if ln[0] not in "!<"]
    ref_txt = [ln for ln in H.oneview(os.path.join(rd, "E.1gdb")) if ln[0] not in "!<"]
    assert ours_txt == ref_txt
    assert g.ncontig == 14 and sorted(int(x) for x in g.clen)[:2] == [0, 11]     # leading N-run: zero-length contig
    ours, ref = Gix(os.path.join(od, "E.gix")), Gix(root + ".gix")
    assert ours.nents == ref.nents and ours.ebytes == ref.ebytes
    assert np.array_equal(ours.index, ref.index)
    assert np.array_equal(ours.perm, ref.perm)                 # equal-length contigs: same tie order
    a, b = ours.entries(), ref.entries().copy()
    assert np.array_equal(ours.partbeg, ref.partbeg)           # same table parts ...
    b[a[:, 8] == 0, 8] = 0          # ... GIXmake races on the lcp byte at first-base boundaries (see the toy-pair test)
    assert np.array_equal(a[:, :7], b[:, :7])
    for cols in (list(range(8)) + list(range(9, a.shape[1])), list(range(9))):
        x, y = a[:, cols], b[:, cols]
        assert np.array_equal(x[np.lexsort(x.T[::-1])], y[np.lexsort(y.T[::-1])])
    g.close()


@needs_ref
def test_reference_binary_1gdb_is_read_directly(tmp_path, built_library):
    """fga_gdb_open on the reference's own binary ONEcode skeleton (<root>.1gdb + .<root>.bps) gives the same genome as
    our ASCII <root>.gdb of the same FASTA"""
    from fastga_amd.gixio import Gdb, fasta_to_gdb
    d = str(tmp_path)
    od, rd = os.path.join(d, "ours"), os.path.join(d, "ref")
    os.makedirs(od); os.makedirs(rd)
    sc = make_edge_scaffolds(3)
    for w in (od, rd):
        write_edge_fasta(os.path.join(w, "E.fa"), sc, seed=1)
    fasta_to_gdb(os.path.join(od, "E.fa"), os.path.join(od, "E"))
    H.run([H.ref_bin("FAtoGDB"), os.path.join(rd, "E.fa"), os.path.join(rd, "E.1gdb")], cwd=rd)
    assert b"\n$ 0\n" in open(os.path.join(rd, "E.1gdb"), "rb").read(4096)          # binary container
    a, b = Gdb(os.path.join(od, "E")), Gdb(os.path.join(rd, "E"))
    assert a.ncontig == b.ncontig and a.seqtot == b.seqtot and a.maxctg == b.maxctg
    assert np.array_equal(a.clen, b.clen)
    for c in range(a.ncontig):
        assert np.array_equal(a.contig(c), b.contig(c))
    # the skeleton written back from the binary read equals the reference's own ONEview of it
    import ctypes as C
    out = os.path.join(d, "back.1aln")
    from fastga_amd.lib import Alns
    A = Alns(0, 0, 0, 0, None, None)
    assert a.L.fga_write_1aln_binary(out.encode(), b.h, None, C.byref(A), 100, b"E", None, b"t") == 0
    skel = [ln for ln in H.oneview(out) if ln[0] in "gSGC"]
    ref = [ln for ln in H.oneview(os.path.join(rd, "E.1gdb")) if ln[0] in "SGC"]
    assert [ln for ln in skel if ln[0] != "g"] == ref
    a.close(); b.close()


@needs_ref
def test_reads_codec_compressed_1gdb(tmp_path, built_library):
    """a .1gdb of a fragmented assembly: after ~100 KB of scaffold names the reference's writer trains a Huffman code
    and compresses the remaining S lines (code in the footer).  Our reader must return the same skeleton the
    reference's ONEview prints."""
    import random
    from fastga_amd.gixio import Gdb
    w = str(tmp_path)
    rnd = random.Random(3)
    fa = os.path.join(w, "many.fa")
    with open(fa, "w") as f:
        for i in range(6000):
            f.write(f">scaffold_{i:06d}_len_{rnd.randint(100, 999)} some description text\n")
            f.write("".join(rnd.choice("ACGT") for _ in range(rnd.randint(60, 140))) + "\n")
    H.run([H.ref_bin("FAtoGDB"), fa], cwd=w)
    raw = open(os.path.join(w, "many.1gdb"), "rb").read()
    assert raw.count(b"\xa5") > 1000                       # type byte of a compressed S line
    g = Gdb(os.path.join(w, "many.1gdb"))
    assert g.ncontig == 6000
    out = os.path.join(w, "ours.gdb")
    assert g.L.fga_gdb_write_skeleton(g.h, out.encode(), b"test", b"cmd") == 0
    ours = [ln for ln in open(out).read().splitlines() if ln[:1] in "SGCf"]
    ref = [ln for ln in H.oneview(os.path.join(w, "many.1gdb")) if ln[:1] in "SGCf"]
    assert ours == ref and len(ours) == 12001
    g.close()


def test_gdb_open_reads_a_large_image_with_several_threads(tmp_path, built_library):
    """fga_gdb_open reads a .bps of 64 MB and more in eight slices with pread: a 300 Mbp genome (75 MB of packed bases)
    comes back exactly as the bytes of the file say, contig by contig, first / middle / last contig and the slice seams"""
    import numpy as np
    from fastga_amd import workload
    from fastga_amd.gixio import Gdb
    d = str(tmp_path)
    ra, _ = workload.build_config4(d, mbp=300.0, divergence=0.01, ncontig=12, threads=8, names=("A", "A2"))
    raw = np.fromfile(os.path.join(d, ".A.bps"), dtype=np.uint8)
    assert len(raw) >= 64 << 20
    g = Gdb(ra + ".gdb")
    off = 0
    for c in range(g.ncontig):
        n = int(g.clen[c])
        nb = (n + 3) // 4
        if c in (0, g.ncontig // 2, g.ncontig - 1) or any(off <= len(raw) * k // 8 < off + nb for k in range(1, 8)):
            packed = raw[off:off + nb]
            want = ((packed[:, None] >> (2 * np.arange(4, dtype=np.uint8))) & 3).reshape(-1)[:n]
            assert np.array_equal(g.contig(c), want), c
        off += nb
    assert off == len(raw)
    g.close()
