"""GPU parity: the index built on the device (fga_dgix_build) against the host producer (fga_gix_build, itself pinned
against the reference's GIXmake byte for byte): same contig order, widths, table parts, prefix index and table bytes;
and the seed merge over device-built indices gives the same seeds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(root, nthreads=8):
    from fastga_amd.gixio import Gix, Gdb
    from fastga_amd import device as D
    g = Gdb(root + ".gdb")
    host = Gix(root + ".gix")
    dev = D.Device(0)
    dg, x = D.build_gix_device(dev, g, nthreads, host_copy=True)
    assert (x.nents, x.ebytes, x.postbytes, x.contbytes, x.nctg, x.nparts) == \
           (host.nents, host.ebytes, host.postbytes, host.contbytes, host.nctg, host.nparts)
    assert np.array_equal(x.perm, host.perm)
    assert np.array_equal(x.index, host.index)
    assert np.array_equal(x.partbeg, host.partbeg)
    a, b = x.entries(), host.entries()
    assert np.array_equal(a, b)
    assert x.maxpre == host.maxpre
    dg.free(); x.close(); host.close(); g.close(); dev.close()


def test_device_gix_equals_host_gix(toy_pair):
    d, ra, rb = toy_pair
    _check(ra)
    _check(rb)


def test_device_gix_on_awkward_genome(tmp_path, built_library):
    """zero-length and sub-k-mer contigs, equal lengths, homopolymer / tandem repeats (two k-mers per base: the key
    buffer is regrown), masks ignored"""
    from tests.edge_inputs import make_edge_scaffolds, write_edge_fasta
    from fastga_amd.gixio import Gdb, fasta_to_gdb, build_gix
    d = str(tmp_path)
    fa = os.path.join(d, "E.fa")
    write_edge_fasta(fa, make_edge_scaffolds(3), seed=1)
    fasta_to_gdb(fa, os.path.join(d, "E"))
    g = Gdb(os.path.join(d, "E.gdb"))
    build_gix(g, os.path.join(d, "E"), 4)
    g.close()
    _check(os.path.join(d, "E"), nthreads=4)


def test_seed_merge_over_device_built_indices(toy_pair):
    from fastga_amd.gixio import Gix, Gdb
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    dev = D.Device(0)
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    dA, dB = dev.upload(A), dev.upload(B)
    want = D.seed_merge(dev, dA, dB).download()
    ga, gb = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    xA, hA = D.build_gix_device(dev, ga, 8)
    xB, hB = D.build_gix_device(dev, gb, 8)
    got = D.seed_merge(dev, xA, xB).download()
    assert len(got) == len(want) > 1000
    names = list(want.dtype.names)
    assert np.array_equal(np.sort(got, order=names), np.sort(want, order=names))
    for o in (dA, dB, xA, xB):
        o.free()
    dev.close()


def test_end_to_end_without_index_files(toy_pair, tmp_path):
    """only <root>.gdb + .bps present: the session builds both indices on the device; same .1aln as the reference run on
    the complete files"""
    import shutil
    from tests.test_end_to_end_gpu import _compare  # noqa: F401
    from fastga_amd import device as D
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    d, ra, rb = toy_pair
    w = str(tmp_path)
    roots = []
    for r in (ra, rb):
        n = os.path.basename(r)
        shutil.copy(r + ".gdb", os.path.join(w, n + ".gdb"))
        shutil.copy(os.path.join(os.path.dirname(r), "." + n + ".bps"), os.path.join(w, "." + n + ".bps"))
        roots.append(os.path.join(w, n))
    ours = os.path.join(w, "ours.1aln")
    st = D.run(roots[0], roots[1], ours, nthreads=8)
    assert st["nlive"] > 0 and not os.path.exists(roots[0] + ".gix")
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=8)
    keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]          # noqa: E731
    assert keep(H.oneview(ours)) == keep(H.oneview(os.path.join(w, "ref.1aln")))


def test_device_gix_mask_bytes_and_files(tmp_path, built_library):
    """soft-mask bytes from the device build equal the host producer's (itself pinned against `GIXmake -T1 #`), and the
    .gix/.ktab files written from the device-built index are the host producer's files byte for byte"""
    import filecmp
    from fastga_amd import workload, synth, device as D
    from fastga_amd.gixio import Gdb, Gix
    d = str(tmp_path)
    lens = synth.contig_lengths(9, 10, 600_000)
    A, mA, _, _ = synth.make_pair(9, lens, 0.0, repeat_frac=0.15, self_only=True)
    ra = workload.build_genome(d, "A", A, masks=mA, use_mask=True)
    host = Gix(ra + ".gix")
    g = Gdb(ra + ".gdb")
    dev = D.Device(0)
    dg, x = D.build_gix_device(dev, g, 8, host_copy=True, use_mask=True)
    assert np.array_equal(x.entries(), host.entries())
    assert int(x.entries()[:, 7].max()) > 0                      # masks really present
    out = os.path.join(d, "dev")
    os.makedirs(out)
    assert dev.L.fga_gix_write_files(x.h, os.path.join(out, "A").encode()) == 0
    names = ["A.gix"] + [f".A.ktab.{p+1}" for p in range(host.nparts)]
    for n in names:
        assert filecmp.cmp(os.path.join(out, n), os.path.join(d, n), shallow=False), n
    dg.free(); x.close(); host.close(); g.close(); dev.close()
