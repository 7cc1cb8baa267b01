"""Named mask files on the GPU path: the index built on the device with a genome's masks named equals the host producer's
(itself pinned against `GIXmake <genome> #<mask>`: tests/test_mask_files.py), and a comparison with `masks1` named is the
comparison the reference makes over the index its own `GIXmake -T1 A #<mask>` wrote (`FastGA -M`), line for line.
(The reference's own `FastGA A #m B` loses its mask arguments: they reach GIXmake through system(), where the shell reads
` #m` as a comment -- so the pin is the two-step form.)"""
import os

import numpy as np
import pytest

from oracle import harness as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def masked_pair(tmp_path_factory, built_library):
    if not (H.have_reference() and os.path.exists(H.ref_bin("BEDtoANO"))):
        pytest.skip("oracle/_ref with BEDtoANO did not travel")
    from fastga_amd import synth
    d = str(tmp_path_factory.mktemp("maskgpu"))
    lens = synth.contig_lengths(31, 6, 600_000)
    A, mA, B, mB = synth.make_pair(31, lens, 0.03, repeat_frac=0.20, inv_frac=0.05, swap_frac=0.05)
    synth.write_fasta(os.path.join(d, "A.fa"), A, prefix="a")
    synth.write_fasta(os.path.join(d, "B.fa"), B, prefix="b")
    for g in "AB":
        H.run([H.ref_bin("FAtoGDB"), g + ".fa"], cwd=d)
    rng = np.random.default_rng(9)
    with open(os.path.join(d, "rep.bed"), "w") as f:                      # mask a tenth of A in 2-kbp pieces
        for c, n in enumerate(lens):
            for s in sorted(rng.integers(0, int(n) - 2000, max(1, int(n) // 20000))):
                f.write(f"a{c}\t{int(s)}\t{int(s) + 2000}\n")
    H.run([H.ref_bin("BEDtoANO"), "rep.bed", "A.1gdb"], cwd=d)
    return d


def test_device_index_with_named_mask_equals_host(masked_pair, built_library):
    import ctypes as C
    from fastga_amd import device as D
    from fastga_amd.gixio import Gdb, Gix, build_gix
    d, L = masked_pair, built_library
    g = Gdb(os.path.join(d, "A.1gdb"))
    arr = (C.c_char_p * 1)(os.path.join(d, "rep").encode())
    assert L.fga_gdb_apply_masks(g.h, arr, 1) == 0, L.fga_last_error()
    od = os.path.join(d, "host")
    os.makedirs(od, exist_ok=True)
    build_gix(g, os.path.join(od, "A"), 8, use_mask=True)
    host = Gix(os.path.join(od, "A.gix"))
    dev = D.Device(0)
    dg, x = D.build_gix_device(dev, g, 8, host_copy=True, use_mask=True)
    a, b = x.entries(), host.entries()
    assert (b[:, 7] != 0).sum() > 1000 and np.array_equal(a, b) and np.array_equal(x.index, host.index)
    dg.free(); x.close(); host.close(); g.close(); dev.close()


def test_comparison_with_a_named_mask_is_the_reference_s(masked_pair, built_library):
    from fastga_amd import device as D
    d = masked_pair
    ra, rb = os.path.join(d, "A"), os.path.join(d, "B")
    H.run([H.ref_bin("GIXmake"), "-T1", f"-P{d}", ra, "#" + os.path.join(d, "rep")], cwd=d)
    H.run([H.ref_bin("GIXmake"), "-T1", f"-P{d}", rb], cwd=d)
    H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=4, flags=("-M",))
    H.ref_fastga(ra, rb, d, os.path.join(d, "refplain"), threads=4)
    ours = os.path.join(d, "ours.1aln")
    st = D.run(ra, rb, ours, nthreads=4, soft_mask=True, masks1=[os.path.join(d, "rep.1ano")], build_index=True,
               reference_threads=4)
    keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]      # noqa: E731
    a, b = keep(H.oneview(ours)), keep(H.oneview(os.path.join(d, "ref.1aln")))
    assert st["nlive"] > 0 and a == b
    assert b != keep(H.oneview(os.path.join(d, "refplain.1aln")))        # the mask really changed the reference's answer


def test_masks_for_one_genome_only_over_several_ranks(masked_pair, built_library):
    """one genome masked (its index is built anew on the device), the other with its index files on disk: a sliced session
    cuts its prefix ranges from ONE kind of count, so fga_multi_open has both indices built on the devices -- and the result
    is the one-GPU run's (pinned to the reference above) with any number of ranks (this combination failed with
    'a sliced session wants both genome indices as files, or neither')"""
    from fastga_amd import device as D
    d = masked_pair
    ra, rb = os.path.join(d, "A"), os.path.join(d, "B")
    if not os.path.exists(rb + ".gix"):
        H.run([H.ref_bin("GIXmake"), "-T1", f"-P{d}", rb], cwd=d)
    keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]      # noqa: E731
    mask = [os.path.join(d, "rep.1ano")]
    one = os.path.join(d, "one_m.1aln")
    st1 = D.run(ra, rb, one, nthreads=4, soft_mask=True, masks1=mask, reference_threads=4)
    ref = keep(H.oneview(one))
    assert st1["nlive"] > 0
    for devices in ((0, 0), (0, 0, 0)):
        out = os.path.join(d, "multi_m%d.1aln" % len(devices))
        st = D.run_multi(ra, rb, out, devices=devices, nthreads=4, soft_mask=True, masks1=mask, reference_threads=4)
        assert keep(H.oneview(out)) == ref and st["nlive"] == st1["nlive"], devices
