"""Full-size parity on the GPU (BASELINE.json configs at the sizes the bench quotes), through the C-ABI, against the
REAL reference FastGA run on the same box and the same index files (oracle/_ref travels prebuilt), or -- for the 1 Gbp
self comparison, which the reference needs minutes for -- against a digest produced with the reference by
tests/golden/make_golden_config3.py.  Covers what the toy-size tests cannot: contig-long alignments (tens of thousands
of wave steps), sequence windows falling back to HBM, arena growth, seed-buffer re-runs, the self + soft-mask mode.

Identity is judged on ONEview's text, line for line: the runs ask for the reference's own tie order (reference_threads =
the reference's -T: records that tie on (aread, abpos) go by the slot of the search thread that held them, la_merge,
FastGA.c:3906-3918; fga_order.c).  The 1 Gbp / 3 Gbp runs compare with digests made by the reference (header, records as a
multiset, (aread, abpos) order; `lines_md5` = the record lines in sequence where the golden file has it)."""
import ctypes as C
import json
import os

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
T = max(4, min(32, os.cpu_count() or 8))


def _write_index_files(roots, use_mask=False, nthreads=8):
    """<root>.gix + .ktab.N for the reference, from index builds on the device (byte-identical to the host producer's
    and to GIXmake's: tests/test_gix_device_gpu.py, tests/test_oracle_vs_reference.py)"""
    from fastga_amd import device as D
    from fastga_amd.gixio import Gdb
    dev = D.Device(0)
    for r in roots:
        g = Gdb(r + ".gdb")
        dgx, xg = D.build_gix_device(dev, g, nthreads, host_copy=True, use_mask=use_mask)
        assert dev.L.fga_gix_write_files(xg.h, r.encode()) == 0, dev.L.fga_last_error()
        dgx.free(); xg.close(); g.close()
    dev.close()


def _reference_index_files_equal_device_builds(roots, d, nthreads=8):
    """<root>.gix + .ktab.N made by the REFERENCE's GIXmake (the files the reference run then consumes), and the table a
    device build gives for the same GDB compared with them: same prefix index, same k-mer order, same entries up to the
    order inside runs of equal k-mers (the reference's sort is not stable there; which copy carries the run's lcp byte
    follows) -- so the two programs of the comparison below do NOT share an index builder (SURVEY hard part 9)."""
    import numpy as np
    from fastga_amd import device as D
    from fastga_amd.gixio import Gdb, Gix
    from oracle import harness as H
    dev = D.Device(0)
    for r in roots:
        H.run([H.ref_bin("GIXmake"), f"-T{nthreads}", f"-P{d}", r], cwd=d)
        ref = Gix(r + ".gix")
        g = Gdb(r + ".gdb")
        dgx, xg = D.build_gix_device(dev, g, nthreads, host_copy=True)
        assert xg.nents == ref.nents and xg.ebytes == ref.ebytes
        assert np.array_equal(xg.index, ref.index) and np.array_equal(xg.perm, ref.perm)
        assert np.array_equal(xg.partbeg, ref.partbeg)
        a, b = xg.entries(), ref.entries()
        assert np.array_equal(a[:, :7], b[:, :7])                     # k-mers in the same order
        zero = np.nonzero(a[:, 8] == 0)[0]                            # part starts + first-base boundaries (GIXmake races there)
        assert len(zero) <= xg.nparts + 4
        b = b.copy()
        b[zero, 8] = 0
        dif = np.nonzero((a != b).any(axis=1))[0]
        assert len(dif) < max(1000, len(a) // 100), len(dif)          # only inside runs of equal k-mers
        for cols in (list(range(8)) + list(range(9, a.shape[1])), list(range(9))):
            x, y = a[dif][:, cols], b[dif][:, cols]
            assert np.array_equal(x[np.lexsort(x.T[::-1])], y[np.lexsort(y.T[::-1])])
        dgx.free(); xg.close(); ref.close(); g.close()
    dev.close()


def _compare_with_reference(ra, rb, d, flags=(), strict=True, pafx=False, ref_threads=T, **kw):
    from fastga_amd import device as D, workload
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    ours = os.path.join(d, "ours.1aln")
    paf = os.path.join(d, "ours.paf") if pafx else None
    st = D.run(ra, rb, ours, nthreads=T, paf_path=paf, paf_flags=2 if pafx else 0, reference_threads=ref_threads, **kw)
    H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=ref_threads, flags=flags)
    a, b = H.oneview(ours), H.oneview(os.path.join(d, "ref.1aln"))
    da, db = workload.digest_1aln(a), workload.digest_1aln(b)
    assert da == db, (da, db, st)
    if strict:
        assert a == b
    if pafx:
        exp = H.run([H.ref_bin("ALNtoPAF"), f"-T{T}", "-x", os.path.join(d, "ref.1aln" if a == b else "ours.1aln")],
                    cwd=d).stdout
        assert open(paf).read() == exp
    return st, da


def test_config2_100mbp_pair_is_identical_to_the_reference(tmp_path_factory, built_library):
    """configs[1] at full size: 100 Mbp x 100 Mbp, 2 %: strict .1aln identity and -pafx identity"""
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("c2"))
    ra, rb = workload.build_config2(d, mbp=100.0, threads=T)
    _reference_index_files_equal_device_builds((ra, rb), d)      # the reference reads its own GIXmake's files ...
    st, dg = _compare_with_reference(ra, rb, d, strict=True, pafx=True, build_index=True)   # ... and ours builds its index on the device
    assert dg["records"] > 1500 and st["nwaves"] > 5_000_000        # contig-long alignments were really extended
    # the work itself is pinned too, not only its outcome: the number of wave steps of the whole comparison.  (A wrong
    # root cell for the trim point of a wave that never sets one changed it by 14 in 7.9 M while every record stayed
    # identical: found through this number.)
    assert st["nwaves"] == 7_877_004


def test_config1_substitute_s1_86mbp_is_identical_to_the_reference(tmp_path_factory, built_library):
    """configs[0]'s stand-in S1: 86 Mbp pair, log-uniform contig lengths, 4.5 % divergence, 15 % repeats,
    5 % inversions / swaps -- the reference run with -T8 like the configuration asks"""
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("s1"))
    ra, rb = workload.build_config1_s1(d, threads=T, gix=False)
    _write_index_files((ra, rb), nthreads=8)
    st, dg = _compare_with_reference(ra, rb, d, strict=True, ref_threads=8)
    assert dg["records"] > 5000 and st["nhits"] > 5000



def test_divergent_50mbp_pair_is_identical_to_the_reference(tmp_path_factory, built_library):
    """the 10 % shape of configs[4] on one GPU: deep waves, WAVE_LAG pruning, wide waves in the LDS ring"""
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("c5"))
    ra, rb = workload.build_pair(d, seed=5, ncontig=40, total=50_000_000, divergence=0.10, repeat_frac=0.05,
                                 inv_frac=0.02, swap_frac=0.02, threads=T, gix=False)
    _write_index_files((ra, rb))
    st, dg = _compare_with_reference(ra, rb, d, strict=True)
    assert dg["records"] > 500


def test_config3_150mbp_self_soft_masked_is_identical_to_the_reference(tmp_path_factory, built_library):
    """configs[2]'s shape against the live reference: repeat-heavy self comparison with -M (new_self_merge_thread with
    mlen = plen, FastGA.c:1791-1799; borders at the main diagonal, FastGA.c:3245-3258)"""
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("c3"))
    root = workload.build_config3(d, mbp=150.0, threads=T)
    _write_index_files((root,), use_mask=True)
    st, dg = _compare_with_reference(root, None, d, flags=("-M",), strict=True, soft_mask=True)
    assert dg["records"] > 5000


@pytest.mark.skipif(os.environ.get("FGA_SKIP_1G") == "1", reason="FGA_SKIP_1G=1")
def test_human_scale_contigs_3x94mbp_pair_is_identical_to_the_reference(tmp_path_factory, built_library):
    """the per-GPU shard shape of configs[3]: three 94-Mbp contigs against their 1 % diverged copies.  One alignment
    spans a whole contig (~10^6 wave steps, ~2 x 10^7 trace-point cells: arena levels up to 2^24 cells), the trace
    scratch is sized per unit, and the number of resident wavefronts must not depend on the contig size."""
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("c4"))
    ra, rb = workload.build_pair(d, seed=3, ncontig=3, total=282_000_000, divergence=0.01, repeat_frac=0.05,
                                 inv_frac=0.02, swap_frac=0.02, threads=T, gix=False)
    _write_index_files((ra, rb), nthreads=3)            # the reference wants -T <= the number of contigs
    os.environ["FGA_EXTEND_PROFILE"] = "1"
    try:
        st, dg = _compare_with_reference(ra, rb, d, strict=True, ref_threads=3)
    finally:
        os.environ.pop("FGA_EXTEND_PROFILE", None)
    assert dg["records"] > 1000 and st["nwaves"] > 3_000_000


@pytest.mark.skipif(os.environ.get("FGA_SKIP_1G") == "1", reason="FGA_SKIP_1G=1")
def test_human_scale_contigs_3x94mbp_at_10_percent_is_identical_to_the_reference(tmp_path_factory, built_library):
    """the same shape at 10 % divergence: every wave of the contig-long alignments is hundreds of diagonals wide (ring
    spills, ~10^6 wave steps each, 2.7 x 10^8 in all); the trace-point pool must stay a small part of the HBM"""
    import time
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("c5"))
    ra, rb = workload.build_pair(d, seed=3, ncontig=3, total=282_000_000, divergence=0.10, repeat_frac=0.05,
                                 inv_frac=0.02, swap_frac=0.02, threads=T, gix=False)
    _write_index_files((ra, rb), nthreads=3)
    t = time.time()
    st, dg = _compare_with_reference(ra, rb, d, strict=True, ref_threads=3)
    print(f"human-scale 10 %: ours {1000*sum(st[k] for k in ('merge_s','sort_s','chain_s','extend_s','filter_s','write_s')):.0f} ms (extension kernel {st['extend_kernel_ms']:.0f} ms), "
          f"peak HBM {st['hbm_peak_bytes']/2**30:.1f} GiB, reference + ours {time.time()-t:.0f} s, {dg['records']} records")
    assert dg["records"] > 1000 and st["hbm_peak_bytes"] < 64 << 30


@pytest.mark.skipif(os.environ.get("FGA_SKIP_1G") == "1", reason="FGA_SKIP_1G=1")
def test_config3_1gbp_self_soft_masked_matches_the_reference_digest(tmp_path_factory, built_library):
    """configs[2] at full size, index built on the device; expected digest: tests/golden/config3_1000m_digest.json, made
    with the real reference by tests/golden/make_golden_config3.py"""
    from fastga_amd import device as D, workload
    from oracle import harness as H
    gold = os.path.join(HERE, "golden", "config3_1000m_digest.json")
    if not os.path.exists(gold):
        pytest.skip("golden digest not generated")
    exp = json.load(open(gold))
    if not os.path.exists(H.ref_bin("ONEview")):
        pytest.skip("oracle/_ref/ONEview did not travel")
    d = str(tmp_path_factory.mktemp("c3g"))
    root = workload.build_config3(d, mbp=1000.0, threads=T)
    ours = os.path.join(d, "ours.1aln")
    st = D.run(root, None, ours, nthreads=T, soft_mask=True, reference_threads=exp.get("reference_threads", 8))
    got = workload.digest_1aln(H.oneview(ours))
    # the reference halves the seed count of a self comparison per merge thread (FastGA.c:1906): sum of floors
    assert 0 <= st["nseeds"] - exp["total_seeds"] <= 64
    for k in ("records", "header_md5", "records_md5", "order_md5", "lines_md5"):      # lines_md5: the record lines in sequence
        assert got[k] == exp[k], (k, got, exp, st)


@pytest.mark.skipif(os.environ.get("FGA_SKIP_1G") == "1", reason="FGA_SKIP_1G=1")
@pytest.mark.parametrize("name,div", [("config4", 0.01), ("config5", 0.10)])
def test_configs_4_and_5_at_3gbp_match_the_reference_digests(tmp_path_factory, built_library, name, div):
    """configs[3] / configs[4] at their stated size on ONE GPU: 3 Gbp x 3 Gbp, 32 contigs of ~94 Mbp, 45 % repeats, 1 % and
    10 % divergence.  Indices built on the device (2.4 G entries each), the seeds exceed one sort pass so phase 2 runs over
    A-contig parts (the reference's NPARTS loop), and the same comparison cut into 8 prefix ranges x 8 parts (the 8-GPU
    decomposition, emulated) must give the same file.  Expected digests: tests/golden/<name>_3000m_digest.json, made on
    a 256-core box with the REAL reference (its own GIXmake -T32 index, FastGA -T32: 74 s and 123 s wall) by
    tests/golden/make_golden_config4.py."""
    from fastga_amd import device as D, parallel, workload
    from oracle import harness as H
    gold = os.path.join(HERE, "golden", f"{name}_3000m_digest.json")
    exp = json.load(open(gold))
    oneview = H.ref_bin("ONEview")
    if not os.path.exists(oneview):
        pytest.skip("oracle/_ref/ONEview did not travel")
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    import shutil, tempfile
    d = tempfile.mkdtemp(prefix="fga_" + name + "_", dir=base)
    try:
        ra, rb = workload.build_config4(d, mbp=3000.0, divergence=div, threads=T)
        ses = D.Session(ra, rb)
        ours = os.path.join(d, "ours.1aln")
        RT = exp.get("reference_threads", 32)                     # ties on (aread, abpos) as FastGA -T32 wrote them
        st = ses.run(out_path=ours, nthreads=T, reference_threads=RT)
        got = workload.digest_1aln_stream(ours, oneview)
        assert st["nseeds"] == exp["total_seeds"] and st["nhits"] == exp["hits"] and st["nalns"] == exp["alignments"]
        for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5"):   # lines_md5: the record lines in sequence
            assert got[k] == exp[k], (k, got, exp, st)
        assert st["nparts"] >= 2                                   # more seeds than one sort pass takes
        assert st["hbm_peak_bytes"] < 286 << 30               # everything the two passes need at once fits the device
        os.unlink(ours)
        out8 = os.path.join(d, "parts8.1aln")
        st8 = parallel.run_parts_on_one_gpu(ses, 8, out_path=out8, nthreads=T, reference_threads=RT)
        assert st8["nseeds"] == exp["total_seeds"] and len(st8["part_seed_counts"]) == 8
        # the ranks' stretches of the .1aln streamed (contigs dealt in original order: within 1.25 x of the balanced deal's
        # heaviest part), and once more with the balanced deal and the file written from the gathered records
        assert st8["streamed"] and max(st8["part_seed_counts"]) < 1.25 * sum(st8["part_seed_counts"]) / 8
        got8 = workload.digest_1aln_stream(out8, oneview)
        assert got8 == got
        os.unlink(out8)
        st8 = parallel.run_parts_on_one_gpu(ses, 8, stream=False, out_path=out8, nthreads=T, reference_threads=RT)
        assert not st8["streamed"] and max(st8["part_seed_counts"]) < 1.1 * min(st8["part_seed_counts"])   # balanced on seed counts
        assert workload.digest_1aln_stream(out8, oneview) == got
        ses.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)
