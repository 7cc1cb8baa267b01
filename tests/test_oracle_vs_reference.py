"""Pins the CPU oracle (oracle/*.c) against the REAL reference built into oracle/_ref.

The reference has no test suite or golden vectors of its own, so parity is pinned by running it:
 * seed merge  : FastGA's own `_pair.*` seed temp files (kept alive by the unlink shim) vs oracle_seed_merge
 * alignment   : reference Local_Alignment (libalign_ref.so through ctypes) vs oracle_local_alignment, call by call
These tests need /root/reference-built binaries and are skipped where oracle/_ref did not travel.
"""
import os

import numpy as np
import pytest

from oracle import harness as H

needs_ref = pytest.mark.skipif(not H.have_reference(), reason="oracle/_ref (real reference build) not present")


@needs_ref
def test_our_gix_is_byte_identical_to_reference_gixmake(toy_pair, tmp_path):
    """fga_gix_build / fga_fasta_to_gdb produce the reference's bytes (.bps, .ktab.*, .gix)."""
    d, ra, rb = toy_pair
    import shutil
    rd = str(tmp_path)
    shutil.copy(ra + ".fa", os.path.join(rd, "A.fa"))
    root = H.ref_build_index(os.path.join(rd, "A.fa"), rd, threads=8)
    assert open(os.path.join(rd, ".A.bps"), "rb").read() == open(os.path.join(d, ".A.bps"), "rb").read()
    from fastga_amd.gixio import Gix
    ours, ref = Gix(ra + ".gix"), Gix(root + ".gix")
    assert ours.nents == ref.nents and ours.ebytes == ref.ebytes
    assert np.array_equal(ours.index, ref.index)
    assert np.array_equal(ours.perm, ref.perm)
    a = ours.entries()
    b = ref.entries()
    # duplicate k-mers may be ordered differently by the reference's unstable sort (which then also decides
    # which copy carries the group's lcp byte): compare (k-mer, mask, payload) and (k-mer, lcp) as multisets.
    # The lcp byte of the entries whose k-mer differs from its predecessor's in the very first base (true LCP 0: at
    # most three rows, the A|C, C|G, G|T boundaries) is excluded: the reference's GIXmake races on it -- 0 on a quiet
    # machine, sometimes a stale 12 under load (seen with concurrent runs of the reference alone).  The merge never
    # reads it: such an entry starts a new 12-mer panel.  Ours is always the true value 0.
    assert np.array_equal(ours.partbeg, ref.partbeg)
    zero = np.nonzero(a[:, 8] == 0)[0]
    assert len(zero) <= ours.nparts + 4                    # part starts + the three first-base boundaries
    b = b.copy()
    b[zero, 8] = 0
    # k-mers are in the same order
    assert np.array_equal(a[:, :7], b[:, :7])
    for cols in (list(range(8)) + list(range(9, a.shape[1])), list(range(9))):
        x, y = a[:, cols], b[:, cols]
        dif = np.nonzero((a[:, :9] != b[:, :9]).any(axis=1))[0]
        assert np.array_equal(x[np.lexsort(x.T[::-1])], y[np.lexsort(y.T[::-1])]), \
            (len(dif), dif[:8].tolist(), a[dif[:8], 7:9].tolist(), b[dif[:8], 7:9].tolist(), ours.nparts, ref.nparts)


@needs_ref
@pytest.mark.parametrize("flags,flip", [((), False), (("-S",), True)])
def test_seed_merge_oracle_matches_reference_seed_files(toy_pair, tmp_path, flags, flip):
    d, ra, rb = toy_pair
    from fastga_amd.gixio import Gix
    r, seeds = H.ref_fastga(ra, rb, str(tmp_path), os.path.join(str(tmp_path), "out"), threads=4,
                            flags=flags, capture_seeds=True)
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    w = 1 + A.pbyte + B.pbyte
    n, c, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte)
    if flip:
        n2, c2, nh2, ts2 = H.oracle_seed_merge(B.table, B.index, B.pbyte, A.table, A.index, A.pbyte, flip=True)
        n, c, nh = n + n2, c + c2, nh + nh2
    assert len(seeds[0]) + len(seeds[1]) == nh * w
    assert np.array_equal(H.sorted_records(seeds[0], w), H.sorted_records(n, w))
    assert np.array_equal(H.sorted_records(seeds[1], w), H.sorted_records(c, w))
    tot = [ln for ln in r.stderr.splitlines() if "Total seeds" in ln]
    assert tot and f"Total seeds = {nh}," in tot[0]


@needs_ref
def test_self_seed_merge_oracle_matches_reference(toy_pair, tmp_path):
    d, ra, rb = toy_pair
    from fastga_amd.gixio import Gix
    r, seeds = H.ref_fastga(ra, None, str(tmp_path), os.path.join(str(tmp_path), "out"), threads=4,
                            capture_seeds=True)
    A = Gix(ra + ".gix")
    w = 1 + 2 * A.pbyte
    n, c, nh, ts = H.oracle_self_seed_merge(A.table, A.index, A.pbyte)
    assert np.array_equal(H.sorted_records(seeds[0], w), H.sorted_records(n, w))
    assert np.array_equal(H.sorted_records(seeds[1], w), H.sorted_records(c, w))


@needs_ref
@pytest.mark.parametrize("freq", [3, 30])
def test_seed_merge_oracle_freq_branch_matches_reference(family_pair, tmp_path, freq):
    """-f: the |R| >= FREQ drop (FastGA.c:799-823) at a cutoff below and above the default"""
    d, ra, rb = family_pair
    from fastga_amd.gixio import Gix
    r, seeds = H.ref_fastga(ra, rb, str(tmp_path), os.path.join(str(tmp_path), "out"), threads=4,
                            flags=(f"-f{freq}",), capture_seeds=True)
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    w = 1 + A.pbyte + B.pbyte
    n, c, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte, freq=freq)
    n10, c10, nh10, _ = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte)
    assert (nh < nh10) if freq < 10 else (nh > nh10)    # the cutoff really changes the seed set on this pair
    assert len(seeds[0]) + len(seeds[1]) == nh * w
    assert np.array_equal(H.sorted_records(seeds[0], w), H.sorted_records(n, w))
    assert np.array_equal(H.sorted_records(seeds[1], w), H.sorted_records(c, w))


@needs_ref
def test_seed_merge_oracle_soft_mask_branch_matches_reference(masked_pair, tmp_path):
    """-M: mlen = plen; T1 entries and T2 partners whose mask byte reaches mlen are dropped (FastGA.c:824-832, 954)"""
    d, ra, rb = masked_pair
    from fastga_amd.gixio import Gix
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    assert A.entries()[:, 7].any() and B.entries()[:, 7].any()          # the tables do carry mask bytes
    w = 1 + A.pbyte + B.pbyte
    r, seeds = H.ref_fastga(ra, rb, str(tmp_path), os.path.join(str(tmp_path), "out"), threads=4, flags=("-M",),
                            capture_seeds=True)
    n, c, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte, soft_mask=True)
    _, _, nh_plain, _ = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte)
    assert 0 < nh < nh_plain
    assert np.array_equal(H.sorted_records(seeds[0], w), H.sorted_records(n, w))
    assert np.array_equal(H.sorted_records(seeds[1], w), H.sorted_records(c, w))
    # -M -S: the flipped pass tests the masks of both sides the other way round (FastGA.c:833-892)
    r, seeds = H.ref_fastga(ra, rb, str(tmp_path), os.path.join(str(tmp_path), "out2"), threads=4,
                            flags=("-M", "-S"), capture_seeds=True)
    n2, c2, nh2, _ = H.oracle_seed_merge(B.table, B.index, B.pbyte, A.table, A.index, A.pbyte, soft_mask=True,
                                         flip=True)
    assert np.array_equal(H.sorted_records(seeds[0], w), H.sorted_records(n + n2, w))
    assert np.array_equal(H.sorted_records(seeds[1], w), H.sorted_records(c + c2, w))


@needs_ref
def test_self_seed_merge_oracle_soft_mask_matches_reference(masked_pair, tmp_path):
    """self comparison with -M (new_self_merge_thread with mlen = plen, FastGA.c:1791-1799): BASELINE config 3's mode"""
    d, ra, rb = masked_pair
    from fastga_amd.gixio import Gix
    A = Gix(ra + ".gix")
    w = 1 + 2 * A.pbyte
    r, seeds = H.ref_fastga(ra, None, str(tmp_path), os.path.join(str(tmp_path), "out"), threads=4, flags=("-M",),
                            capture_seeds=True)
    n, c, nh, ts = H.oracle_self_seed_merge(A.table, A.index, A.pbyte, soft_mask=True)
    _, _, nh_plain, _ = H.oracle_self_seed_merge(A.table, A.index, A.pbyte)
    assert 0 < nh < nh_plain
    assert np.array_equal(H.sorted_records(seeds[0], w), H.sorted_records(n, w))
    assert np.array_equal(H.sorted_records(seeds[1], w), H.sorted_records(c, w))


def _starts_in_range(A, B, low, hgh, anti):
    """every start point (anti+k)/2, (anti-k)/2 of Local_Alignment lies inside the two sequences -- what FastGA's own
    calls guarantee (hit boxes come from seeds inside the contigs); outside, the reference reads past its buffers and
    its result depends on whatever lies there"""
    while ((anti - hgh) >> 1) < 0:              # Local_Alignment's own adjustment (align.c:1463-1464)
        hgh -= 1
    for k in (low, hgh):
        x = (anti + k) >> 1
        if not (0 <= x <= len(A) and 0 <= x - k <= len(B)):
            return False
    return hgh >= low


def _random_case(rng):
    while True:
        c = _random_case_any(rng)
        if _starts_in_range(c[0], c[1], c[3], c[4], c[5]):
            return c


def _random_case_any(rng):
    from fastga_amd import synth
    n = int(rng.integers(300, 6000))
    A = rng.integers(0, 4, n, dtype=np.uint8)
    div = float(rng.choice([0.0, 0.01, 0.03, 0.08, 0.15, 0.3]))
    i0 = int(rng.integers(0, n // 3))
    i1 = int(rng.integers(2 * n // 3, n))
    acomp = bool(rng.random() < 0.4)
    Ause = synth.revcomp(A) if acomp else A
    pre = rng.integers(0, 4, int(rng.integers(0, 300)), dtype=np.uint8)
    post = rng.integers(0, 4, int(rng.integers(0, 300)), dtype=np.uint8)
    if rng.random() < 0.3:
        pre = pre[:0]
    if rng.random() < 0.3:
        post = post[:0]
    B = np.concatenate([pre, synth.mutate(rng, Ause[i0:i1], div), post])
    a = int(rng.integers(i0, i1))
    b = min(len(B) - 1, len(pre) + (a - i0))
    if rng.random() < 0.1:
        b = int(rng.integers(0, len(B)))          # off-diagonal probe: mostly "nothing found" paths
    w = int(rng.integers(0, 120))
    low = (a - b) - int(rng.integers(0, w + 1))
    lb = hb = -1
    if rng.random() < 0.2:
        lb, hb = int(rng.integers(0, 50)), int(rng.integers(0, 50))
    return Ause, B, acomp, low, low + w, a + b, lb, hb


@needs_ref
def test_local_alignment_oracle_matches_reference_call_by_call():
    rng = np.random.default_rng(20260926)
    spec = H.oracle_spec()
    ref = H.RefAligner()
    for _ in range(300):
        A, B, acomp, low, hgh, anti, lb, hb = _random_case(rng)
        abuf, bbuf = H.pad_seq(A), H.pad_seq(B)
        r = ref.align(abuf, bbuf, low, hgh, anti, lb, hb, acomp)
        o = H.oracle_local_alignment(abuf, bbuf, spec, low, hgh, anti, lb, hb, acomp)
        assert r[:5] == o[:5]
        assert np.array_equal(r[5], o[5])
    ref.close()


@needs_ref
def test_local_alignment_oracle_long_and_self_with_borders():
    from fastga_amd import synth
    rng = np.random.default_rng(7)
    freq = (0.3, 0.2, 0.2, 0.3)
    spec = H.oracle_spec(freq=freq)
    ref = H.RefAligner(freq=freq)
    for trial in range(12):
        n = int(rng.integers(20000, 120000))
        A = rng.integers(0, 4, n, dtype=np.uint8)
        if trial % 3 == 2:                      # self comparison with a planted diverged copy, border = diagonal 0
            cp = synth.mutate(rng, A[1000:6000], 0.05)
            A[n // 2:n // 2 + len(cp)] = cp
            abuf = H.pad_seq(A)
            a, b = n // 2 + 2500, 3500
            low, hgh, anti = a - b - 20, a - b + 20, a + b
            r = ref.align(abuf, abuf, low, hgh, anti, low - 1, -1, selfie=True)
            o = H.oracle_local_alignment(abuf, abuf, spec, low, hgh, anti, low - 1, -1, selfie=True)
        else:
            B = synth.mutate(rng, A, float(rng.choice([0.01, 0.05, 0.1, 0.2])))
            abuf, bbuf = H.pad_seq(A), H.pad_seq(B)
            a = int(rng.integers(0, n))
            b = min(len(B) - 1, int(a * len(B) / n))
            low, hgh, anti = a - b - 30, a - b + 30, a + b
            r = ref.align(abuf, bbuf, low, hgh, anti)
            o = H.oracle_local_alignment(abuf, bbuf, spec, low, hgh, anti)
        assert r[:5] == o[:5]
        assert np.array_equal(r[5], o[5])
        assert r[2] - r[0] > 1000
    ref.close()


@needs_ref
def test_soft_mask_bytes_match_single_thread_reference(tmp_path, built_library):
    """Lower-case FASTA runs -> per-entry mask byte (GIXmake `#`, GIXmake.c:1100-1108).  The reference's masked
    build races between its threads (a thread parks a sentinel in the *next* contig's first mask interval while
    another thread may be scanning that contig, GIXmake.c:1085-1088), so the pin is `GIXmake -T1`."""
    import shutil
    from fastga_amd import synth
    from fastga_amd.gixio import Gdb, Gix, fasta_to_gdb, build_gix
    d = str(tmp_path)
    lens = synth.contig_lengths(3, 10, 400_000)
    A, mA, _, _ = synth.make_pair(3, lens, 0.0, repeat_frac=0.15, self_only=True)
    fa = os.path.join(d, "A.fa")
    synth.write_fasta(fa, A, prefix="a", masks=mA)
    od = os.path.join(d, "ours")
    os.makedirs(od)
    fasta_to_gdb(fa, os.path.join(od, "A"))
    g = Gdb(os.path.join(od, "A.gdb"))
    assert g.L.fga_gdb_nmask(g.h) > 0
    build_gix(g, os.path.join(od, "A"), 8, use_mask=True)
    H.run([H.ref_bin("FAtoGDB"), fa], cwd=d)
    H.run([H.ref_bin("GIXmake"), "-T1", f"-P{d}", os.path.join(d, "A"), "#"], cwd=d)
    ga_, gb_ = Gix(os.path.join(od, "A.gix")), Gix(os.path.join(d, "A.gix"))    # keep alive: entries() are views
    a, b = ga_.entries(), gb_.entries()
    assert a.shape == b.shape and (a[:, 7] != 0).sum() > 1000
    for cols in (list(range(8)) + list(range(9, a.shape[1])), list(range(7)) + [8]):
        x, y = a[:, cols], b[:, cols]
        assert np.array_equal(x[np.lexsort(x.T[::-1])], y[np.lexsort(y.T[::-1])])


@needs_ref
def test_trace_pts_oracle_matches_reference_call_by_call():
    """oracle/trace_oracle.c against the reference's Compute_Trace_PTS (align.c:6171, GREEDIEST) on the alignments the
    reference's Local_Alignment finds: same edit script, same difference count"""
    from fastga_amd import synth
    rng = np.random.default_rng(20260927)
    ref = H.RefAligner()
    done = 0
    for _ in range(400):
        A, B, acomp, low, hgh, anti, lb, hb = _random_case(rng)
        if acomp:
            continue
        abuf, bbuf = H.pad_seq(A), H.pad_seq(B)
        p = ref.align(abuf, bbuf, low, hgh, anti, lb, hb, False)
        if p[2] <= p[0]:
            continue
        rd, rt = ref.trace_pts(abuf, bbuf, p)
        od, ot = H.oracle_trace_pts(abuf, bbuf, p)
        assert rd == od
        assert np.array_equal(rt, ot)
        done += 1
    assert done > 100
    for trial in range(4):                       # self comparisons: the edit script may not touch the main diagonal
        n = 40000
        A = rng.integers(0, 4, n, dtype=np.uint8)
        cp = synth.mutate(rng, A[1000:6000], 0.08)
        A[n // 2:n // 2 + len(cp)] = cp
        abuf = H.pad_seq(A)
        a, b = n // 2 + 2500, 3500
        if trial & 1:
            a, b = b, a
        low, hgh, anti = a - b - 20, a - b + 20, a + b
        lb, hb = (low - 1, -1) if a > b else (-1, hgh + 1)
        p = ref.align(abuf, abuf, low, hgh, anti, lb, hb, selfie=True)
        assert p[2] - p[0] > 1000
        rd, rt = ref.trace_pts(abuf, abuf, p, selfie=True)
        od, ot = H.oracle_trace_pts(abuf, abuf, p, selfie=True)
        assert rd == od and np.array_equal(rt, ot) and len(rt) > 50
    ref.close()


@needs_ref
def test_gap_improver_oracle_matches_reference_call_by_call():
    """oracle/gap_oracle.c against the reference's Gap_Improver (align.c:6714) applied to its own Compute_Trace_PTS
    output; inputs with clustered indels so that boxes form and are rewritten"""
    from fastga_amd import synth
    rng = np.random.default_rng(20260928)
    ref = H.RefAligner()
    done = changed = 0
    for it in range(300):
        n = int(rng.integers(1500, 12000))
        A = rng.integers(0, 4, n, dtype=np.uint8)
        if it % 3 == 0:                            # low-complexity stretches make shifted gaps cheap
            for _ in range(int(rng.integers(1, 6))):
                u = rng.integers(0, 4, int(rng.integers(1, 5)), dtype=np.uint8)
                p0 = int(rng.integers(0, n - 200))
                ln = int(rng.integers(20, 150))
                A[p0:p0 + ln] = np.tile(u, ln // len(u) + 1)[:ln]
        B = synth.mutate(rng, A, float(rng.choice([0.02, 0.05, 0.1, 0.2])))
        if it % 2 == 0:                            # multi-base indels
            B = list(B)
            for _ in range(int(rng.integers(1, 12))):
                p0 = int(rng.integers(10, len(B) - 40))
                if rng.random() < 0.5:
                    del B[p0:p0 + int(rng.integers(1, 9))]
                else:
                    B[p0:p0] = list(rng.integers(0, 4, int(rng.integers(1, 9))))
            B = np.array(B, dtype=np.uint8)
        abuf, bbuf = H.pad_seq(A), H.pad_seq(B)
        a = n // 2
        b = min(len(B) - 1, a)
        p = ref.align(abuf, bbuf, a - b - 40, a - b + 40, a + b)
        if p[2] - p[0] < 300:
            continue
        d0, t0 = ref.trace_pts(abuf, bbuf, p)
        rd, rt = ref.trace_pts(abuf, bbuf, p, improve=True)
        od, ot = H.oracle_gap_improver(abuf, bbuf, p, t0, d0)
        assert rd == od
        assert np.array_equal(rt, ot)
        assert len(rt) == len(t0)
        changed += int(not np.array_equal(rt, t0))
        done += 1
    assert done > 150 and changed > 30
    ref.close()
