"""GPU end to end: our FastGA hot path vs the REAL reference FastGA (oracle/_ref) on the same GDB/GIX files:
identical `.1aln` content as printed by the reference's own ONEview, minus provenance ('!') and path ('<') lines
(SURVEY.md hard part 10)."""
import os

import pytest

pytestmark = pytest.mark.gpu


def _compare(ra, rb, workdir, strict_order=True, allow_empty=False, **kw):
    from fastga_amd import device as D
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    ours = os.path.join(workdir, "ours.1aln")
    st = D.run(ra, rb, ours, nthreads=8, reference_threads=8, **kw)       # ties on (aread, abpos) as FastGA -T8 orders them
    flags = []
    if kw.get("symmetric"):
        flags.append("-S")
    if "freq" in kw:
        flags.append(f"-f{kw['freq']}")
    if "identity" in kw:
        flags.append(f"-i{kw['identity']}")
    if "chain_min" in kw:
        flags.append(f"-c{kw['chain_min']}")
    H.ref_fastga(ra, rb, workdir, os.path.join(workdir, "ref"), threads=8, flags=flags)
    a = H.oneview(ours)
    b = H.oneview(os.path.join(workdir, "ref.1aln"))
    assert allow_empty or st["nlive"] > 0
    assert len(a) == len(b), (len(a), len(b), st)
    if strict_order:
        for x, y in zip(a, b):
            assert x == y
    else:
        # records that tie on (aread, abpos) are ordered by thread slot in the reference's la_merge
        # (FastGA.c:3906-3918, SURVEY.md hard part 7): compare header verbatim, records as a multiset,
        # and the (aread, abpos) order of both files
        ha, ra_ = _split_records(a)
        hb, rb_ = _split_records(b)
        assert ha == hb
        assert sorted(ra_) == sorted(rb_)
        ka = [tuple(int(v) for v in r[0].split()[1:3]) for r in ra_]
        kb = [tuple(int(v) for v in r[0].split()[1:3]) for r in rb_]
        assert ka == sorted(ka) and kb == sorted(kb) and ka == kb
    return st


def _split_records(lines):
    first = next((i for i, ln in enumerate(lines) if ln.startswith("A ")), len(lines))
    recs, cur = [], []
    for ln in lines[first:]:
        if ln.startswith("A ") and cur:
            recs.append(tuple(cur))
            cur = []
        cur.append(ln)
    if cur:
        recs.append(tuple(cur))
    return lines[:first], recs


def test_pair_default_matches_reference(toy_pair, tmp_path):
    d, ra, rb = toy_pair
    st = _compare(ra, rb, str(tmp_path))
    assert st["nhits"] >= st["nlive"]


def test_pair_symmetric_and_options(toy_pair, tmp_path):
    d, ra, rb = toy_pair
    _compare(ra, rb, str(tmp_path), symmetric=True, freq=6, identity=0.8, chain_min=60)


def test_cutoff_above_255_matches_reference(huge_family_pair, tmp_path):
    """-f330 on a 450-copy family whose k-mers have 250-400 partners: some stay below the cutoff, some do not (at the
    default -f10 they are all dropped); the reference takes any -f (FastGA.c:4497-4499)"""
    d, ra, rb = huge_family_pair
    st = _compare(ra, rb, str(tmp_path), freq=330)
    assert st["nseeds"] > 2_000_000 and st["nlive"] > 1000


def test_divergent_pair_matches_reference(tmp_path, built_library):
    from fastga_amd import workload
    d = str(tmp_path)
    ra, rb = workload.build_pair(d, seed=77, ncontig=10, total=800_000, divergence=0.10,
                                 repeat_frac=0.10, inv_frac=0.05, swap_frac=0.05)
    _compare(ra, rb, d)


def test_self_comparison_matches_reference(tmp_path, built_library):
    """FastGA A (self): both (p,q) and (q,p) seeds, borders at the main diagonal (FastGA.c:3245-3258)."""
    from fastga_amd import workload, synth
    import numpy as np
    d = str(tmp_path)
    rng = np.random.default_rng(5)
    lens = synth.contig_lengths(5, 10, 600_000)
    A, mA, _, _ = synth.make_pair(5, lens, 0.0, repeat_frac=0.0, self_only=True)
    # plant diverged copies inside and across contigs so the self comparison has something to find
    for _ in range(30):
        c1, c2 = rng.integers(0, len(A), 2)
        L = int(rng.integers(800, 6000))
        if len(A[c1]) <= L + 10 or len(A[c2]) <= L + 10:
            continue
        s = int(rng.integers(0, len(A[c1]) - L))
        t = int(rng.integers(0, len(A[c2]) - L))
        cp = synth.mutate(rng, A[c1][s:s + L], float(rng.uniform(0.01, 0.08)))[:L]
        if rng.random() < 0.4:
            cp = synth.revcomp(cp)
        A[c2][t:t + len(cp)] = cp
    ra = workload.build_genome(d, "S", A)
    _compare(ra, None, d)


def test_soft_masked_pair_matches_reference(tmp_path, built_library):
    """-M: mlen = plen, seeds inside masked (lower-case) repeats are dropped (FastGA.c:824-832, 954)."""
    from fastga_amd import workload, synth
    d = str(tmp_path)
    lens = synth.contig_lengths(9, 10, 600_000)
    A, mA, B, mB = synth.make_pair(9, lens, 0.03, repeat_frac=0.15, inv_frac=0.05, swap_frac=0.05)
    ra = workload.build_genome(d, "A", A, masks=mA, use_mask=True)
    rb = workload.build_genome(d, "B", B, masks=None)
    from fastga_amd import device as D
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    plain = D.run(ra, rb, os.path.join(d, "plain.1aln"), nthreads=8)
    ours = os.path.join(d, "ours.1aln")
    st = D.run(ra, rb, ours, nthreads=8, soft_mask=True)
    assert st["nseeds"] < plain["nseeds"]                 # the mask really removed seeds
    H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=8, flags=["-M"])
    assert H.oneview(ours) == H.oneview(os.path.join(d, "ref.1aln"))


def test_reference_tools_accept_our_1aln(toy_pair, tmp_path):
    """Drop-in at the tool level: the reference's own ALNtoPAF (plain and -x, which re-aligns between trace points
    and so needs the bases through the GDB named in the file) prints the same PAF for our .1aln as for its own."""
    from fastga_amd import device as D
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    d, ra, rb = toy_pair
    w = str(tmp_path)
    ours = os.path.join(w, "ours.1aln")
    D.run(ra, rb, ours, nthreads=8)
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=8)
    for flags in ((), ("-x",)):
        a = H.run([H.ref_bin("ALNtoPAF"), "-T4", *flags, ours], cwd=w).stdout.splitlines()
        b = H.run([H.ref_bin("ALNtoPAF"), "-T4", *flags, os.path.join(w, "ref.1aln")], cwd=w).stdout.splitlines()
        assert len(a) > 10 and a == b, flags


def test_runs_on_the_reference_s_own_index_files(toy_pair, tmp_path):
    """The drop-in scenario: FAtoGDB + GIXmake of the reference build the inputs (binary <root>.1gdb, .bps, .gix,
    .ktab.*); our pipeline reads them as they are and writes the same .1aln as the reference FastGA on the same files."""
    import shutil
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    d, ra, rb = toy_pair
    w = str(tmp_path)
    roots = []
    for r in (ra, rb):
        fa = os.path.join(w, os.path.basename(r) + ".fa")
        shutil.copy(r + ".fa", fa)
        roots.append(H.ref_build_index(fa, w, threads=8))
    assert os.path.exists(roots[0] + ".1gdb") and not os.path.exists(roots[0] + ".gdb")
    _compare(roots[0], roots[1], w)


def test_native_paf_equals_alntopaf_on_our_1aln(toy_pair, tmp_path):
    """`FastGA -paf[x|S|ms]` without a second process: edit scripts on the device (fga_trace_pts), gap regrouping and
    formatting on the host, byte-identical to what the reference's ALNtoPAF prints for the .1aln of the same run"""
    from fastga_amd import device as D
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    d, ra, rb = toy_pair
    w = str(tmp_path)
    ours = os.path.join(w, "ours.1aln")
    for opts, flags in (("", 0), ("x", 2), ("S", 8), ("ms", 1 | 4)):
        paf = os.path.join(w, f"ours_{opts}.paf")
        st = D.run(ra, rb, ours, nthreads=8, paf_path=paf, paf_flags=flags)
        exp = H.run([H.ref_bin("ALNtoPAF"), "-T4"] + (["-" + opts] if opts else []) + [ours], cwd=w).stdout
        got = open(paf).read()
        assert got.count("\n") == st["nlive"] > 10
        assert got == exp, opts
        assert (st["trace_kernel_ms"] > 0) == (flags != 0)
    # self comparison, complement and forward same-contig alignments included
    ours = os.path.join(w, "self.1aln")
    paf = os.path.join(w, "self.paf")
    st = D.run(ra, None, ours, nthreads=8, paf_path=paf, paf_flags=2)
    exp = H.run([H.ref_bin("ALNtoPAF"), "-T4", "-x", ours], cwd=w).stdout
    assert open(paf).read() == exp and st["nlive"] > 0


def test_cli_default_output_is_paf_on_stdout(toy_pair, tmp_path):
    from oracle import harness as H
    import subprocess
    d, ra, rb = toy_pair
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastga_amd", "bin", "FastGA")
    w = str(tmp_path)
    r = subprocess.run([exe, "-pafx", "-1:cli", ra, rb], cwd=w, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) > 10 and all("\tcg:Z:" in ln and len(ln.split("\t")) >= 15 for ln in lines)
    if H.have_reference():
        assert r.stdout == H.run([H.ref_bin("ALNtoPAF"), "-x", os.path.join(w, "cli.1aln")], cwd=w).stdout
    r3 = subprocess.run([exe, "-psl", "-1:cli2", ra, rb], cwd=w, capture_output=True, text=True)
    assert r3.returncode == 0 and len(r3.stdout.splitlines()) == len(lines)
    if H.have_reference() and os.path.exists(H.ref_bin("ALNtoPSL")):
        assert r3.stdout == H.run([H.ref_bin("ALNtoPSL"), os.path.join(w, "cli2.1aln")], cwd=w).stdout
    r2 = subprocess.run([exe, ra, rb], cwd=w, capture_output=True, text=True)
    assert r2.returncode == 0 and len(r2.stdout.splitlines()) == len(lines)
    assert all("cg:Z:" not in ln for ln in r2.stdout.splitlines())


def test_converter_tools_on_reference_files(toy_pair, tmp_path):
    """bin/ALNtoPAF and bin/ALNtoPSL (fga_read_1aln -> fga_trace_pts -> writers) on a .1aln made by the REFERENCE FastGA:
    byte-identical to the reference's converters, pair and self"""
    import subprocess
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    d, ra, rb = toy_pair
    w = str(tmp_path)
    bindir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastga_amd", "bin")
    for tag, b in (("pair", rb), ("self", None)):
        H.ref_fastga(ra, b, w, os.path.join(w, tag), threads=4)
        aln = os.path.join(w, tag + ".1aln")
        for opts in ("-x", "-mS", "-xsw"):
            exp = H.run([H.ref_bin("ALNtoPAF"), "-T4", opts, aln], cwd=w).stdout
            r = subprocess.run([os.path.join(bindir, "ALNtoPAF"), "-T4", opts, aln], cwd=w, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert r.stdout == exp, (tag, opts)
        exp = H.run([H.ref_bin("ALNtoPSL"), "-T4", aln], cwd=w).stdout
        r = subprocess.run([os.path.join(bindir, "ALNtoPSL"), "-T4", aln], cwd=w, capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout == exp, tag


def test_cli_process_contract_log_threads_cleanup(tmp_path, built_library):
    """-L:<log> gets the -v lines and the reference's resource lines; -T reaches the device index build; a GDB made from
    a FASTA source is removed again unless -k, and with -k the index files are written too (FastGA.c:152-196, 4444-4637)"""
    import gzip
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "fastga_amd", "bin", "FastGA")
    w = str(tmp_path)
    for n in ("toy_A", "toy_B"):
        with gzip.open(os.path.join(root, "tests", "golden", n + ".fa.gz"), "rb") as f:
            open(os.path.join(w, n + ".fa"), "wb").write(f.read())
    r = subprocess.run([exe, "-v", "-T4", "-L:run.log", "-1:out", "toy_A.fa", "toy_B.fa"], cwd=w, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(os.path.join(w, "out.1aln")) > 0
    log = open(os.path.join(w, "run.log")).read()
    for key in ("Creating genome data base (GDB)", "Total seeds =", "non-redundant aln's", "Resources for phase:",
                "Total Resources:"):
        assert key in log and key in r.stderr, key
    assert "FastGA -v -T4" in log                                       # the command line opens the log entry
    assert not os.path.exists(os.path.join(w, "toy_A.gdb")) and not os.path.exists(os.path.join(w, ".toy_B.bps"))
    r = subprocess.run([exe, "-k", "-1:out2", "toy_A.fa", "toy_B.fa"], cwd=w, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for f in ("toy_A.gdb", ".toy_A.bps", "toy_A.gix", "toy_B.gdb", "toy_B.gix"):
        assert os.path.exists(os.path.join(w, f)), f
    from oracle import harness as H
    if H.have_reference():
        keep = lambda p: [ln for ln in H.oneview(p) if ln[0] not in "!<"]      # noqa: E731
        assert keep(os.path.join(w, "out.1aln")) == keep(os.path.join(w, "out2.1aln"))
        golden = [ln.rstrip("\n") for ln in open(os.path.join(root, "tests", "golden", "toy_AvB.1aln.txt"))]
        assert keep(os.path.join(w, "out.1aln")) == [ln for ln in golden if ln[0] not in "!<"]


def test_native_paf_at_a_scale_where_the_device_regroups(tmp_path, built_library, capfd):
    """25 Mbp repeat-heavy genome against itself (> 10^5 short alignments, millions of indels): the set is large enough for
    fga_trace_pts_regrouped to keep the scripts on the device (Gap_Improver one lane per alignment, long ones handed back
    to the formatter threads) -- the PAF with =/X CIGARs and the PSL must still be ALNtoPAF's / ALNtoPSL's, byte for byte"""
    from fastga_amd import device as D, synth, workload
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    w = str(tmp_path)
    lens = synth.contig_lengths(2, 20, 25_000_000)
    A, _, _, _ = synth.make_pair(2, lens, 0.02, repeat_frac=0.30, inv_frac=0.02, swap_frac=0.02, self_only=True)
    ra = workload.build_genome(w, "A", A, threads=16, gix=False)
    ours, paf, psl = os.path.join(w, "s.1aln"), os.path.join(w, "s.paf"), os.path.join(w, "s.psl")
    os.environ["FGA_TRACE_TIMING"] = "1"                    # (its stderr line says how many alignments were handed back)
    try:
        st = D.run(ra, None, ours, nthreads=16, paf_path=paf, paf_flags=2, build_index=True)
        D.run(ra, None, ours, nthreads=16, paf_path=psl, paf_flags=32, build_index=True)
    finally:
        del os.environ["FGA_TRACE_TIMING"]
    import re
    said = re.findall(r"regrouping [0-9.]+ ms \((\d+) of (\d+) alignments handed back", capfd.readouterr().err)
    assert len(said) == 2 and all(int(n) == st["nlive"] and int(back) < 0.1 * int(n) for back, n in said), said
    assert st["nlive"] > 50_000 and st["trace_kernel_ms"] > 0
    exp = H.run([H.ref_bin("ALNtoPAF"), "-T16", "-x", ours], cwd=w).stdout
    assert open(paf).read() == exp
    exp = H.run([H.ref_bin("ALNtoPSL"), "-T16", ours], cwd=w).stdout
    assert open(psl).read() == exp
