"""GPU parity: fga_trace_pts (edit scripts from trace points, Compute_Trace_PTS align.c:6171 in GREEDIEST mode) against
the CPU oracle oracle/trace_oracle.c, which tests/test_oracle_vs_reference.py pins call by call against the reference."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _alignments(dev, ra, rb, self_cmp=False, aln_rate=0.35):
    """merge -> sort -> chain -> extend on a prebuilt pair; returns everything the trace stage needs"""
    from fastga_amd.gixio import Gix, Gdb
    from fastga_amd import device as D
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    ga, gb = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    amx, bmx = int(ga.maxctg), int(gb.maxctg)
    dA, dB = dev.upload(A), dev.upload(B)
    seeds = D.seed_merge(dev, dA, None if self_cmp else dB)
    keys = D.seed_sort(dev, seeds, amx, bmx, A.nctg, B.nctg)
    clen = np.zeros(A.nctg, dtype=np.int64)               # the index pads the contig table to the thread count
    clen[:len(ga.clen)] = ga.clen
    hl = D.chain_scan_device(dev, keys, 2000, 170, amx, bmx, clen[A.perm])
    f4 = (C.c_float * 4)()
    ga.L.fga_gdb_freq(ga.h, f4)
    pa, table, score = D.align_spec(0.7, 100, list(f4))
    dga = D.DeviceGenome(dev, ga, A.perm, True)
    dgb = D.DeviceGenome(dev, gb, B.perm, True)
    alns, tb, _ = D.extend(dev, dga, dgb, hl, pa, table, score, aln_min=50, aln_rate=aln_rate, self_cmp=self_cmp)
    keys.free(); seeds.free(); dA.free(); dB.free()
    return ga, gb, dga, dgb, alns, tb


def _check_against_oracle(ga, gb, alns, tb, res, self_cmp=False):
    from fastga_amd import synth
    from oracle import harness as H
    toff, tlen, diffs, ints, _ = res
    assert len(tlen) == len(alns) and len(toff) == len(alns) + 1
    cache = {}
    nindel = 0
    for i, a in enumerate(alns):
        c1, c2, comp = int(a["aread"]), int(a["bread"]), int(a["flags"]) & 1
        if ("a", c1) not in cache:
            cache[("a", c1)] = H.pad_seq(ga.contig(c1))
        if ("b", c2, comp) not in cache:
            s = gb.contig(c2)
            cache[("b", c2, comp)] = H.pad_seq(synth.revcomp(s) if comp else s)
        t = tb[int(a["toff"]):int(a["toff"]) + int(a["tlen"])].astype(np.uint16)
        path = (int(a["abpos"]), int(a["bbpos"]), int(a["aepos"]), int(a["bepos"]), int(a["diffs"]), t)
        selfie = bool(self_cmp and c1 == c2 and not comp)
        od, ot = H.oracle_trace_pts(cache[("a", c1)], cache[("b", c2, comp)], path, selfie=selfie)
        assert int(tlen[i]) == len(ot), i
        assert int(diffs[i]) == od, i
        assert np.array_equal(ints[int(toff[i]):int(toff[i]) + int(tlen[i])], ot), i
        assert od <= int(a["diffs"])                       # the rebuilt script is never worse than the wave's count
        nindel += len(ot)
    assert int(toff[-1]) == nindel == len(ints)
    return nindel


def _replay(ga, gb, alns, res, limit=None):
    """size-independent property: walking an edit script over the two sequences consumes exactly [abpos,aepos) and
    [bbpos,bepos) and meets exactly `diffs` differences (indels + mismatched columns)"""
    from fastga_amd import synth
    toff, tlen, diffs, ints, _ = res
    cache = {}
    for i, a in enumerate(alns[:limit]):
        c1, c2, comp = int(a["aread"]), int(a["bread"]), int(a["flags"]) & 1
        if ("a", c1) not in cache:
            cache[("a", c1)] = ga.contig(c1)
        if ("b", c2, comp) not in cache:
            s = gb.contig(c2)
            cache[("b", c2, comp)] = synth.revcomp(s) if comp else s
        A, B = cache[("a", c1)], cache[("b", c2, comp)]
        k, h, nd = int(a["abpos"]), int(a["bbpos"]), 0           # 0-based next unaligned base of A and of B
        for e in ints[int(toff[i]):int(toff[i + 1])]:
            e = int(e)
            n = (-e - 1 - k) if e < 0 else (e - 1 - h)
            assert n >= 0
            nd += int((A[k:k + n] != B[h:h + n]).sum()) + 1
            k, h = (k + n, h + n + 1) if e < 0 else (k + n + 1, h + n)
        n = int(a["aepos"]) - k
        assert n >= 0 and h + n == int(a["bepos"])
        nd += int((A[k:k + n] != B[h:h + n]).sum())
        assert nd == int(diffs[i])


def test_trace_pts_matches_oracle_on_pipeline_alignments(toy_pair):
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    dev = D.Device(0)
    ga, gb, dga, dgb, alns, tb = _alignments(dev, ra, rb)
    assert len(alns) > 20 and (alns["flags"] & 1).any() and not (alns["flags"] & 1).all()
    res = D.trace_pts(dev, dga, dgb, alns, tb)
    n = _check_against_oracle(ga, gb, alns, tb, res)
    _replay(ga, gb, alns, res)
    assert n > 1000 and res[4]["panels"] == int((alns["tlen"] // 2).sum())
    # order independence and empty input
    perm = np.random.default_rng(3).permutation(len(alns))
    res2 = D.trace_pts(dev, dga, dgb, alns[perm], tb)
    for q, i in enumerate(perm[:50]):
        assert np.array_equal(res2[3][int(res2[0][q]):int(res2[0][q + 1])], res[3][int(res[0][i]):int(res[0][i + 1])])
    import os
    os.environ["FGA_TRACE_CELLS"] = "20000"                 # many scratch batches instead of one
    try:
        res3 = D.trace_pts(dev, dga, dgb, alns, tb)
    finally:
        del os.environ["FGA_TRACE_CELLS"]
    assert all(np.array_equal(x, y) for x, y in zip(res[:4], res3[:4]))
    empty = D.trace_pts(dev, dga, dgb, alns[:0], tb[:0])
    assert len(empty[1]) == 0 and len(empty[3]) == 0
    dga.free(); dgb.free(); dev.close()


def test_trace_pts_high_divergence_and_missing_revcomp(tmp_path, built_library):
    from fastga_amd import device as D, workload
    ra, rb = workload.build_pair(str(tmp_path), seed=5, ncontig=5, total=300_000, divergence=0.15, inv_frac=0.1)
    dev = D.Device(0)
    ga, gb, dga, dgb, alns, tb = _alignments(dev, ra, rb, aln_rate=0.45)
    assert len(alns) > 5
    res = D.trace_pts(dev, dga, dgb, alns, tb)
    _check_against_oracle(ga, gb, alns, tb, res)
    from fastga_amd.gixio import Gix
    plain = D.DeviceGenome(dev, gb, Gix(rb + ".gix").perm, False)
    if (alns["flags"] & 1).any():
        with pytest.raises(RuntimeError, match="reverse-complement"):
            D.trace_pts(dev, dga, plain, alns, tb)
    bad = alns.copy()                                   # a trace that cannot be realised must be reported, not written
    tb2 = tb.copy()
    k = int(np.argmax(bad["tlen"]))
    tb2[int(bad["toff"][k])] = 0
    tb2[int(bad["toff"][k]) + 1] = 160
    with pytest.raises(RuntimeError, match="inconsistent"):
        D.trace_pts(dev, dga, dgb, bad, tb2)
    for field, value in (("aread", 10**6), ("toff", len(tb)), ("aepos", 2**30)):     # arguments are checked up front
        broken = alns.copy()
        broken[field][0] = value
        with pytest.raises(RuntimeError, match="alignment 0"):
            D.trace_pts(dev, dga, dgb, broken, tb)
    plain.free(); dga.free(); dgb.free(); dev.close()


def test_trace_pts_self_comparison(tmp_path, built_library):
    from fastga_amd import device as D, workload
    ra, _ = workload.build_pair(str(tmp_path), seed=9, ncontig=4, total=400_000, divergence=0.05, repeat_frac=0.15)
    dev = D.Device(0)
    ga, gb, dga, dgb, alns, tb = _alignments(dev, ra, ra, self_cmp=True)
    assert len(alns) > 5
    same = (alns["aread"] == alns["bread"]) & ((alns["flags"] & 1) == 0)
    assert same.any()
    for flag in (False, True):
        res = D.trace_pts(dev, dga, dgb, alns, tb, self_cmp=flag)
        _check_against_oracle(ga, gb, alns, tb, res, self_cmp=flag)
    dga.free(); dgb.free(); dev.close()


def test_trace_pts_replays_at_bench_scale(tmp_path, built_library):
    """20 Mbp pair (a fifth of the bench pair, 0.7 M panels): every script replays to its own difference count, and a
    sample equals the oracle"""
    from fastga_amd import device as D, workload
    ra, rb = workload.build_pair(str(tmp_path), seed=1, ncontig=16, total=20_000_000, divergence=0.02,
                                 repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02, threads=16)
    dev = D.Device(0)
    ga, gb, dga, dgb, alns, tb = _alignments(dev, ra, rb)
    assert int(alns["tlen"].sum()) // 2 > 300_000
    res = D.trace_pts(dev, dga, dgb, alns, tb)
    _replay(ga, gb, alns, res)
    pick = np.random.default_rng(5).choice(len(alns), size=min(40, len(alns)), replace=False)
    sub = alns[pick]
    res2 = D.trace_pts(dev, dga, dgb, sub, tb)
    _check_against_oracle(ga, gb, sub, tb, res2)
    for q, i in enumerate(pick):
        assert np.array_equal(res2[3][int(res2[0][q]):int(res2[0][q + 1])], res[3][int(res[0][i]):int(res[0][i + 1])])
    dga.free(); dgb.free(); dev.close()


# ---------------------------------------------------------------------------------------------------------------------
#  Gap_Improver on the device (fga_trace_pts_regrouped): against the host regrouping, which the CPU suite pins against
#  the oracle and the reference's ALNtoPAF (tests/test_paf_writer.py, tests/test_gap_core.py)
# ---------------------------------------------------------------------------------------------------------------------

def _host_improved(L, ga, gb, alns, tb, res, self_cmp, resume=None):
    from fastga_amd.lib import Alns, Traces
    cp = [np.ascontiguousarray(x).copy() for x in res[:4]]
    rs = None if resume is None else np.ascontiguousarray(resume, dtype=np.int32).copy()
    A = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    T = Traces(len(alns), len(cp[3]), 0, *(x.ctypes.data for x in cp), None if rs is None else rs.ctypes.data)
    assert L.fga_gap_improve(ga.h, None if self_cmp else gb.h, C.byref(A), C.byref(T)) == 0, L.fga_last_error()
    assert rs is None or (rs == -1).all()
    return cp


def _regroup_checks(L, dev, ga, gb, dga, dgb, alns, tb, self_cmp=False, tmax=512):
    import os
    from fastga_amd import device as D
    alns, tb = np.ascontiguousarray(alns), np.ascontiguousarray(tb, dtype=np.uint8)
    plain = D.trace_pts(dev, dga, dgb, alns, tb)
    want = _host_improved(L, ga, gb, alns, tb, plain, self_cmp)
    changed = int((want[3] != plain[3]).sum())
    os.environ["FGA_REGROUP_CAPS"] = f"520,16384,{tmax}"   # 512: the defaults of a large set (a small one goes to the host whole)
    try:
        got = D.trace_pts(dev, dga, dgb, alns, tb, regrouped=True)
    finally:
        del os.environ["FGA_REGROUP_CAPS"]
    rs = got[4]["resume"]
    long_ = plain[1] > tmax                                # scripts one lane would hold the launch for: the host's
    assert (rs[long_] == 0).all() and (rs[~long_] == -1).sum() > 0.8 * (~long_).sum() > 0
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    for i in np.nonzero(rs == -1)[0]:
        o, e = int(want[0][i]), int(want[0][i + 1])
        assert np.array_equal(got[3][o:e], want[3][o:e]) and got[2][i] == want[2][i], i
    fin = _host_improved(L, ga, gb, alns, tb, got, self_cmp, resume=rs)
    assert np.array_equal(fin[3], want[3]) and np.array_equal(fin[2], want[2])
    # the default policy on a set this small: nothing for the device to win, the host regroups all of it
    small = D.trace_pts(dev, dga, dgb, alns, tb, regrouped=True)
    assert (small[4]["resume"][plain[1] >= 2] == 0).all() and np.array_equal(small[3], plain[3])
    # a lane scratch most boxes do not fit, then scripts "too long for one lane": handed back, finished by the host
    back = 0
    for caps in ("3,12,100000", "1024,16384,8"):
        os.environ["FGA_REGROUP_CAPS"] = caps
        try:
            part = D.trace_pts(dev, dga, dgb, alns, tb, regrouped=True)
        finally:
            del os.environ["FGA_REGROUP_CAPS"]
        rs = part[4]["resume"]
        back += int((rs >= 0).sum())
        if caps.endswith(",8"):
            long_ = plain[1] > 8
            assert (rs[long_] == 0).all() and (rs[~long_ & (plain[1] >= 2)] == -1).all()
        fin = _host_improved(L, ga, gb, alns, tb, part, self_cmp, resume=rs)
        assert np.array_equal(fin[3], want[3]) and np.array_equal(fin[2], want[2])
    return changed, back


def test_regrouped_scripts_equal_the_host_gap_improver(toy_pair, tmp_path, built_library):
    from fastga_amd import device as D, workload
    L = built_library
    dev = D.Device(0)
    d, ra, rb = toy_pair
    ga, gb, dga, dgb, alns, tb = _alignments(dev, ra, rb)
    changed, back = _regroup_checks(L, dev, ga, gb, dga, dgb, alns, tb)
    assert changed > 0 and back > 0
    dga.free(); dgb.free()
    # 15 % divergence: crowded boxes, both strands
    ra, rb = workload.build_pair(str(tmp_path), seed=5, ncontig=5, total=600_000, divergence=0.15, inv_frac=0.1)
    ga, gb, dga, dgb, alns, tb = _alignments(dev, ra, rb, aln_rate=0.45)
    changed, back = _regroup_checks(L, dev, ga, gb, dga, dgb, alns, tb, tmax=100000)   # contig-long scripts on lanes
    assert changed > 500 and back > 0
    dga.free(); dgb.free()
    # self comparison (the readers load A and B separately: self flag off)
    sd = str(tmp_path / "self")
    import os
    os.makedirs(sd)
    ra, _ = workload.build_pair(sd, seed=9, ncontig=4, total=400_000, divergence=0.05, repeat_frac=0.15)
    ga, gb, dga, dgb, alns, tb = _alignments(dev, ra, ra, self_cmp=True)
    changed, back = _regroup_checks(L, dev, ga, gb, dga, dgb, alns, tb, self_cmp=True)
    assert changed > 0
    dga.free(); dgb.free(); dev.close()
