"""GPU parity: chain scan + wave-extension kernel vs the CPU oracle's Local_Alignment, replaying the reference's
per-unit extension loop (FastGA.c:3227-3341) in Python around the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def replay_unit(H, spec, aseq_pad, bseq_pad, comp, hits, aln_min, aln_rate):
    """the reference's hit loop for one unit; returns list of (abpos,bbpos,aepos,bepos,diffs,trace bytes)."""
    alen, blen = len(aseq_pad) - 2, len(bseq_pad) - 2
    mlen = alen + blen
    out = []
    alast = -1
    for h in hits:
        dgmin, dgmax, alow, ahgh = int(h["dgmin"]), int(h["dgmax"]), int(h["alow"]), int(h["ahgh"])
        if ahgh <= alast:
            continue
        if alow < alast:
            alow = alast
        ahgh -= 128
        while True:
            amid = alow + 128
            if amid > ahgh:
                amid = ahgh
                if amid + dgmin < 0:
                    dgmin = -amid
                    if dgmin > dgmax:
                        break
            r = H.oracle_local_alignment(aseq_pad, bseq_pad, spec, dgmin, dgmax, amid, -1, -1, acomp=bool(comp))
            abpos, bbpos, aepos, bepos, diffs, tr = r
            rlen = aepos - abpos
            if rlen >= aln_min and aln_rate * rlen >= diffs:
                out.append((abpos, bbpos, aepos, bepos, diffs, (tr & 0xff).astype(np.uint8)))
            eant = mlen - (abpos + bbpos) if comp else aepos + bepos
            alow = amid if eant <= alow else eant
            if not alow < ahgh:
                break
        alast = alow
    return out


@pytest.mark.parametrize("build", ["latency", "throughput"])
def test_extension_matches_oracle(toy_pair, build, monkeypatch):
    """build = throughput: FGA_EXTEND_NARROW=1 sends the same units through ext_mid (one-copy ring updated in place, waves of
    61 .. 124 diagonals in two register blocks, the handoff of wider ones to ext_full), which the pipeline only picks for
    10^4 units and more"""
    from fastga_amd.gixio import Gix, Gdb
    if build == "throughput":
        monkeypatch.setenv("FGA_EXTEND_NARROW", "1")
    from fastga_amd import device as D, synth
    from oracle import harness as H
    d, ra, rb = toy_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    ga, gb = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    amx, bmx = int(ga.maxctg), int(gb.maxctg)
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    seeds = D.seed_merge(dev, dA, dB)
    keys = D.seed_sort(dev, seeds, amx, bmx, A.nctg, B.nctg)
    kh = keys.download()
    alen_sorted = ga.clen[A.perm]
    hl = D.chain_scan(kh, (keys.wa, keys.wb, keys.wd, keys.wt), 2000, 170, amx, bmx, alen_sorted, nthreads=4)
    assert hl.nhits > 10
    freq = (C := __import__("ctypes")).c_float * 4
    f4 = freq()
    ga.L.fga_gdb_freq(ga.h, f4)
    pa, table, score = D.align_spec(0.7, 100, list(f4))
    dga = D.DeviceGenome(dev, ga, A.perm, True)
    dgb = D.DeviceGenome(dev, gb, B.perm, False)
    alns, tb, stats = D.extend(dev, dga, dgb, hl, pa, table, score, aln_min=50, aln_rate=0.35)
    assert len(alns) > 5 and stats["waves"] > 1000

    spec = H.oracle_spec(0.7, 100, tuple(f4), 0)
    assert spec.ave_path == pa
    assert np.array_equal(np.ctypeslib.as_array(spec.table), table)
    units, hits = hl.units, hl.hits
    by_unit = {}
    for a in alns:
        by_unit.setdefault(int(a["unit"]), []).append(a)
    cache = {}
    total = 0
    for u, U in enumerate(units):
        c1, c2 = int(A.perm[U["actg"]]), int(B.perm[U["bctg"]])
        comp = int(U["comp"])
        ka = (c1, comp)
        if ka not in cache:
            s = ga.contig(c1)
            cache[ka] = H.pad_seq(synth.revcomp(s) if comp else s)
        kb = ("b", c2)
        if kb not in cache:
            cache[kb] = H.pad_seq(gb.contig(c2))
        exp = replay_unit(H, spec, cache[ka], cache[kb], comp,
                          hits[U["first_hit"]:U["first_hit"] + U["nhits"]], 50, 0.35)
        got = sorted(by_unit.get(u, []), key=lambda a: int(a["seq"]))
        assert len(got) == len(exp), (u, len(got), len(exp))
        for g, e in zip(got, exp):
            assert (int(g["abpos"]), int(g["bbpos"]), int(g["aepos"]), int(g["bepos"]), int(g["diffs"])) == e[:5]
            assert int(g["aread"]) == c1 and int(g["bread"]) == c2 and int(g["flags"]) == comp
            t = tb[int(g["toff"]):int(g["toff"]) + int(g["tlen"])]
            assert np.array_equal(t, e[5])
        total += len(exp)
    assert total == len(alns)
    dga.free(); dgb.free(); keys.free(); seeds.free(); dA.free(); dB.free(); dev.close()
