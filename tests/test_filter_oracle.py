"""The host redundancy filter (fga_filter.c: shared-endpoint rule, path_gap, fuse, containment) against the CPU
restatement of the reference's sweeps (oracle/filter_oracle.py) on random groups of overlapping records that share
end points, run along a common diagonal, meet on trace points and contain one another."""
import ctypes as C

import numpy as np


def _random_group(rng):
    from fastga_amd.device import ALN_DTYPE
    n = int(rng.integers(2, 40))
    recs = np.zeros(n, ALN_DTYPE)
    tbs, off = [], 0
    base_d = int(rng.integers(-50, 50))
    for i in range(n):
        ab = int(rng.integers(0, 3000))
        ae = ab + int(rng.integers(60, 2500))
        j = int(rng.integers(0, max(i, 1)))
        if rng.random() < 0.3 and i > 0:                  # share a start or an end with an earlier record
            if rng.random() < 0.5:
                ab = int(recs[j]["abpos"])
                ae = max(ab + 60, ae)
            else:
                ae = int(recs[j]["aepos"])
                ab = max(0, min(ab, ae - 60))
        npan = (ae - 1) // 100 - ab // 100 + 1
        bl = []
        for p in range(npan):
            a0 = max(ab, (ab // 100 + p) * 100)
            a1 = min(ae, (ab // 100 + p + 1) * 100)
            d = int(rng.integers(-2, 3)) if rng.random() < 0.3 else 0
            bl.append(max(0, min(255, (a1 - a0) + d)))
        bb = max(0, ab + base_d + int(rng.integers(-3, 4)) * (rng.random() < 0.4))
        if rng.random() < 0.3 and i > 0 and recs[j]["abpos"] == ab:
            bb = int(recs[j]["bbpos"])
        be = bb + sum(bl)
        if rng.random() < 0.3 and i > 0 and recs[j]["aepos"] == ae:
            be = int(recs[j]["bepos"])
            bb = be - sum(bl)
            if bb < 0:
                bb, be = 0, sum(bl)
        tr = np.zeros(2 * npan, np.uint8)
        tr[1::2] = bl
        tr[0::2] = rng.integers(0, 6, npan)
        recs[i] = (2 * npan, int(tr[0::2].sum()), ab, bb, ae, be, 0, 0, 0, i // 3, i % 3, 0, off)
        tbs.append(tr)
        off += 2 * npan
    return recs, np.concatenate(tbs)


def test_filter_equals_oracle_restatement(built_library):
    from fastga_amd.lib import Alns
    from fastga_amd.device import ALN_DTYPE
    from oracle.filter_oracle import Rec, filter_group
    L = built_library
    rng = np.random.default_rng(20260926)
    dropped = fused = 0
    for _ in range(400):
        recs, tb = _random_group(rng)
        A = Alns(len(recs), len(tb), 0, 0, recs.ctypes.data, tb.ctypes.data)
        out = C.POINTER(Alns)()
        assert L.fga_filter_alignments_mt(C.byref(A), 1, C.byref(out)) == 0
        o = out.contents
        got = np.frombuffer((C.c_char * (o.naln * ALN_DTYPE.itemsize)).from_address(o.alns), dtype=ALN_DTYPE).copy()
        gt = np.frombuffer((C.c_char * max(o.ntrace, 1)).from_address(o.tbytes), dtype=np.uint8)[:o.ntrace].copy()
        L.fga_alns_free(out)
        exp = filter_group([Rec(int(r["abpos"]), int(r["bbpos"]), int(r["aepos"]), int(r["bepos"]), int(r["diffs"]),
                                tb[int(r["toff"]):int(r["toff"]) + int(r["tlen"])].tolist(), i)
                            for i, r in enumerate(recs)])
        assert len(got) == len(exp)
        for g, e in zip(got, exp):
            assert (int(g["abpos"]), int(g["bbpos"]), int(g["aepos"]), int(g["bepos"]), int(g["diffs"])) == \
                   (e.abpos, e.bbpos, e.aepos, e.bepos, e.diffs)
            assert gt[int(g["toff"]):int(g["toff"]) + int(g["tlen"])].tolist() == e.trace
        dropped += len(recs) - len(got)
        boxes = {(int(r["abpos"]), int(r["aepos"])) for r in recs}
        fused += sum(1 for g in got if (int(g["abpos"]), int(g["aepos"])) not in boxes)
    assert dropped > 1000 and fused > 100          # the stress really exercises elimination and fusing


def test_filter_is_the_same_for_any_number_of_threads(built_library):
    """above 50,000 records the driver's passes (discovery order, record build, final order, copy out) run on a team
    of threads: the result must not depend on the team's size"""
    from fastga_amd.lib import Alns
    from fastga_amd.device import ALN_DTYPE
    L = built_library
    rng = np.random.default_rng(7)
    parts, tbs, off, unit = [], [], 0, 0
    while sum(len(p) for p in parts) < 60000:
        recs, tb = _random_group(rng)
        recs["aread"] = unit % 37
        recs["bread"] = (unit // 37) % 5
        recs["unit"] = unit
        recs["seq"] = np.arange(len(recs))
        recs["toff"] += off
        parts.append(recs); tbs.append(tb); off += len(tb); unit += 1
    recs = np.concatenate(parts)
    tb = np.concatenate(tbs)
    perm = rng.permutation(len(recs))              # the extension kernel delivers records in any order
    recs = np.ascontiguousarray(recs[perm])
    res = []
    for nt in (1, 3, 16):
        A = Alns(len(recs), len(tb), 0, 0, recs.ctypes.data, tb.ctypes.data)
        out = C.POINTER(Alns)()
        assert L.fga_filter_alignments_mt(C.byref(A), nt, C.byref(out)) == 0
        o = out.contents
        got = np.frombuffer((C.c_char * (o.naln * ALN_DTYPE.itemsize)).from_address(o.alns), dtype=ALN_DTYPE).copy()
        gt = np.frombuffer((C.c_char * max(o.ntrace, 1)).from_address(o.tbytes), dtype=np.uint8)[:o.ntrace].copy()
        L.fga_alns_free(out)
        res.append((got.tobytes(), gt.tobytes()))
    assert 0 < len(res[0][0]) < recs.nbytes
    assert res[0] == res[1] == res[2]
