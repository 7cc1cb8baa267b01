"""Tables of more than 2^32 entries (the reference indexes with int64, libfastk.c:785-907; GIXmake.c:1888-1901 sizes payloads
freely).  A 6 Gbp genome's index holds 4.76 G entries: the view's prefix index keeps the LOW words of the cumulative counts and
the prefixes at which they pass a multiple of 2^32 (fga_view.car); differences inside a tile are u32 arithmetic that wraps with
them, absolute positions add the carries.  Checked where it matters -- at the start of the table, just before, across and
after the prefix of the carry, and at the table's end -- against the pinned oracle on the host copy of the same tables, in the
three modes; and the whole comparison of a 376 Mbp genome against the 6 Gbp one against the real reference's digest
(tests/golden/config6g_pair_digest.json, made by tools/config6g_check.py --reference on the GPU box)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _host_gib():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 2**20
    except OSError:
        pass
    return 0.0


@pytest.fixture(scope="module")
def big(tmp_path_factory, built_library):
    from fastga_amd import workload, device as D
    dev = D.Device(0)
    hbm = dev.L.fga_dev_available(dev.h)
    dev.close()
    if _host_gib() < 200 or hbm < 240 * 2**30:
        pytest.skip("needs 200 GB of host memory and 240 GB of free device memory")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = str(tmp_path_factory.mktemp("six_gbp")) if base is None else os.path.join(base, "fga_six_gbp_%d" % os.getpid())
    os.makedirs(d, exist_ok=True)
    rg, rs = workload.build_config6g(d, threads=32)
    yield d, rg, rs
    import shutil
    shutil.rmtree(d, ignore_errors=True)


def test_merge_across_the_2_32_carry_equals_the_oracle(big):
    from fastga_amd import device as D
    from fastga_amd.gixio import Gdb
    from oracle import harness as H
    d, rg, rs = big
    dev = D.Device(0)
    gG, gS = Gdb(rg + ".gdb"), Gdb(rs + ".gdb")
    dS, xS = D.build_gix_device(dev, gS, 32, host_copy=True)
    dG, xG = D.build_gix_device(dev, gG, 32, host_copy=True)
    assert xG.nents > 2**32
    pc = int(np.searchsorted(xG.index, 2**32, side="left"))         # the first prefix whose cumulative count reaches 2^32
    assert 0 < pc < (1 << 24) - 200
    NP = 1 << 24
    ranges = [(0, 48), (pc - 90, pc - 40), (pc - 24, pc + 24), (pc + 40, pc + 90), (NP - 48, NP)]
    w = 1 + xS.pbyte + xG.pbyte
    for p0, p1 in ranges:
        # pair: the small genome's table against the big one's (table 2 beyond 2^32)
        n, c, nh, ts = H.oracle_seed_merge(xS.table, xS.index, xS.pbyte, xG.table, xG.index, xG.pbyte, pfirst=p0, plast=p1)
        s = D.seed_merge(dev, dS, dG, prefix_begin=p0, prefix_end=p1)
        got = s.download(); s.free()
        gn, gc = D.seeds_to_reference_bytes(got, xS.postbytes, xS.contbytes, xG.postbytes, xG.contbytes)
        assert len(got) == nh and nh > 0, (p0, p1, len(got), nh)
        assert np.array_equal(H.sorted_records(gn, w), H.sorted_records(n, w)) and \
            np.array_equal(H.sorted_records(gc, w), H.sorted_records(c, w)), ("pair", p0, p1)
        # FLIP (-S second pass): the big table as table 1
        n, c, nh, ts = H.oracle_seed_merge(xG.table, xG.index, xG.pbyte, xS.table, xS.index, xS.pbyte, flip=True, pfirst=p0, plast=p1)
        s = D.seed_merge(dev, dG, dS, flip=True, prefix_begin=p0, prefix_end=p1)
        got = s.download(); s.free()
        gn, gc = D.seeds_to_reference_bytes(got, xS.postbytes, xS.contbytes, xG.postbytes, xG.contbytes)
        assert len(got) == nh, ("flip", p0, p1, len(got), nh)
        assert np.array_equal(H.sorted_records(gn, w), H.sorted_records(n, w)) and \
            np.array_equal(H.sorted_records(gc, w), H.sorted_records(c, w)), ("flip", p0, p1)
    w2 = 1 + 2 * xG.pbyte
    for p0, p1 in [(0, 8), (pc - 4, pc + 4), (NP - 8, NP)]:
        # self: one table beyond 2^32 (the oracle halves its total as the reference does: compare the records)
        n, c, nh, ts = H.oracle_self_seed_merge(xG.table, xG.index, xG.pbyte, pfirst=p0, plast=p1)
        s = D.seed_merge(dev, dG, None, prefix_begin=p0, prefix_end=p1)
        got = s.download(); s.free()
        gn, gc = D.seeds_to_reference_bytes(got, xG.postbytes, xG.contbytes, xG.postbytes, xG.contbytes)
        assert np.array_equal(H.sorted_records(gn, w2), H.sorted_records(n, w2)) and \
            np.array_equal(H.sorted_records(gc, w2), H.sorted_records(c, w2)), ("self", p0, p1)
    dS.free(); dG.free(); xS.close(); xG.close(); gG.close(); gS.close(); dev.close()


def test_comparison_against_a_6_gbp_genome_is_the_reference_s(big):
    from fastga_amd import device as D, workload
    from oracle import harness as H
    gold = os.path.join(HERE, "golden", "config6g_pair_digest.json")
    if not os.path.exists(gold) or not os.path.exists(H.ref_bin("ONEview")):
        pytest.skip("no golden digest / oracle/_ref/ONEview did not travel")
    g = json.load(open(gold))
    d, rg, rs = big
    out = os.path.join(d, "pair.1aln")
    st = D.run(rs, rg, out, nthreads=32, reference_threads=32)
    assert (st["nseeds"], st["nhits"], st["nalns"], st["nlive"]) == (g["total_seeds"], g["hits"], g["alignments"], g["records"])
    got = workload.digest_1aln_stream(out, H.ref_bin("ONEview"))
    assert all(got[k] == g[k] for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5"))
