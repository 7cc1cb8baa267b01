"""GPU parity: the device chain scan (thread-per-record kernels -- workgroup-tiled for dense, straight from HBM for sparse
key streams, every workgroup size -- + wave-parallel kernels) against oracle/chain_oracle.c -- the
sequential restatement of the reference's scan (align_contigs, FastGA.c:3016-3176) that tests/test_chain_oracle.py pins
to the hit boxes a DEBUG_HIT build of the reference prints -- on the same sorted records: bit-exact hits and units (the
product's host scan, fga_chain_scan, is compared too).  The small/long unit threshold is varied so that every unit also
goes through the wave-parallel kernel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scan_both(ra, rb, limits, chain_min=170, chain_break=2000):
    from fastga_amd.gixio import Gix, Gdb
    from fastga_amd import device as D
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    ga, gb = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    amx, bmx = int(ga.maxctg), int(gb.maxctg)
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    seeds = D.seed_merge(dev, dA, dB)
    keys = D.seed_sort(dev, seeds, amx, bmx, A.nctg, B.nctg)
    seeds.free()
    alen_sorted = ga.clen[A.perm]
    ref = D.chain_scan(keys.download(), (keys.wa, keys.wb, keys.wd, keys.wt), chain_break, chain_min, amx, bmx,
                       alen_sorted, nthreads=4)
    ru, rh = ref.units, ref.hits
    _equals_oracle(keys, ru, rh, chain_break, chain_min, amx, bmx, alen_sorted)
    out = []
    for lim in limits:
        blk = None                                   # (limit, workgroup size): 0 = the kernel for sparse key streams
        if isinstance(lim, tuple):
            lim, blk = lim
        if lim is None:
            os.environ.pop("FGA_CHAIN_SMALL_LIMIT", None)
        else:
            os.environ["FGA_CHAIN_SMALL_LIMIT"] = str(lim)
        if blk is not None:
            os.environ["FGA_CHAIN_BLOCK"] = str(blk)
        try:
            got = D.chain_scan_device(dev, keys, chain_break, chain_min, amx, bmx, alen_sorted)
        finally:
            os.environ.pop("FGA_CHAIN_SMALL_LIMIT", None)
            os.environ.pop("FGA_CHAIN_BLOCK", None)
        out.append((lim, got.units, got.hits))
        got.free()
    ref.free(); keys.free(); dA.free(); dB.free(); dev.close()
    return ru, rh, out


def _equals_oracle(keys, units, hits, chain_break, chain_min, amx, bmx, alen_sorted):
    """units + hits (device or host product scan) == the pinned oracle's rows on the device's own sorted keys"""
    from oracle import harness as H
    f = H.unpack_keys(keys.download(), keys.wa, keys.wb, keys.wd, keys.wt)
    rows = H.oracle_chain_scan(f, chain_break, chain_min, amx, bmx, alen_sorted)
    exp = [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[5]), int(r[6]), int(r[7]), int(r[8]), int(r[9]))
           for r in rows]
    got = []
    for x in units:
        for q in range(int(x["nhits"])):
            y = hits[int(x["first_hit"]) + q]
            got.append((int(x["comp"]), int(x["actg"]), int(x["bctg"]), int(x["bucket"]), int(y["cov"]),
                        int(y["dgmin"]), int(y["dgmax"]), int(y["alow"]), int(y["ahgh"])))
    assert len(exp) > 0 and got == exp


@pytest.mark.parametrize("chain_min", [170, 100])
def test_device_chain_scan_matches_host(toy_pair, chain_min):
    d, ra, rb = toy_pair
    ru, rh, out = _scan_both(ra, rb, [None, 3, 0, (None, 0), (None, 256), (None, 1024), (5, 0), (5, 1024)], chain_min=chain_min)
    assert len(rh) > 20
    for lim, u, h in out:
        assert len(u) == len(ru) and len(h) == len(rh), (lim, len(u), len(ru), len(h), len(rh))
        assert np.array_equal(u, ru), lim
        assert np.array_equal(h, rh), lim


def test_device_chain_scan_self(toy_pair):
    """self comparison: every contig pairs with itself, which gives the longest buckets"""
    d, ra, rb = toy_pair
    ru, rh, out = _scan_both(ra, ra, [None, 0, (None, 0), (None, 256), (7, 1024)])
    assert len(rh) > 0
    for lim, u, h in out:
        assert np.array_equal(u, ru) and np.array_equal(h, rh), lim


def test_chain_scan_is_independent_of_tie_order(toy_pair):
    """fga_seed_sort with anti_order_only leaves records of equal (strand, contigs, bucket, anti) in arrival order (two
    radix passes fewer); the chain scan -- device and host -- must give exactly the hits of the fully sorted records"""
    from fastga_amd.gixio import Gix, Gdb
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    ga, gb = Gdb(ra + ".gdb"), Gdb(rb + ".gdb")
    amx, bmx = int(ga.maxctg), int(gb.maxctg)
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    alen_sorted = ga.clen[A.perm]
    res = []
    for partial in (False, True):
        seeds = D.seed_merge(dev, dA, dB)
        keys = D.seed_sort(dev, seeds, amx, bmx, A.nctg, B.nctg, anti_order_only=partial)
        seeds.free()
        k = keys.download()
        if partial:       # sorted on everything above the low 12 bits
            hi, lo = k["hi"], k["lo"] >> np.uint64(12)
            assert np.all((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] >= lo[:-1])))
        dv = D.chain_scan_device(dev, keys, 2000, 170, amx, bmx, alen_sorted)
        ho = D.chain_scan(k, (keys.wa, keys.wb, keys.wd, keys.wt), 2000, 170, amx, bmx, alen_sorted, nthreads=4)
        res.append((dv.units, dv.hits, ho.units, ho.hits))
        dv.free(); ho.free(); keys.free()
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(x, y)
    assert np.array_equal(res[0][0], res[0][2]) and np.array_equal(res[0][1], res[0][3])
    dA.free(); dB.free(); dev.close()
