"""GPU parity: HIP seed merge (through the C-ABI) vs the CPU oracle, multiset-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _compare(dev, A, B, dA, dB, **kw):
    from fastga_amd import device as D
    from oracle import harness as H
    flip = kw.get("flip", False)
    seeds = D.seed_merge(dev, dA, dB, **kw)
    got = seeds.download()
    seeds.free()
    if B is None:
        n, c, nh, ts = H.oracle_self_seed_merge(A.table, A.index, A.pbyte, freq=kw.get("freq", 10),
                                                soft_mask=kw.get("soft_mask", False))
        G1, G2 = A, A
        # the oracle reports the reference's halved total; a seed and its mirror need not both exist (plen and the
        # frequency test are taken from the T1 entry's point of view), so the full count can be odd
        assert len(got) in (2 * nh, 2 * nh + 1)
    else:
        n, c, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte,
                                           freq=kw.get("freq", 10), soft_mask=kw.get("soft_mask", False),
                                           flip=flip)
        G1, G2 = (B, A) if flip else (A, B)
        assert len(got) == nh
        assert seeds.plen_sum == ts
    gn, gc = D.seeds_to_reference_bytes(got, G1.postbytes, G1.contbytes, G2.postbytes, G2.contbytes)
    w = 1 + G1.pbyte + G2.pbyte
    assert np.array_equal(H.sorted_records(gn, w), H.sorted_records(n, w))
    assert np.array_equal(H.sorted_records(gc, w), H.sorted_records(c, w))
    return len(got)


@pytest.fixture(scope="module")
def loaded(toy_pair):
    from fastga_amd.gixio import Gix
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    yield dev, A, B, dA, dB
    dA.free(); dB.free(); dev.close()


def test_pair_merge(loaded):
    dev, A, B, dA, dB = loaded
    assert _compare(dev, A, B, dA, dB) > 1000


def test_pair_merge_flip(loaded):
    dev, A, B, dA, dB = loaded
    # -S second pass: table 1 is genome 2
    assert _compare(dev, B, A, dB, dA, flip=True) > 1000


def test_pair_merge_freq_and_mask(loaded):
    dev, A, B, dA, dB = loaded
    _compare(dev, A, B, dA, dB, freq=3)
    _compare(dev, A, B, dA, dB, freq=50, soft_mask=True)


def test_soft_mask_modes_with_real_mask_bytes(masked_pair):
    """-M on tables that carry mask bytes: pair, the flipped pass of -S, and the self comparison (BASELINE config 3's
    mode, new_self_merge_thread with mlen = plen, FastGA.c:1791-1799); the oracle's branches are pinned to the
    reference's seed files in tests/test_oracle_vs_reference.py"""
    from fastga_amd.gixio import Gix
    from fastga_amd import device as D
    d, ra, rb = masked_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    plain = _compare(dev, A, B, dA, dB)
    assert 0 < _compare(dev, A, B, dA, dB, soft_mask=True) < plain
    _compare(dev, B, A, dB, dA, flip=True, soft_mask=True)
    sp = _compare(dev, A, None, dA, None)
    assert 0 < _compare(dev, A, None, dA, None, soft_mask=True) < sp
    _compare(dev, A, None, dA, None, soft_mask=True, freq=4)
    dA.free(); dB.free(); dev.close()


def test_frequency_cutoff_on_a_repeat_family(family_pair):
    from fastga_amd.gixio import Gix
    from fastga_amd import device as D
    d, ra, rb = family_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    n3, n10, n30 = (_compare(dev, A, B, dA, dB, freq=f) for f in (3, 10, 30))
    assert n3 < n10 < n30
    _compare(dev, A, None, dA, None, freq=30)
    _compare(dev, B, A, dB, dA, flip=True, freq=30)
    dA.free(); dB.free(); dev.close()


def test_large_cutoffs_on_a_300_copy_family(big_family_pair):
    """-f64 and -f200 (the wide-window build: the sub-tile margin FREQ+2 no longer fits a 256-entry window) on panels of
    hundreds of entries, pair / flipped / self, against the pinned oracle"""
    from fastga_amd.gixio import Gix
    from fastga_amd import device as D
    d, ra, rb = big_family_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    n10, n64, n200 = (_compare(dev, A, B, dA, dB, freq=f) for f in (10, 64, 200))
    assert n10 < n64 < n200
    _compare(dev, A, B, dA, dB, freq=400)                     # (above 255: the 4096-entry windows, see the next test)
    _compare(dev, B, A, dB, dA, flip=True, freq=64)
    _compare(dev, B, A, dB, dA, flip=True, freq=200)
    _compare(dev, A, None, dA, None, freq=64)
    _compare(dev, A, None, dA, None, freq=255)
    _compare(dev, A, B, dA, dB, freq=120, soft_mask=True)
    dA.free(); dB.free(); dev.close()


def test_cutoffs_above_255_on_a_450_copy_family(huge_family_pair):
    """-f256 .. -f450 where they matter: k-mers with 250-400 partners (the reference takes any -f, FastGA.c:4497-4499; the
    kernel runs them on 4096-entry windows with 64-bit result words), pair / flipped / self / soft-masked, against the oracle"""
    from fastga_amd.gixio import Gix
    from fastga_amd import device as D
    d, ra, rb = huge_family_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    n255, n300, n450 = (_compare(dev, A, B, dA, dB, freq=f) for f in (255, 300, 450))
    assert n255 < n300 < n450, (n255, n300, n450)
    _compare(dev, B, A, dB, dA, flip=True, freq=330)
    _compare(dev, A, None, dA, None, freq=330)
    _compare(dev, A, B, dA, dB, freq=340, soft_mask=True)
    dA.free(); dB.free(); dev.close()


def test_dense_panels_are_streamed_in_windows(dense_pair):
    """every panel holds ~150 entries and many more than a tile: windows of the T2 panel, margins, the skip over windows
    without a T1 key, the self comparison's runs with margins -- all modes against the oracle"""
    from fastga_amd.gixio import Gix
    from fastga_amd import device as D
    d, ra, rb = dense_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    dev = D.Device(0)
    dA, dB = dev.upload(A), dev.upload(B)
    assert _compare(dev, A, B, dA, dB) > 100_000
    _compare(dev, A, B, dA, dB, freq=3)
    _compare(dev, A, B, dA, dB, freq=40)
    _compare(dev, A, B, dA, dB, freq=100)
    _compare(dev, B, A, dB, dA, flip=True)
    _compare(dev, B, A, dB, dA, flip=True, freq=100)
    _compare(dev, A, None, dA, None)
    _compare(dev, A, None, dA, None, freq=100)
    _compare(dev, A, None, dA, None, freq=255)                    # the wide-window build on panels of thousands
    for f in (256, 1000, 1982):                                   # the 4096-entry windows, up to the largest cutoff they hold
        _compare(dev, A, B, dA, dB, freq=f)
    _compare(dev, B, A, dB, dA, flip=True, freq=700)
    _compare(dev, A, None, dA, None, freq=1982)
    # beyond the largest window: the window-free kernel (any cutoff, like the reference), every mode
    for f in (1983, 2500, 6000):
        _compare(dev, A, B, dA, dB, freq=f)
    _compare(dev, B, A, dB, dA, flip=True, freq=2500)
    _compare(dev, A, None, dA, None, freq=2500)
    _compare(dev, A, B, dA, dB, freq=3000, soft_mask=True)
    _compare(dev, B, A, dB, dA, flip=True, freq=3000, soft_mask=True)
    _compare(dev, A, None, dA, None, freq=100000, soft_mask=True)
    _compare(dev, A, B, dA, dB, freq=600, soft_mask=True)
    # the tables carry mask bytes: -M inside the streamed windows, all modes
    plain = _compare(dev, A, B, dA, dB, freq=20)
    assert 0 < _compare(dev, A, B, dA, dB, freq=20, soft_mask=True) < plain
    _compare(dev, B, A, dB, dA, flip=True, soft_mask=True)
    _compare(dev, A, None, dA, None, soft_mask=True)
    _compare(dev, A, B, dA, dB, freq=200, soft_mask=True)
    dA.free(); dB.free(); dev.close()


def test_empty_prefix_range_is_an_empty_shard(loaded):
    """a prefix cut that repeats (one panel heavier than a shard's share) gives an empty range: no seeds, no error;
    (0,0) stays "everything" and fga_merge_prefix_cuts never produces it as a shard"""
    from fastga_amd import device as D
    dev, A, B, dA, dB = loaded
    s = D.seed_merge(dev, dA, dB, prefix_begin=5000, prefix_end=5000)
    assert s.count == 0 and len(s.download()) == 0
    s.free()
    full = D.seed_merge(dev, dA, dB)
    n = full.count
    full.free()
    s = D.seed_merge(dev, dA, dB, prefix_begin=0, prefix_end=0)
    assert s.count == n
    s.free()


def test_self_merge(loaded):
    dev, A, B, dA, dB = loaded
    _compare(dev, A, None, dA, None)


def test_prefix_range_shards_union_is_full(loaded):
    """multi-GPU phase-1 sharding: per-range launches produce disjoint seed sets whose union is the full set."""
    from fastga_amd import device as D
    from fastga_amd.parallel import prefix_shards
    dev, A, B, dA, dB = loaded
    full = D.seed_merge(dev, dA, dB)
    allseeds = np.sort(full.download().view(np.dtype((np.void, 16))))
    full.free()
    parts = []
    for b, e in prefix_shards(A.index, B.index, 3):
        s = D.seed_merge(dev, dA, dB, prefix_begin=b, prefix_end=e)
        parts.append(s.download())
        s.free()
    got = np.sort(np.concatenate(parts).view(np.dtype((np.void, 16))))
    assert np.array_equal(got, allseeds)
