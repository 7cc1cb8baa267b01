import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """The in-tree libfastga_amd.so; built on demand (hipcc cross-compiles without a GPU)."""
    from fastga_amd.lib import lib_path, load_library
    if not os.path.exists(lib_path()):
        import __graft_entry__ as g
        g.build()
    return load_library()


@pytest.fixture(scope="session")
def toy_pair(tmp_path_factory, built_library):
    """~0.6 Mbp pair, 12 contigs, 3 % divergence with repeats and rearrangements, built by our own tools."""
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("toy"))
    ra, rb = workload.build_pair(d, seed=11, ncontig=12, total=600_000, divergence=0.03,
                                 repeat_frac=0.05, inv_frac=0.05, swap_frac=0.05)
    return d, ra, rb


@pytest.fixture(scope="session")
def family_pair(tmp_path_factory, built_library):
    """0.4 Mbp pair with a 2 kbp family planted 20 times at 1 % divergence: its k-mers have 10-25 partners, so the
    frequency cutoff decides their fate at -f3, the default 10 and -f30 alike"""
    from fastga_amd import workload, synth
    d = str(tmp_path_factory.mktemp("family"))
    rng = np.random.default_rng(77)
    lens = synth.contig_lengths(21, 8, 400_000)
    A = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in lens]
    fam = rng.integers(0, 4, 2000, dtype=np.uint8)
    for k in range(20):
        c = A[k % len(A)]
        cp = synth.mutate(rng, fam, 0.01)
        if k % 3 == 0:
            cp = synth.revcomp(cp)
        p0 = int(rng.integers(0, len(c) - len(cp) - 1))
        c[p0:p0 + len(cp)] = cp
    B = [synth.mutate(rng, c, 0.02) for c in A]
    return d, workload.build_genome(d, "A", A), workload.build_genome(d, "B", B)


@pytest.fixture(scope="session")
def big_family_pair(tmp_path_factory, built_library):
    """0.5 Mbp pair with a 1.5 kbp family planted 300 times at 0.5 % divergence (both strands): its k-mers have 100-300
    partners, their 12-mer panels exceed a merge tile (streamed windows) and cutoffs of -f64 / -f200 decide their fate
    (the wide-window build of the merge kernel)"""
    from fastga_amd import workload, synth
    d = str(tmp_path_factory.mktemp("bigfam"))
    rng = np.random.default_rng(78)
    lens = synth.contig_lengths(22, 6, 500_000)
    A = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in lens]
    fam = rng.integers(0, 4, 1500, dtype=np.uint8)
    for k in range(300):
        c = A[k % len(A)]
        cp = synth.mutate(rng, fam, 0.005)
        if k % 3 == 0:
            cp = synth.revcomp(cp)
        p0 = int(rng.integers(0, len(c) - len(cp) - 1))
        c[p0:p0 + len(cp)] = cp
    B = [synth.mutate(rng, c, 0.02) for c in A]
    return d, workload.build_genome(d, "A", A), workload.build_genome(d, "B", B)


@pytest.fixture(scope="session")
def huge_family_pair(tmp_path_factory, built_library):
    """0.8 Mbp pair with a 1 kbp family planted 450 times at 0.1 % divergence (a third on the other strand), the second
    genome 0.5 % away: its k-mers have 250-400 partners -- cutoffs between -f256 and -f450 decide their fate (the reference
    takes any -f; here they run on the merge kernel's 4096-entry windows with 64-bit result words)"""
    from fastga_amd import workload, synth
    d = str(tmp_path_factory.mktemp("hugefam"))
    rng = np.random.default_rng(80)
    lens = synth.contig_lengths(24, 6, 800_000)
    A = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in lens]
    fam = rng.integers(0, 4, 1000, dtype=np.uint8)
    for k in range(450):
        c = A[k % len(A)]
        cp = synth.mutate(rng, fam, 0.001)
        if k % 3 == 0:
            cp = synth.revcomp(cp)
        p0 = int(rng.integers(0, len(c) - len(cp) - 1))
        c[p0:p0 + len(cp)] = cp
    B = [synth.mutate(rng, c, 0.005) for c in A]
    return d, workload.build_genome(d, "A", A), workload.build_genome(d, "B", B)


@pytest.fixture(scope="session")
def dense_pair(tmp_path_factory, built_library):
    """3 Mbp of A/C-only sequence against its 3 % diverged copy: the forward-strand k-mers crowd into 4096 of the 2^24
    12-mer panels (and the complement ones into another 4096), ~150 entries each and many beyond a tile -- the panel
    shapes of a 3 Gbp table at toy size, so the streamed windows of the merge kernel meet the oracle entry for entry"""
    from fastga_amd import workload, synth
    d = str(tmp_path_factory.mktemp("dense"))
    rng = np.random.default_rng(79)
    lens = synth.contig_lengths(23, 6, 3_000_000)
    A = [rng.integers(0, 2, int(L), dtype=np.uint8) for L in lens]          # a / c only
    # a stretch of very low complexity on top: period-7 tandem, panels of thousands of near-identical k-mers
    unit = rng.integers(0, 2, 7, dtype=np.uint8)
    A[0][1000:41000] = synth.mutate(rng, np.tile(unit, 6000), 0.01)[:40000]
    B = [synth.mutate(rng, c, 0.03) for c in A]
    # lower-case stretches in both genomes: the index carries mask bytes, so -M also decides inside streamed windows
    mA = [np.zeros(len(c), dtype=bool) for c in A]
    mB = [np.zeros(len(c), dtype=bool) for c in B]
    for ms in (mA, mB):
        for m in ms:
            for _ in range(40):
                s0 = int(rng.integers(0, max(1, len(m) - 4000)))
                m[s0:s0 + int(rng.integers(100, 4000))] = True
    return (d, workload.build_genome(d, "A", A, masks=mA, use_mask=True),
            workload.build_genome(d, "B", B, masks=mB, use_mask=True))


@pytest.fixture(scope="session")
def masked_pair(tmp_path_factory, built_library):
    """~0.6 Mbp pair whose repeat copies are lower case in BOTH genomes; indices carry the mask bytes (host producer,
    pinned against `GIXmake -T1 ... #` by tests/test_edge_cases.py)"""
    from fastga_amd import workload, synth
    d = str(tmp_path_factory.mktemp("masked"))
    lens = synth.contig_lengths(9, 10, 600_000)
    A, mA, B, mB = synth.make_pair(9, lens, 0.03, repeat_frac=0.15, inv_frac=0.05, swap_frac=0.05)
    rng = np.random.default_rng(3)
    for m in mB:                                        # B: arbitrary lower-case stretches as well
        for _ in range(6):
            s0 = int(rng.integers(0, max(1, len(m) - 3000)))
            m[s0:s0 + int(rng.integers(200, 3000))] = True
    ra = workload.build_genome(d, "A", A, masks=mA, use_mask=True)
    rb = workload.build_genome(d, "B", B, masks=mB, use_mask=True)
    return d, ra, rb
