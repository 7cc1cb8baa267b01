"""GPU end to end: our FastGA hot path vs the REAL reference FastGA (oracle/_ref) on the same GDB/GIX files:
identical `.1aln` content as printed by the reference's own ONEview, minus provenance ('!') and path ('<') lines
(SURVEY.md hard part 10)."""
import os

import pytest

pytestmark = pytest.mark.gpu


def _compare(ra, rb, workdir, **kw):
    from fastga_amd import device as D
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    ours = os.path.join(workdir, "ours.1aln")
    st = D.run(ra, rb, ours, nthreads=8, **kw)
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
    assert st["nlive"] > 0
    assert len(a) == len(b), (len(a), len(b), st)
    for x, y in zip(a, b):
        assert x == y
    return st


def test_pair_default_matches_reference(toy_pair, tmp_path):
    d, ra, rb = toy_pair
    st = _compare(ra, rb, str(tmp_path))
    assert st["nhits"] >= st["nlive"]


def test_pair_symmetric_and_options(toy_pair, tmp_path):
    d, ra, rb = toy_pair
    _compare(ra, rb, str(tmp_path), symmetric=True, freq=6, identity=0.8, chain_min=60)


def test_divergent_pair_matches_reference(tmp_path, built_library):
    from fastga_amd import workload
    d = str(tmp_path)
    ra, rb = workload.build_pair(d, seed=77, ncontig=10, total=800_000, divergence=0.10,
                                 repeat_frac=0.10, inv_frac=0.05, swap_frac=0.05)
    _compare(ra, rb, d)
