"""GPU: the hot path on the golden inputs (tests/golden) must print what the reference printed -- no reference build
needed on the box: `.1aln` (ASCII form, equal to the reference's ONEview text), PAF with CIGARs, PSL, seed totals."""
import json
import os

import pytest

from tests.test_golden import GOLD, golden_roots, golden_lines

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("device_index", [False, True])
def test_hot_path_reproduces_the_golden_files(tmp_path, built_library, monkeypatch, device_index):
    from fastga_amd import device as D
    w = str(tmp_path)
    ra, rb = golden_roots(w, gix=not device_index)           # without .gix files the indices are built on the device
    stats = json.load(open(os.path.join(GOLD, "toy_stats.json")))
    monkeypatch.setenv("FGA_ALN_ASCII", "1")
    out, paf = os.path.join(w, "o.1aln"), os.path.join(w, "o.paf")
    for tag, b in (("AvB", rb), ("AvA", None)):
        st = D.run(ra, b, out, nthreads=4, paf_path=paf, paf_flags=2)
        # the self total is halved per thread by the reference (FastGA.c:1906): up to one short per thread of its -T4 run
        assert 0 <= st["nseeds"] - stats[tag]["total_seeds"] <= (4 if b is None else 0)
        assert st["nlive"] == stats[tag]["records"]
        assert [ln for ln in open(out).read().splitlines() if ln[:1] not in "!<"] == golden_lines(f"toy_{tag}.1aln.txt")
        assert open(paf).read() == open(os.path.join(GOLD, f"toy_{tag}.x.paf")).read()
    st = D.run(ra, rb, None, nthreads=4, paf_path=paf, paf_flags=8)
    assert open(paf).read() == open(os.path.join(GOLD, "toy_AvB.S.paf")).read()
    st = D.run(ra, rb, None, nthreads=4, paf_path=paf, paf_flags=32)
    assert open(paf).read() == open(os.path.join(GOLD, "toy_AvB.psl")).read()
    st = D.run(ra, rb, None, nthreads=4, paf_path=paf, paf_flags=0)
    assert open(paf).read() == open(os.path.join(GOLD, "toy_AvB.paf")).read()
