"""CPU-side checks: the C-ABI library loads and exports every symbol include/fastga_amd.h declares; the
device entry points fail loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported(built_library):
    hdr = open(os.path.join(ROOT, "include", "fastga_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(fga_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(built_library, n)]
    assert not missing, missing


def test_device_open_fails_loudly_without_gpu(built_library):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from fastga_amd import device as D
    from fastga_amd.lib import FgaError
    with pytest.raises(FgaError, match="no CPU fallback"):
        D.Device(0)


def test_gdb_gix_roundtrip_through_c_abi(toy_pair):
    from fastga_amd.gixio import Gdb, Gix
    d, ra, rb = toy_pair
    g = Gdb(ra + ".gdb")
    x = Gix(ra + ".gix")
    assert g.ncontig == 12 and x.nctg == 12
    assert x.nents == x.index[-1] and x.ebytes == 9 + x.postbytes + x.contbytes
    # contig order in the index is length-descending
    lens = g.clen[x.perm]
    assert all(lens[i] >= lens[i + 1] for i in range(len(lens) - 1))
    # every entry's k-mer suffix really is the genome at its position (forward strand entries)
    e = x.entries()
    import numpy as np
    rng = np.random.default_rng(0)
    for j in rng.integers(0, x.nents, 200):
        ent = e[j]
        pay = int.from_bytes(bytes(ent[9:9 + x.pbyte]), "little")
        pos = pay & ((1 << (8 * x.postbytes)) - 1)
        ctg = pay >> (8 * x.postbytes)
        sign = ctg >> (8 * x.contbytes - 1)
        ctg &= (1 << (8 * x.contbytes - 1)) - 1
        seq = g.contig(int(x.perm[ctg]))
        suf = [(int(ent[b]) >> s) & 3 for b in range(7) for s in (6, 4, 2, 0)]
        if sign == 0:
            assert list(seq[pos + 12:pos + 40]) == suf
        else:
            kmer = [3 - int(v) for v in seq[pos - 40:pos][::-1]]
            assert kmer[12:] == suf
