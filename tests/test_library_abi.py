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


def test_product_sources_keep_the_stream_rules():
    """Source rules the library's concurrency rests on, checked on the text of fastga_amd/csrc: every device
    context launches on its own NON-BLOCKING stream, so nothing may go to the legacy default stream and expect to be
    ordered with it -- no hipMemset (asynchronous, default stream: the fill of a buffer could land after the kernel
    launched behind it had written it, which cost fga_run_multi alignments in 3 % of its runs), no kernel launch or
    asynchronous copy / fill on stream 0; and the product never reaches for the oracle."""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastga_amd", "csrc")
    files = sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.inc"))
                   + glob.glob(os.path.join(root, "*.c")) + glob.glob(os.path.join(root, "*.h*")))
    assert len(files) > 20
    bad = []
    for f in files:
        text = open(f, errors="replace").read()
        code = re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))      # comments may name what is banned
        for m in re.finditer(r"\bhipMemset\s*\(", code):
            bad.append((os.path.basename(f), "hipMemset("))
        for m in re.finditer(r"\bhipMemsetAsync\s*\(([^;]*?)\)\s*[;)!=|&]", code, flags=re.S):
            last = m.group(1).split(",")[-1].strip()
            if last in ("0", "NULL", "nullptr", "(hipStream_t) 0"):
                bad.append((os.path.basename(f), "hipMemsetAsync on stream 0"))
        for m in re.finditer(r"hipLaunchKernelGGL\s*\(([^;]*?)\)\s*;", code, flags=re.S):
            args = m.group(1).split(",")
            if len(args) >= 5 and args[4].strip() in ("0", "NULL", "nullptr"):
                bad.append((os.path.basename(f), "kernel launch on stream 0"))
        if re.search(r"oracle/|liboracle|libalign_ref", code):
            bad.append((os.path.basename(f), "mentions the oracle"))
    assert not bad, bad
    dev = open(os.path.join(root, "fga_device.hip")).read()
    assert "hipStreamCreateWithFlags(&d->stream,hipStreamNonBlocking)" in dev
