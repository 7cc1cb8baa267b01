"""Build synthetic inputs end to end with the product's own producers (FASTA -> GDB -> GIX)."""
import os

import numpy as np

from . import synth
from .gixio import Gdb, Gix, fasta_to_gdb, build_gix  # noqa: F401


def build_genome(workdir, name, contigs, masks=None, threads=8, use_mask=False, gix=True):
    """write <name>.fa, <name>.gdb, .<name>.bps and (gix=True) <name>.gix, .<name>.ktab.* with the host producer;
    returns the root path.  With gix=False a Session builds the index on the device instead."""
    fa = os.path.join(workdir, name + ".fa")
    root = os.path.join(workdir, name)
    synth.write_fasta(fa, contigs, prefix=name.lower(), masks=masks)
    fasta_to_gdb(fa, root)
    if gix:
        g = Gdb(root + ".gdb")
        build_gix(g, root, threads, use_mask=use_mask)
        g.close()
    return root


def build_pair(workdir, seed, ncontig, total, divergence, repeat_frac=0.0, inv_frac=0.0, swap_frac=0.0,
               threads=8, names=("A", "B"), gix=True):
    lens = synth.contig_lengths(seed, ncontig, total)
    A, mA, B, mB = synth.make_pair(seed, lens, divergence, repeat_frac, inv_frac, swap_frac)
    ra = build_genome(workdir, names[0], A, None, threads, gix=gix)
    rb = build_genome(workdir, names[1], B, None, threads, gix=gix)
    return ra, rb


# ---- BASELINE.json configs as concrete inputs -------------------------------------------------------

def build_config1_s1(workdir, mbp=86.0, seed=20260926, threads=8, gix=False):
    """configs[0]'s substitute S1 (the EXAMPLE blobs are not in the image): 86 Mbp vs 86 Mbp, 40 contigs with log-uniform
    lengths (0.2 .. 12 Mbp before scaling), 4.5 % divergence, 15 % of the bases in repeat copies, 5 % of 40-kbp blocks
    inverted and 5 % swapped"""
    lens = synth.contig_lengths_loguniform(seed, 40, int(mbp * 1e6))
    A, mA, B, mB = synth.make_pair(seed, lens, 0.045, 0.15, 0.05, 0.05)
    ra = build_genome(workdir, "A", A, None, threads, gix=gix)
    rb = build_genome(workdir, "B", B, None, threads, gix=gix)
    return ra, rb


def build_config2(workdir, mbp=100.0, seed=1, divergence=0.02, ncontig=40, threads=8, gix=False):
    """configs[1]: synthetic pair, 2 % divergence, 40 contigs, 5 % repeats, 2 % of 40-kbp blocks inverted / swapped"""
    return build_pair(workdir, seed=seed, ncontig=ncontig, total=int(mbp * 1e6), divergence=divergence,
                      repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02, threads=threads, gix=gix)


def build_config3(workdir, mbp=1000.0, seed=2, ncontig=40, repeat_frac=0.30, tandem_frac=0.02, mask_frac=0.85,
                  threads=8, gix=False, name="S"):
    """configs[2]: one repeat-heavy genome for a self comparison with -M: 30 % of the bases in copies of four repeat
    families (1-15 % diverged, both strands) plus tandem arrays; mask_frac of the copies are lower case (soft mask), the
    rest are left for the comparison to find.  Returns the root; the GDB carries the mask intervals."""
    rng_lens = synth.contig_lengths(seed, ncontig, int(mbp * 1e6))
    rng = np.random.default_rng(seed)
    A = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in rng_lens]
    masks = synth.plant_repeats(rng, A, repeat_frac, mask_frac=mask_frac)
    synth.plant_tandem_arrays(rng, A, masks, tandem_frac, mask_frac=mask_frac)
    return build_genome(workdir, name, A, masks=masks, threads=threads, use_mask=True, gix=gix)


def build_config4(workdir, mbp=3000.0, divergence=0.01, seed=3, ncontig=32, repeat_frac=0.45, nfam=None, inv_frac=0.02,
                  swap_frac=0.02, threads=8, gix=False, names=("A", "B"), reuse_a=None):
    """configs[3] (divergence 0.01) and configs[4] (0.10): a human-scale pair, 32 contigs of ~94 Mbp at 3 Gbp, 45 % of the
    bases in diverged copies of `nfam` repeat families (default: one family per ~234 kbp, i.e. ~40 copies of each whatever
    the genome size -- with a fixed number of families the copy number, and with it the repeat-induced hits, would grow
    quadratically), 2 % of 40-kbp blocks inverted / swapped, made by the C generator (synth.write_pair_fast: seconds instead of minutes).  Genome A depends on (seed, mbp, ncontig,
    repeat_frac, nfam) only, so both configurations can share it: pass the root of an A built before as `reuse_a`.
    Returns (rootA, rootB)."""
    lens = synth.contig_lengths(seed, ncontig, int(mbp * 1e6))
    if nfam is None:
        nfam = max(4, int(round(mbp * 256 / 60)))
    fa, fb = os.path.join(workdir, names[0] + ".fa"), os.path.join(workdir, names[1] + ".fa")
    synth.write_pair_fast(seed, lens, divergence, fa, fb, repeat_frac=repeat_frac, nfam=nfam, inv_frac=inv_frac,
                          swap_frac=swap_frac, bseed=int(round(divergence * 1000)), prefix_a=names[0].lower(),
                          prefix_b=names[1].lower(), threads=threads)
    roots = []
    for nm, f in zip(names, (fa, fb)):
        root = os.path.join(workdir, nm)
        if nm == names[0] and reuse_a is not None:
            os.unlink(f)
            roots.append(reuse_a)
            continue
        fasta_to_gdb(f, root)
        os.unlink(f)                                     # 3 GB of text each: the GDB is what everything reads
        if gix:
            g = Gdb(root + ".gdb")
            build_gix(g, root, threads)
            g.close()
        roots.append(root)
    return tuple(roots)



def build_config6g(workdir, mbp=6000.0, ncontig=64, nsmall=4, repeats=0.01, threads=8):
    """A genome whose index holds more than 2^32 entries (6 Gbp in 64 contigs of ~94 Mbp, 1 % repeats: 4.76 G entries) and a
    small one that is homologous to its beginning: the first `nsmall` contigs of its 1 %-diverged copy (C generator, seed 6).
    Returns (root of the big genome G, root of the small genome S)."""
    lens = synth.contig_lengths(6, ncontig, int(mbp * 1e6))
    fa, fb = os.path.join(workdir, "G.fa"), os.path.join(workdir, "B.fa")
    synth.write_pair_fast(6, lens, 0.01, fa, fb, repeat_frac=repeats, nfam=max(4, int(round(mbp * 256 / 60))), inv_frac=0.02,
                          swap_frac=0.02, bseed=10, prefix_a="g", prefix_b="s", threads=threads)
    fs = os.path.join(workdir, "S.fa")
    with open(fb) as src, open(fs, "w") as dst:            # the first nsmall records of the copy
        n = 0
        for ln in src:
            if ln.startswith(">"):
                n += 1
                if n > nsmall:
                    break
            dst.write(ln)
    os.unlink(fb)
    roots = []
    for nm, f in (("G", fa), ("S", fs)):
        root = os.path.join(workdir, nm)
        fasta_to_gdb(f, root)
        os.unlink(f)
        roots.append(root)
    return tuple(roots)


def digest_1aln(lines):
    """A digest of a .1aln as ONEview prints it that does not depend on how ties on (aread, abpos) are ordered (the
    reference orders them by the thread slot that held the record, FastGA.c:3906-3918): record count, md5 of the
    header lines, md5 of the records as a sorted multiset, md5 of the (aread, abpos) sequence."""
    import hashlib
    lines = [ln for ln in lines if ln[:1] not in ("!", "<")]
    first = next((i for i, ln in enumerate(lines) if ln.startswith("A ")), len(lines))
    recs, cur = [], []
    for ln in lines[first:]:
        if ln.startswith("A ") and cur:
            recs.append("\n".join(cur))
            cur = []
        cur.append(ln)
    if cur:
        recs.append("\n".join(cur))
    order = " ".join(" ".join(r.split("\n", 1)[0].split()[1:3]) for r in recs)
    md5 = lambda t: hashlib.md5(t.encode()).hexdigest()      # noqa: E731
    return {"records": len(recs), "header_md5": md5("\n".join(lines[:first])),
            "records_md5": md5("\n".join(sorted(recs))), "order_md5": md5(order),
            "lines_md5": md5("".join(ln + "\n" for ln in lines[first:]))}      # the record lines in sequence: ties included


def digest_1aln_stream(path, oneview_bin):
    """digest_1aln for files of millions of records: ONEview's text is consumed line by line, nothing is held.  The
    records as a multiset are digested order-independently (sum of the records' md5 values mod 2^128) instead of by
    sorting them; header and (aread, abpos) order as in digest_1aln.  Keys are different from digest_1aln's on purpose
    (records_sum128 instead of records_md5)."""
    import hashlib
    import subprocess
    p = subprocess.Popen([oneview_bin, path], stdout=subprocess.PIPE, text=True, bufsize=1 << 20)
    head, order, seq = hashlib.md5(), hashlib.md5(), hashlib.md5()
    total, nrec, cur, in_head, first = 0, 0, [], True, True
    mask = (1 << 128) - 1

    def close_record():
        nonlocal total, nrec, cur
        if cur:
            total = (total + int.from_bytes(hashlib.md5("\n".join(cur).encode()).digest(), "big")) & mask
            nrec += 1
            cur = []

    for ln in p.stdout:
        ln = ln.rstrip("\n")
        if ln[:1] in ("!", "<"):
            continue
        if ln.startswith("A "):
            in_head = False
            close_record()
            f = ln.split()
            order.update((("" if first else " ") + f[1] + " " + f[2]).encode())
            first = False
        if in_head:
            head.update((ln + "\n").encode())
        else:
            cur.append(ln)
            seq.update((ln + "\n").encode())
    close_record()
    if p.wait() != 0:
        raise RuntimeError(f"{oneview_bin} {path} failed")
    return {"records": nrec, "header_md5": head.hexdigest(), "records_sum128": f"{total:032x}",
            "order_md5": order.hexdigest(), "lines_md5": seq.hexdigest()}      # lines_md5: the record lines in sequence


def digest_1aln_records(path):
    """A digest of a .1aln through this library's own reader (fga_read_1aln; no reference tool needed): record count, md5 of
    the records' nine fields in file order, md5 of all trace bytes in file order.  Two files with the same digest hold the
    same records in the same sequence; it is not comparable with digest_1aln's ONEview-text digests."""
    import ctypes as C
    import hashlib
    import numpy as np
    from .lib import load_library, Alns, check
    from .device import ALN_DTYPE
    L = load_library()
    out = C.POINTER(Alns)()
    check(L.fga_read_1aln(path.encode(), C.byref(out), None, None, None), "fga_read_1aln")
    o = out.contents
    a = np.frombuffer((C.c_char * (o.naln * ALN_DTYPE.itemsize)).from_address(o.alns), dtype=ALN_DTYPE) \
        if o.naln else np.zeros(0, ALN_DTYPE)
    h = hashlib.md5()
    for f in ("aread", "bread", "abpos", "bbpos", "aepos", "bepos", "flags", "diffs", "tlen"):
        h.update(np.ascontiguousarray(a[f]).tobytes())
    t = hashlib.md5()
    if o.ntrace:
        t.update((C.c_char * o.ntrace).from_address(o.tbytes))
    res = {"records": int(o.naln), "fields_md5": h.hexdigest(), "trace_md5": t.hexdigest()}
    L.fga_alns_free(out)
    return res
