"""Synthetic genome pairs for parity tests and bench.py (SURVEY.md 8d recipe).

All sequence is i.i.d. ACGT except for planted repeat families; the second genome is the first one after
block rearrangements (inversions, swaps) and point divergence (60 % substitutions, 20 % 1-bp deletions,
20 % 1-bp insertions).  Everything is numpy-vectorised so that 100 Mbp takes seconds.
"""
import numpy as np

_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)
_LUT_UP = np.frombuffer(b"ACGT", dtype=np.uint8)
_LUT_LO = np.frombuffer(b"acgt", dtype=np.uint8)


def revcomp(s):
    return _COMP[s[::-1]]


def mutate(rng, s, rate):
    """point divergence: substitutions (0.6), deletions (0.2), insertions (0.2) at total `rate`."""
    n = len(s)
    if rate <= 0 or n == 0:
        return s.copy()
    r = rng.random(n)
    sub = r < 0.6 * rate
    dele = (r >= 0.6 * rate) & (r < 0.8 * rate)
    ins = (r >= 0.8 * rate) & (r < rate)
    out = s.copy()
    k = int(sub.sum())
    if k:
        out[sub] = (out[sub] + rng.integers(1, 4, k, dtype=np.uint8)) & 3
    reps = np.ones(n, dtype=np.int64)
    reps[dele] = 0
    reps[ins] = 2
    res = np.repeat(out, reps)
    # the second copy of an inserted position becomes a random base
    ends = np.cumsum(reps)
    ipos = ends[ins] - 1
    if len(ipos):
        res[ipos] = rng.integers(0, 4, len(ipos), dtype=np.uint8)
    return res


def plant_repeats(rng, contigs, frac, fam_lens=(300, 1000, 3000, 6000), div_lo=0.01, div_hi=0.15, mask_frac=1.0):
    """Overwrite ~frac of the bases with diverged copies (both strands) of a few repeat families.
    Returns a parallel list of boolean masks marking repeat-copy bases (every copy with mask_frac = 1, else that
    fraction of the copies -- an annotation that misses some)."""
    masks = [np.zeros(len(c), dtype=bool) for c in contigs]
    if frac <= 0:
        return masks
    total = sum(len(c) for c in contigs)
    fams = [rng.integers(0, 4, L, dtype=np.uint8) for L in fam_lens]
    target = int(frac * total)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    placed = 0
    while placed < target:
        f = fams[int(rng.integers(0, len(fams)))]
        copy = mutate(rng, f, float(rng.uniform(div_lo, div_hi)))
        if rng.random() < 0.5:
            copy = revcomp(copy)
        c = int(rng.choice(len(contigs), p=lens / lens.sum()))
        if len(contigs[c]) <= len(copy) + 1:
            continue
        pos = int(rng.integers(0, len(contigs[c]) - len(copy)))
        contigs[c][pos:pos + len(copy)] = copy
        if mask_frac >= 1.0 or rng.random() < mask_frac:
            masks[c][pos:pos + len(copy)] = True
        placed += len(copy)
    return masks


def plant_tandem_arrays(rng, contigs, masks, frac, unit_lo=20, unit_hi=400, copies_lo=10, copies_hi=200,
                        div=0.03, mask_frac=1.0):
    """Overwrite ~frac of the bases with tandem arrays (a random unit repeated head to tail, every copy diverged by
    `div`), masked like plant_repeats' copies."""
    total = sum(len(c) for c in contigs)
    target = int(frac * total)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    placed = 0
    while placed < target:
        unit = rng.integers(0, 4, int(rng.integers(unit_lo, unit_hi)), dtype=np.uint8)
        n = int(rng.integers(copies_lo, copies_hi))
        arr = mutate(rng, np.tile(unit, n), div)
        c = int(rng.choice(len(contigs), p=lens / lens.sum()))
        if len(contigs[c]) <= len(arr) + 1:
            continue
        pos = int(rng.integers(0, len(contigs[c]) - len(arr)))
        contigs[c][pos:pos + len(arr)] = arr
        if mask_frac >= 1.0 or rng.random() < mask_frac:
            masks[c][pos:pos + len(arr)] = True
        placed += len(arr)


def rearrange(rng, s, inv_frac, swap_frac, mean_block=40000):
    """cut into blocks (exponential lengths, mean `mean_block`); invert / swap a fraction of them."""
    n = len(s)
    if n < 4 * 1000 or (inv_frac <= 0 and swap_frac <= 0):
        return s
    cuts = [0]
    while cuts[-1] < n:
        cuts.append(cuts[-1] + max(1000, int(rng.exponential(mean_block))))
    cuts[-1] = n
    blocks = [s[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
    nb = len(blocks)
    for i in range(nb):
        if rng.random() < inv_frac:
            blocks[i] = revcomp(blocks[i])
    nsw = int(swap_frac * nb / 2)
    for _ in range(nsw):
        i, j = rng.integers(0, nb, 2)
        blocks[i], blocks[j] = blocks[j], blocks[i]
    return np.concatenate(blocks)


def make_pair(seed, contig_lens, divergence, repeat_frac=0.0, inv_frac=0.0, swap_frac=0.0,
              self_only=False):
    """Returns (contigsA, masksA, contigsB, masksB); B is None when self_only."""
    rng = np.random.default_rng(seed)
    A = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in contig_lens]
    mA = plant_repeats(rng, A, repeat_frac)
    if self_only:
        return A, mA, None, None
    B, mB = [], []
    for c, m in zip(A, mA):
        b = rearrange(rng, c, inv_frac, swap_frac)
        b = mutate(rng, b, divergence)
        B.append(b)
        mB.append(np.zeros(len(b), dtype=bool))
    return A, mA, B, mB


def write_fasta(path, contigs, prefix="ctg", masks=None, width=80):
    """FASTA writer; bases covered by `masks` are written in lower case (soft mask)."""
    with open(path, "wb") as f:
        for i, s in enumerate(contigs):
            f.write(f">{prefix}{i}\n".encode())
            txt = _LUT_UP[s]
            if masks is not None and masks[i] is not None and masks[i].any():
                txt = np.where(masks[i], _LUT_LO[s], txt)
            n = len(txt)
            full = (n // width) * width
            if full:
                body = np.empty((n // width, width + 1), dtype=np.uint8)
                body[:, :width] = txt[:full].reshape(-1, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if n > full:
                f.write(txt[full:].tobytes() + b"\n")


def contig_lengths_loguniform(seed, ncontig, total, lo=0.2e6, hi=12e6):
    """ncontig distinct lengths, log-uniform between lo and hi, scaled to sum to ~total."""
    rng = np.random.default_rng(seed + 104729)
    w = np.exp(rng.uniform(np.log(lo), np.log(hi), ncontig))
    lens = np.maximum(1000, (w / w.sum() * total).astype(np.int64))
    return lens + np.arange(ncontig)     # break ties


def contig_lengths(seed, ncontig, total, spread=0.3):
    """ncontig distinct lengths summing to ~total (distinct so the length sort has no ties)."""
    rng = np.random.default_rng(seed + 7919)
    w = 1.0 + spread * (rng.random(ncontig) - 0.5)
    lens = np.maximum(1000, (w / w.sum() * total).astype(np.int64))
    lens = lens + np.arange(ncontig)     # break ties
    return lens


# ---- the fast generator for human-scale pairs (tools/fga_synth.c -> fastga_amd/libfga_synth.so) ----------------------

def write_pair_fast(seed, contig_lens, divergence, fasta_a, fasta_b=None, repeat_frac=0.0, nfam=4, inv_frac=0.0,
                    swap_frac=0.0, bseed=0, prefix_a="a", prefix_b="b", threads=8):
    """FASTA files of a synthetic pair straight from C threads (same recipe as make_pair, its own random streams: one
    per contig, so the files do not depend on `threads`).  Genome A depends on (seed, lens, repeat_frac, nfam) only;
    B also on (bseed, divergence, inv_frac, swap_frac).  Returns B's contig lengths (None without fasta_b)."""
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfga_synth.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make -C fastga_amd/csrc`")
    L = C.CDLL(path)
    L.fga_synth_pair.restype = C.c_int
    L.fga_synth_pair.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                 C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p,
                                 C.c_int]
    lens = np.ascontiguousarray(contig_lens, dtype=np.int64)
    blen = np.zeros(len(lens), dtype=np.int64)
    rc = L.fga_synth_pair(seed, bseed, len(lens), lens.ctypes.data, divergence, repeat_frac, nfam, inv_frac, swap_frac,
                          fasta_a.encode(), prefix_a.encode(), fasta_b.encode() if fasta_b else None,
                          prefix_b.encode(), blen.ctypes.data, threads)
    if rc != 0:
        raise RuntimeError("fga_synth_pair failed (memory or file error)")
    return blen if fasta_b else None
