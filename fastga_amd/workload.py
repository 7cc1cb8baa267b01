"""Build synthetic inputs end to end with the product's own producers (FASTA -> GDB -> GIX)."""
import os

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
