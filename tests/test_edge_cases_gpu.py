"""GPU end to end on awkward inputs, against the real reference (oracle/_ref) run on the same GDB/GIX files:
scaffolds with N gaps and a zero-length contig, contigs shorter than the k-mer, equal-length contigs, homopolymer /
dinucleotide / tandem repeats (a k-mer panel far larger than a merge tile: the oversize-tile path; wide waves: the LDS
ring of the extension), identical genomes (snakes of whole-contig length) and unrelated genomes (no alignment)."""
import os

import numpy as np
import pytest

from tests.edge_inputs import make_edge_scaffolds, write_edge_fasta
from tests.test_end_to_end_gpu import _compare

pytestmark = pytest.mark.gpu


def _build(d, name, scaffolds, seed):
    from fastga_amd.gixio import Gdb, fasta_to_gdb, build_gix
    fa = os.path.join(d, name + ".fa")
    write_edge_fasta(fa, scaffolds, seed=seed)
    root = os.path.join(d, name)
    fasta_to_gdb(fa, root)
    g = Gdb(root + ".gdb")
    build_gix(g, root, 8)
    g.close()
    return root


def test_gapped_short_and_low_complexity_contigs(tmp_path, built_library):
    d = str(tmp_path)
    base = make_edge_scaffolds(3)
    ra = _build(d, "A", base, 1)
    rb = _build(d, "B", make_edge_scaffolds(4, divergence=0.03, base=base), 2)
    st = _compare(ra, rb, d)
    assert st["nlive"] >= 8
    # the same pair with -S and a higher frequency cutoff drives more of the repeat seeds through
    _compare(ra, rb, d, symmetric=True, freq=30)


def test_identical_genomes(tmp_path, built_library):
    """0 % divergence: one snake per contig, many trace points crossed in a single wave step"""
    d = str(tmp_path)
    base = make_edge_scaffolds(5)
    ra = _build(d, "A", base, 1)
    rb = _build(d, "B", base, 2)
    st = _compare(ra, rb, d)
    assert st["nlive"] >= 5


def test_unrelated_genomes_give_an_empty_1aln(tmp_path, built_library):
    from fastga_amd import workload, synth
    d = str(tmp_path)
    la = synth.contig_lengths(21, 4, 120_000)
    A, _, _, _ = synth.make_pair(21, la, 0.0, self_only=True)
    B, _, _, _ = synth.make_pair(22, la + 3, 0.0, self_only=True)
    ra = workload.build_genome(d, "A", A)
    rb = workload.build_genome(d, "B", B)
    st = _compare(ra, rb, d, allow_empty=True)
    assert st["nlive"] == 0


def test_self_comparison_of_the_edge_genome(tmp_path, built_library):
    d = str(tmp_path)
    ra = _build(d, "A", make_edge_scaffolds(3), 1)
    _compare(ra, None, d, strict_order=False, allow_empty=True)


def test_seven_byte_payloads_many_small_contigs_and_a_long_one(tmp_path, built_library):
    """More than 32,768 contigs (3 contig bytes) with one beyond 16.7 Mbp (4 position bytes): a 7-byte payload, which
    GIXmake sizes freely (GIXmake.c:1888-1901).  It does not fit under the k-mer in the device builder's 128-bit key (the
    key then carries the k-mer's position in a concatenation of the contigs) and needs all seven payload bytes of the
    uploaded entries.  Index files from the host producer, then the same comparison with both indices built on the
    device: both must be the reference's .1aln (its own FastGA on the host producer's files)."""
    from fastga_amd import workload, synth, device as D
    from fastga_amd.gixio import Gix
    from oracle import harness as H
    d = str(tmp_path)
    rng = np.random.default_rng(707)
    big = rng.integers(0, 4, 17_200_000, dtype=np.uint8)
    small = [rng.integers(0, 4, int(n), dtype=np.uint8) for n in rng.integers(60, 140, 33_000)]
    A = [big] + small
    B = [synth.mutate(rng, big, 0.02)] + [synth.mutate(rng, c, 0.03) for c in small[:20_000]] + \
        [rng.integers(0, 4, int(n), dtype=np.uint8) for n in rng.integers(60, 140, 13_500)]
    ra = workload.build_genome(d, "A", A, threads=8)
    rb = workload.build_genome(d, "B", B, threads=8)
    xa = Gix(ra + ".gix")
    assert xa.postbytes == 4 and xa.contbytes == 3
    st = _compare(ra, rb, d)                                         # index files uploaded: every payload byte of the entries
    assert st["nlive"] > 100
    ref = H.oneview(os.path.join(d, "ref.1aln"))
    dev = os.path.join(d, "dev.1aln")
    st2 = D.run(ra, rb, dev, nthreads=8, reference_threads=8, build_index=True)      # both indices built on the device
    assert st2["nseeds"] == st["nseeds"] and st2["nlive"] == st["nlive"]
    assert H.oneview(dev) == H.oneview(os.path.join(d, "ours.1aln")) == ref
