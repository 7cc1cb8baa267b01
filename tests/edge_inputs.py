"""Hand-made awkward inputs shared by the CPU producer-parity test and the GPU end-to-end test: scaffolds with N gaps
(several contigs per scaffold, leading/trailing N), equal-length contigs (ties in the length sort), contigs shorter
than the 40-mer (and than the 12-mer prefix), homopolymer / dinucleotide / tandem repeats (one k-mer panel far larger
than a merge tile, wide extension waves), lower-case runs and ragged line lengths."""
import numpy as np

_UP = np.frombuffer(b"ACGT", dtype=np.uint8)


def _txt(s):
    return _UP[s].tobytes().decode()


def make_edge_scaffolds(seed, divergence=0.0, base=None):
    """list of (header, sequence text with N runs).  With `base` (the return value of a previous call with
    divergence 0) the ACGT stretches are mutated copies of it, so that the two genomes align."""
    from fastga_amd import synth
    rng = np.random.default_rng(seed)
    if base is not None:
        out = []
        for name, parts in base:
            q = []
            for p in parts:
                q.append(p if isinstance(p, int) else synth.mutate(rng, p, divergence))
            out.append((name, q))
        return out
    R = lambda n: rng.integers(0, 4, n, dtype=np.uint8)          # noqa: E731
    unit37 = R(37)
    scaf = [
        ("plain one", [R(30011)]),
        ("gapped leading-and-trailing-N", [25, R(12007), 300, R(12007), 1, R(9001), 40]),      # two equal-length contigs
        ("short pieces", [R(8000), 10, R(39), 10, R(40), 10, R(11), 10, R(41), 10, R(7000)]),
        ("low complexity", [R(5000), np.zeros(3000, np.uint8), R(5000), np.tile(np.array([0, 1], np.uint8), 1500),
                            R(5000), np.tile(unit37, 120), R(5000)]),
        ("twin a", [R(15013)]),
        ("twin b", [R(15013)]),                                                                    # equal lengths again
    ]
    # concatenate adjacent arrays of the low-complexity scaffold into one contig
    fixed = []
    for name, parts in scaf:
        q = []
        for p in parts:
            if q and not isinstance(p, int) and not isinstance(q[-1], int):
                q[-1] = np.concatenate([q[-1], p])
            else:
                q.append(p)
        fixed.append((name, q))
    return fixed


def write_edge_fasta(path, scaffolds, seed=0):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for name, parts in scaffolds:
            f.write(">" + name + "\n")
            txt = "".join("N" * p if isinstance(p, int) else _txt(p) for p in parts)
            # lower-case a stretch and use ragged line lengths
            if len(txt) > 4000:
                txt = txt[:1000] + txt[1000:1800].lower() + txt[1800:]
            i = 0
            while i < len(txt):
                w = int(rng.integers(50, 90))
                f.write(txt[i:i + w] + "\n")
                i += w
