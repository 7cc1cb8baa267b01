#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the golden vectors of this directory with the REAL reference
(oracle/_ref, built from /root/reference by `make -C oracle ref`).  Run it where the reference build exists; the
outputs are data only (inputs + what the reference's own tools print for them) and are committed:

  toy_A.fa.gz, toy_B.fa.gz   two synthetic genomes (3 contigs each, ~60 kbp, 4 % divergence, one inverted block,
                             a planted repeat family), generator = fastga_amd.synth with the seed below
  toy_AvB.1aln.txt           ONEview text of `FastGA -T4 -1:x A B` (reference FAtoGDB + GIXmake + FastGA), without the
                             provenance ('!') and path ('<') lines
  toy_AvB.paf / .x.paf / .S.paf / .psl     ALNtoPAF (plain, -x, -S) and ALNtoPSL of that .1aln
  toy_AvA.1aln.txt, toy_AvA.x.paf          the same for the self comparison `FastGA -1:y A`
  toy_stats.json             seed totals and alignment counts printed by `FastGA -v`
"""
import gzip, json, os, re, shutil, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np                                              # noqa: E402
from fastga_amd import synth                                    # noqa: E402
from oracle import harness as H                                 # noqa: E402

SEED = 20260926


def make_genomes():
    rng = np.random.default_rng(SEED)
    lens = [28000, 19000, 13000]
    A = [rng.integers(0, 4, n, dtype=np.uint8) for n in lens]
    fam = rng.integers(0, 4, 700, dtype=np.uint8)               # a repeat family, copies 3-8 % diverged, both strands
    for c, pos, rc in ((0, 3000, False), (0, 17000, True), (1, 5000, False), (2, 8000, True)):
        cp = synth.mutate(rng, fam, float(rng.uniform(0.03, 0.08)))
        if rc:
            cp = synth.revcomp(cp)
        A[c][pos:pos + len(cp)] = cp[:len(A[c]) - pos]
    B = [synth.mutate(rng, a, 0.04) for a in A]
    blk = B[0][9000:13000].copy()
    B[0][9000:13000] = synth.revcomp(blk)                       # an inversion
    B = [B[1], B[0], B[2]]                                      # contig order differs between the genomes
    return A, B


def main():
    if not H.have_reference():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    A, B = make_genomes()
    w = tempfile.mkdtemp(prefix="fga_golden_")
    for name, g in (("A", A), ("B", B)):
        fa = os.path.join(w, f"{name}.fa")
        synth.write_fasta(fa, g, prefix=f"toy{name}_")
        with open(fa, "rb") as f, gzip.GzipFile(os.path.join(HERE, f"toy_{name}.fa.gz"), "wb", mtime=0) as z:
            z.write(f.read())
        H.run([H.ref_bin("FAtoGDB"), fa], cwd=w)
        H.run([H.ref_bin("GIXmake"), "-T4", f"-P{w}", os.path.join(w, name)], cwd=w)
    stats = {}
    for tag, b in (("AvB", "B"), ("AvA", None)):
        r, _ = H.ref_fastga(os.path.join(w, "A"), os.path.join(w, b) if b else None, w, os.path.join(w, tag), threads=4)
        m = re.search(r"Total seeds = ([\d,]+)", r.stderr)
        aln = os.path.join(w, tag + ".1aln")
        txt = H.oneview(aln)
        open(os.path.join(HERE, f"toy_{tag}.1aln.txt"), "w").write("\n".join(txt) + "\n")
        stats[tag] = {"total_seeds": int(m.group(1).replace(",", "")) if m else None,
                      "records": sum(1 for ln in txt if ln.startswith("A "))}
        opts = (("", "paf"), ("-x", "x.paf"), ("-S", "S.paf")) if b else (("-x", "x.paf"),)
        for fl, ext in opts:
            out = H.run([H.ref_bin("ALNtoPAF"), "-T2"] + ([fl] if fl else []) + [aln], cwd=w).stdout
            open(os.path.join(HERE, f"toy_{tag}.{ext}"), "w").write(out)
        if b:
            out = H.run([H.ref_bin("ALNtoPSL"), "-T2", aln], cwd=w).stdout
            open(os.path.join(HERE, f"toy_{tag}.psl"), "w").write(out)
    json.dump(stats, open(os.path.join(HERE, "toy_stats.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(w, ignore_errors=True)
    print(stats)


if __name__ == "__main__":
    main()
