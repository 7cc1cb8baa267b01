#!/usr/bin/env python3
"""tests/golden/make_golden_config4.py -- golden digests for BASELINE.json configs[3] / [4] (3 Gbp x 3 Gbp, 1 % and 10 %
divergence), produced with the REAL reference: oracle/_ref/GIXmake builds both indices, oracle/_ref/FastGA runs the
comparison, fastga_amd.workload.digest_1aln_stream digests what ONEview prints for its .1aln.  The genomes are not
committed (2 x 3 Gbp); they are regenerated bit for bit by fastga_amd.workload.build_config4 (tools/fga_synth.c) from
their seed.  The work is done by tools/config4_check.py (which, where a GPU is present, also runs the product on the same
genomes first -- with an index built on the device, so the two programs do not share an index builder -- and compares).

A 3 Gbp pair needs ~70 GB for the two index file sets and ~25 GB of seed files: run it where /dev/shm or a disk has that
(the committed digests were made on the GPU box: 256 cores, 3 TB RAM, work directory in /dev/shm, reference -T32:
74 s and 123 s wall; ours 6.9 s and 10.1 s on one MI355X, identical digests):

  python tests/golden/make_golden_config4.py [--mbp 3000] [--threads 32] [--workdir /dev/shm/fga_golden]
"""
import argparse, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=3000.0)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--workdir", default="/dev/shm/fga_golden")
ap.add_argument("--no-gpu", action="store_true")
ap.add_argument("--outdir", default=HERE, help="where the digests go (default: beside this script)")
a = ap.parse_args()
for name, div in (("config4", 0.01), ("config5", 0.10)):
    wd = os.path.join(a.workdir, name)
    os.makedirs(wd, exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "config4_check.py"), "--mbp", f"{a.mbp:g}", "--div", f"{div:g}",
           "--reference", "--ref-threads", str(a.threads), "--workdir", wd,
           "--golden", os.path.join(a.outdir, f"{name}_{a.mbp:g}m_digest.json")] + (["--no-gpu"] if a.no_gpu else [])
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    subprocess.run(["rm", "-rf", wd])
