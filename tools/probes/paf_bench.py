#!/usr/bin/env python3
"""tools/paf_bench.py -- time native PAF output (-pafx) after the hot path on a synthetic pair, with the reference's
ALNtoPAF -x on the same .1aln beside it when oracle/_ref is present"""
import argparse, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import workload, device as D

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=100.0)
ap.add_argument("--div", type=float, default=0.02)
ap.add_argument("--threads", type=int, default=32)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_pb_")
ra, rb = workload.build_pair(d, seed=1, ncontig=40, total=int(a.mbp * 1e6), divergence=a.div,
                             repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02, threads=16, gix=False)
ses = D.Session(ra, rb, device=0)
out, paf = os.path.join(d, "x.1aln"), os.path.join(d, "x.paf")
for flags, name in ((0, "plain"), (2, "-x"), (2 | 8, "-xS")):
    for rep in range(2):
        st = ses.run(out_path=out, nthreads=a.threads, paf_path=paf, paf_flags=flags)
    print(f"native {name:5s}: hot path {1000*st['phase23_s']:.1f} ms | edit scripts {1000*st['trace_s']:.1f} ms "
          f"(kernels {st['trace_kernel_ms']:.2f} ms) | regroup+format {1000*st['paf_s']:.1f} ms | "
          f"{st['nlive']} alignments, {os.path.getsize(paf)/1e6:.1f} MB", flush=True)
ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ALNtoPAF")
if os.path.exists(ref):
    for fl in ([], ["-x"]):
        t = time.time()
        r = subprocess.run([ref, f"-T{a.threads}"] + fl + [out], cwd=d, capture_output=True)
        w = time.time() - t
        print(f"reference ALNtoPAF {' '.join(fl) or 'plain'} -T{a.threads}: {1000*w:.0f} ms wall, rc {r.returncode}, "
              f"{len(r.stdout)/1e6:.1f} MB", flush=True)
    st = ses.run(out_path=out, nthreads=a.threads, paf_path=paf, paf_flags=2)
    print("identical to reference -x:", open(paf, "rb").read() == r.stdout, flush=True)
