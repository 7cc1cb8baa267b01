#!/usr/bin/env python3
"""tools/multi_3g_check.py -- fga_run_multi at human scale: BASELINE configs[3] (3 Gbp x 3 Gbp, 1 %) through the C-ABI's
multi-GPU entry with `--devices` (default 0,0: two virtual ranks on one GPU, each holding half of both tables), the digest of
its .1aln against the golden the real reference's file gave (tests/golden/config4_3000m_digest.json, lines_md5 included)."""
import argparse, json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=3000.0)
ap.add_argument("--div", type=float, default=0.01)
ap.add_argument("--devices", default="0,0")
ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 8))
a = ap.parse_args()
from fastga_amd import workload, device as D
from oracle import harness as H
d = tempfile.mkdtemp(prefix="fga_m3g_")
try:
    t = time.time()
    ra, rb = workload.build_config4(d, mbp=a.mbp, divergence=a.div, threads=a.threads)
    print(f"genomes + GDBs: {time.time()-t:.1f} s", flush=True)
    devs = tuple(int(x) for x in a.devices.split(","))
    out = os.path.join(d, "multi.1aln")
    for rep in range(2):
        t = time.time()
        st = D.run_multi(ra, rb, out, devices=devs, nthreads=a.threads, reference_threads=32)
        dt = time.time() - t
        print(f"fga_run_multi over devices {list(devs)}: {dt:.2f} s (cold: sessions opened and closed inside) | seeds {st['nseeds']} "
              f"hits {st['nhits']} alns {st['nalns']} records {st['nlive']} | slowest rank: merge {st['merge_s']:.2f} sort {st['sort_s']:.2f} "
              f"chain {st['chain_s']:.2f} extend {st['extend_s']:.2f} filter {st['filter_s']:.2f} write {st['write_s']:.2f} s, "
              f"open {st['upload_s']:.2f} s, peak HBM {st['hbm_peak_bytes']/2**30:.1f} GiB", flush=True)
    if abs(a.mbp - 3000.0) < 1e-9 and os.path.exists(H.ref_bin("ONEview")):
        name = "config4" if a.div < 0.05 else "config5"
        g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"{name}_3000m_digest.json")))
        got = workload.digest_1aln_stream(out, H.ref_bin("ONEview"))
        print("digest == golden (records, header, multiset, order, lines):",
              all(got[k] == g[k] for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5")), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
