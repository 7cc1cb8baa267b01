#!/usr/bin/env python3
"""tools/config3_check.py -- BASELINE configs[2]: the repeat-heavy self comparison with -M at a given size; stage times of
two passes over the resident inputs (index built on the device)."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import workload, device as D
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=1000.0)
ap.add_argument("--threads", type=int, default=32)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_c3_")
t = time.time()
root = workload.build_config3(d, mbp=a.mbp, threads=a.threads)
print(f"genome + GDB {time.time()-t:.1f}s", flush=True); t = time.time()
ses = D.Session(root, None)
print(f"upload + device index {time.time()-t:.2f}s  table {ses.table_bytes/1e9:.2f} GB", flush=True)
for rep in range(2):
    t = time.time()
    st = ses.run(out_path=os.path.join(d, "out.1aln"), nthreads=a.threads, soft_mask=True)
    dt = time.time() - t
    print(f"run {rep}: {dt*1000:.0f} ms = {a.mbp*1e-3/dt:.2f} Gbp/s  seeds {st['nseeds']} hits {st['nhits']} units {st['nunits']} "
          f"alns {st['nalns']} live {st['nlive']} waves {st['nwaves']} parts {st['nparts']}", flush=True)
    print("   stages ms:", {k: round(1000*st[k], 1) for k in ("merge_s","sort_s","chain_s","extend_s","filter_s","write_s")},
          "kernels ms:", {k: round(st[k], 2) for k in ("merge_kernel_ms","sort_kernel_ms","extend_kernel_ms")}, flush=True)
    print(f"   peak HBM in use {st['hbm_peak_bytes']/2**30:.1f} GiB", flush=True)
    alg = ses.table_bytes + 2 * st["nseeds"] * ses.seed_bytes
    print(f"   merge {alg/st['merge_kernel_ms']/1e6:.0f} GB/s algorithmic ({alg/st['merge_kernel_ms']/1e6/80:.1f}% of 8 TB/s)", flush=True)
ses.close()
