#!/usr/bin/env python3
"""tools/scale_check.py -- robustness at scale: build a synthetic genome (pair or self, repeat-heavy, soft-masked) with
our own producers and run the whole hot path once; prints stage times and counts."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastga_amd import workload, synth, device as D

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=300.0)
ap.add_argument("--contigs", type=int, default=40)
ap.add_argument("--div", type=float, default=0.02)
ap.add_argument("--repeats", type=float, default=0.30)
ap.add_argument("--self", dest="self_", action="store_true")
ap.add_argument("--mask", action="store_true")
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--host-gix", action="store_true", help="build .gix files on the host instead of on the device")
ap.add_argument("--pafx", action="store_true", help="also write PAF with CIGARs (edit scripts on the device)")
ap.add_argument("--seed", type=int, default=2)
ap.add_argument("--inv", type=float, default=0.02, help="fraction of 40-kbp blocks inverted / swapped")
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_scale_")
t = time.time()
lens = synth.contig_lengths(a.seed, a.contigs, int(a.mbp * 1e6))
A, mA, B, mB = synth.make_pair(a.seed, lens, a.div, repeat_frac=a.repeats, inv_frac=a.inv, swap_frac=a.inv, self_only=a.self_)
print(f"synth {time.time()-t:.1f}s", flush=True); t = time.time()
ra = workload.build_genome(d, "A", A, masks=mA if a.mask else None, threads=a.threads, use_mask=a.mask, gix=a.host_gix)
rb = None if a.self_ else workload.build_genome(d, "B", B, threads=a.threads, gix=a.host_gix)
print(f"GDB+GIX build {time.time()-t:.1f}s", flush=True); t = time.time()
ses = D.Session(ra, rb)
print(f"load+upload {time.time()-t:.1f}s  table bytes {ses.table_bytes/1e9:.2f} GB", flush=True)
for rep in range(2):
    t = time.time()
    st = ses.run(out_path=os.path.join(d, "out.1aln"), nthreads=a.threads, soft_mask=a.mask,
                 paf_path=os.path.join(d, "out.paf") if a.pafx else None, paf_flags=2 if a.pafx else 0)
    dt = time.time() - t
    print(f"run {rep}: {dt*1000:.0f} ms  seeds {st['nseeds']} hits {st['nhits']} units {st['nunits']} alns {st['nalns']} "
          f"live {st['nlive']} calls {st['ncalls']} waves {st['nwaves']}", flush=True)
    print("   stages ms:", {k: round(1000*st[k], 1) for k in ("merge_s","sort_s","download_s","chain_s","extend_s","filter_s","write_s")},
          "kernels ms:", {k: round(st[k], 2) for k in ("merge_kernel_ms","sort_kernel_ms","extend_kernel_ms")}, flush=True)
    if a.pafx:
        print(f"   PAF -x: edit scripts {1000*st['trace_s']:.1f} ms (kernels {st['trace_kernel_ms']:.1f} ms), regroup+format "
              f"{1000*st['paf_s']:.1f} ms, {os.path.getsize(os.path.join(d, 'out.paf'))/1e6:.0f} MB", flush=True)
    print(f"   peak HBM in use {st['hbm_peak_bytes']/2**30:.1f} GiB", flush=True)
    alg = ses.table_bytes + st["nseeds"] * (2 if a.self_ else 1) * ses.seed_bytes
    print(f"   merge kernel {alg/st['merge_kernel_ms']/1e6:.0f} GB/s algorithmic", flush=True)
ses.close()
