#!/usr/bin/env python3
"""tools/gix_scan_probe.py -- the device index build alone on one genome of the human-scale generator (timing experiments)"""
import argparse, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastga_amd import workload, device as D
from fastga_amd.gixio import Gdb
ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=1000.0)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="fga_gsp_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    ra, rb = workload.build_config4(d, mbp=a.mbp, divergence=0.01, threads=16)
    g = Gdb(ra + ".gdb")
    dev = D.Device(0)
    for rep in range(2):
        t = time.time()
        dg, x = D.build_gix_device(dev, g, 8)
        print(f"rep {rep}: device GIX build {1000*(time.time()-t):.1f} ms wall, kernels {dev.stage_ms(5):.2f} ms, {x.nents} entries", flush=True)
        dg.free(); x.close()
finally:
    shutil.rmtree(d, ignore_errors=True)
