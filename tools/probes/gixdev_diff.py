#!/usr/bin/env python3
"""tools/gixdev_diff.py -- where does the device-built index differ from the host-built one (debug aid)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fastga_amd import workload, device as D
from fastga_amd.gixio import Gix, Gdb
d = tempfile.mkdtemp(prefix="fga_gd_")
ra, rb = workload.build_pair(d, seed=11, ncontig=12, total=600_000, divergence=0.03, repeat_frac=0.05, inv_frac=0.05, swap_frac=0.05)
g = Gdb(ra + ".gdb"); host = Gix(ra + ".gix")
dev = D.Device(0)
dg, x = D.build_gix_device(dev, g, 8, host_copy=True)
a, b = x.entries(), host.entries()
dif = np.nonzero((a != b).any(axis=1))[0]
print("rows", len(a), "differing", len(dif), "cols differing:", np.nonzero((a != b).any(axis=0))[0])
for r in dif[:6]:
    print(r, a[r].tolist(), b[r].tolist(), "prev", b[r-1].tolist())
print("stage ms", dev.stage_ms(5))
