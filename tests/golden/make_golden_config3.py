#!/usr/bin/env python3
"""tests/golden/make_golden_config3.py -- golden digest for BASELINE.json configs[2] (1 Gbp repeat-heavy self comparison,
soft mask on), produced with the REAL reference (oracle/_ref/FastGA -M).  The genome is not committed (1 Gbp); it is
regenerated bit for bit from fastga_amd.workload.build_config3 and its seed, index files from the product's host
producer (whose mask bytes tests/test_edge_cases.py pins against `GIXmake -T1 ... #`; the reference's own masked GIXmake
races for -T > 1).  Output: config3_<mbp>m_digest.json = fastga_amd.workload.digest_1aln of what ONEview prints for the
reference's .1aln + the totals of `FastGA -v`.

  python tests/golden/make_golden_config3.py [--mbp 1000] [--threads 8] [--workdir DIR]
"""
import argparse, json, os, re, sys, tempfile, time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from fastga_amd import workload                                 # noqa: E402
from oracle import harness as H                                 # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=1000.0)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--workdir", default=None)
ap.add_argument("--device-index", action="store_true",
                help="index files from a device build (byte-identical to the host producer's, tests/test_gix_device_gpu.py) "
                     "instead of the host producer (minutes at 1 Gbp)")
ap.add_argument("--out", default=None, help="where the digest goes (default: beside this script)")
a = ap.parse_args()
d = a.workdir or tempfile.mkdtemp(prefix="fga_golden_c3_")
os.makedirs(d, exist_ok=True)
t = time.time()
root = workload.build_config3(d, mbp=a.mbp, threads=a.threads, gix=not a.device_index)
if a.device_index:
    from fastga_amd import device as D
    from fastga_amd.gixio import Gdb
    dev = D.Device(0)
    g = Gdb(root + ".gdb")
    dgx, xg = D.build_gix_device(dev, g, a.threads, host_copy=True, use_mask=True)
    assert dev.L.fga_gix_write_files(xg.h, root.encode()) == 0, dev.L.fga_last_error()
    dgx.free(); xg.close(); g.close(); dev.close()
print(f"genome + GDB + masked GIX: {time.time()-t:.0f} s", flush=True)
t = time.time()
r, _ = H.ref_fastga(root, None, d, os.path.join(d, "ref"), threads=a.threads, flags=("-M",))
ref_s = time.time() - t
print(f"reference FastGA -M -T{a.threads}: {ref_s:.0f} s", flush=True)
err = r.stderr.replace("\r", "\n")
dig = workload.digest_1aln(H.oneview(os.path.join(d, "ref.1aln")))
dig["reference_seconds"], dig["reference_threads"] = round(ref_s, 1), a.threads
m = re.search(r"Total seeds = (\d+)", err)
dig["total_seeds"] = int(m.group(1)) if m else None
m = re.search(r"Total hits over \d+bp = (\d+), (\d+) aln's, (\d+) non-redundant", err)
dig["hits"], dig["alignments"], dig["nonredundant"] = (int(m.group(k)) for k in (1, 2, 3)) if m else (None,) * 3
dig["generator"] = f"fastga_amd.workload.build_config3(mbp={a.mbp:g}) + oracle/_ref/FastGA -M -T{a.threads}"
out = a.out or os.path.join(HERE, f"config3_{a.mbp:g}m_digest.json")
json.dump(dig, open(out, "w"), indent=1)
print(json.dumps(dig), "->", out)
