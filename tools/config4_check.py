#!/usr/bin/env python3
"""tools/config4_check.py -- BASELINE configs[3] / [4] on ONE GPU: the human-scale pair (32 contigs of ~94 Mbp, 3 Gbp per
genome, 1 % or 10 % divergence, 45 % repeats) built by the C generator, indices built on the device, the comparison run by
fga_session_run (multi-pass over A-contig parts when the seeds exceed one sort pass) and, with --parts N, once more as N
prefix ranges x N parts (the multi-GPU cut emulated on one GPU: fastga_amd/parallel.py::run_parts_on_one_gpu).
With --reference the real reference (oracle/_ref: GIXmake + FastGA) runs on the same genomes and the digests are compared;
--golden FILE writes the reference's digest (tests/golden/make_golden_config4.py calls this)."""
import argparse, json, os, re, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=3000.0)
ap.add_argument("--div", type=float, default=0.01)
ap.add_argument("--contigs", type=int, default=32)
ap.add_argument("--repeats", type=float, default=0.45)
ap.add_argument("--nfam", type=int, default=0, help="repeat families (0: one per ~234 kbp)")
ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 8))
ap.add_argument("--parts", type=int, default=0, help="also run as N prefix ranges x N parts on this GPU")
ap.add_argument("--pass-seeds", type=int, default=0)
ap.add_argument("--reference", action="store_true")
ap.add_argument("--ref-threads", type=int, default=0)
ap.add_argument("--golden", default=None)
ap.add_argument("--workdir", default=None)
ap.add_argument("--no-gpu", action="store_true", help="reference only (golden made on a box without a GPU)")
ap.add_argument("--keep", action="store_true")
ap.add_argument("--no-digest", action="store_true", help="skip the ONEview digest of our .1aln (profiling runs)")
ap.add_argument("--runs", type=int, default=1, help="comparisons of the session (the later ones find its buffers in place)")
a = ap.parse_args()

def sh(cmd):
    return subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout.strip()

print("box:", sh("nproc"), "cores |", sh("free -g | sed -n 2p"), "|", sh("df -h --output=avail,target /tmp /dev/shm . | tr '\\n' ' '"), flush=True)
from fastga_amd import workload
from oracle import harness as H

d = a.workdir or tempfile.mkdtemp(prefix="fga_c4_")
os.makedirs(d, exist_ok=True)
t = time.time()
ra, rb = workload.build_config4(d, mbp=a.mbp, divergence=a.div, ncontig=a.contigs, repeat_frac=a.repeats, nfam=a.nfam or None,
                                threads=a.threads)
print(f"genomes + GDBs: {time.time()-t:.1f} s", flush=True)
oneview = H.ref_bin("ONEview")
res = {"mbp": a.mbp, "div": a.div}
ours = os.path.join(d, "ours.1aln")
if not a.no_gpu:
    from fastga_amd import device as D, parallel
    t = time.time()
    ses = D.Session(ra, rb)
    print(f"upload + 2 device index builds: {time.time()-t:.2f} s, tables {ses.table_bytes/1e9:.1f} GB", flush=True)
    for rep in range(max(1, a.runs)):
        print(f"---- comparison {rep+1} of the session", file=sys.stderr, flush=True)
        t = time.time()
        st = ses.run(out_path=ours, nthreads=a.threads, pass_seeds=a.pass_seeds, reference_threads=a.ref_threads or a.threads)
        dt = time.time() - t
    print(f"fga_session_run: {dt:.2f} s = {a.mbp*1e-3/dt:.2f} Gbp-pair/s | seeds {st['nseeds']} hits {st['nhits']} units {st['nunits']} "
          f"alns {st['nalns']} records {st['nlive']} waves {st['nwaves']} parts {st['nparts']} peak HBM {st['hbm_peak_bytes']/2**30:.1f} GiB", flush=True)
    print("   stages s:", {k: round(st[k], 2) for k in ("merge_s", "sort_s", "chain_s", "extend_s", "filter_s", "write_s")},
          "kernels ms:", {k: round(st[k], 1) for k in ("merge_kernel_ms", "sort_kernel_ms", "extend_kernel_ms")}, flush=True)
    alg = ses.table_bytes + st["nseeds"] * ses.seed_bytes
    print(f"   seed merge: {alg/1e9:.1f} GB algorithmic in {st['merge_kernel_ms']:.1f} ms = {alg/st['merge_kernel_ms']/1e6:.0f} GB/s "
          f"= {alg/st['merge_kernel_ms']/1e6/8000:.3f} of 8 TB/s", flush=True)
    res["ours"] = {k: st[k] for k in ("nseeds", "nhits", "nunits", "nalns", "nlive", "nwaves", "nparts", "hbm_peak_bytes")}
    res["ours"]["seconds"] = dt
    if os.path.exists(oneview) and not a.no_digest:
        t = time.time()
        res["ours_digest"] = workload.digest_1aln_stream(ours, oneview)
        print(f"   digest ({time.time()-t:.0f} s): {res['ours_digest']}", flush=True)
    if a.parts > 1:
        out2 = os.path.join(d, "parts.1aln")
        t = time.time()
        st2 = parallel.run_parts_on_one_gpu(ses, a.parts, out_path=out2, nthreads=a.threads,
                                            reference_threads=a.ref_threads or a.threads)
        print(f"{a.parts} ranges x {a.parts} parts: {time.time()-t:.2f} s | seeds {st2['nseeds']} records {st2['nlive']} "
              f"part seeds {st2['part_seed_counts']}", flush=True)
        if os.path.exists(oneview):
            dg2 = workload.digest_1aln_stream(out2, oneview)
            res["parts_digest"] = dg2
            print("   parts digest == one-pass digest:", dg2 == res.get("ours_digest"), flush=True)
        os.unlink(out2)
    ses.close()

if a.reference:
    T = a.ref_threads or a.threads
    t = time.time()
    for r in (ra, rb):
        H.run([H.ref_bin("GIXmake"), f"-T{T}", f"-P{d}", r], cwd=d)
    print(f"reference GIXmake -T{T} x 2: {time.time()-t:.0f} s", flush=True)
    t = time.time()
    r, _ = H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=T)
    rs = time.time() - t
    err = r.stderr.replace("\r", "\n")
    print(f"reference FastGA -T{T}: {rs:.0f} s wall", flush=True)
    for ln in err.splitlines():
        if "Resources" in ln or "Total seeds" in ln or "Total hits" in ln:
            print("   ", ln.strip(), flush=True)
    dg = workload.digest_1aln_stream(os.path.join(d, "ref.1aln"), oneview)
    m = re.search(r"Total seeds = (\d+)", err)
    dg["total_seeds"] = int(m.group(1)) if m else None
    m = re.search(r"Total hits over \d+bp = (\d+), (\d+) aln's, (\d+) non-redundant", err)
    dg["hits"], dg["alignments"], dg["nonredundant"] = (int(m.group(k)) for k in (1, 2, 3)) if m else (None,) * 3
    dg["reference_seconds"], dg["reference_threads"] = round(rs, 1), T
    dg["generator"] = (f"fastga_amd.workload.build_config4(mbp={a.mbp:g}, divergence={a.div:g}, ncontig={a.contigs}, "
                       f"repeat_frac={a.repeats:g}, nfam={a.nfam or 'default'}) + oracle/_ref/GIXmake -T{T} + oracle/_ref/FastGA -T{T}")
    res["reference_digest"] = dg
    print("   reference digest:", dg, flush=True)
    if "ours_digest" in res:
        same = all(res["ours_digest"][k] == dg[k] for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5"))
        print("   ours == reference:", same, flush=True)
        res["identical"] = same
    if a.golden:
        json.dump(dg, open(a.golden, "w"), indent=1)
print(json.dumps(res))
if not a.keep and a.workdir is None:
    shutil.rmtree(d, ignore_errors=True)
