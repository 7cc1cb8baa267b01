#!/usr/bin/env python3
"""tools/config6g_check.py -- tables of more than 2^32 entries: a 6 Gbp genome (64 contigs of ~94 Mbp, 1 % repeats: 4.8 G index
entries) on ONE GPU,
  self   the genome against itself (the self kernel over one table beyond 2^32)
  pair   the first 4 contigs of its 1 %-diverged copy (376 Mbp) against it (table 2 of a pair comparison beyond 2^32)
both with the index built on the device.  With --reference the real reference (oracle/_ref: GIXmake + FastGA) runs on the same
genomes and the digests are compared; --golden-dir writes the reference's digests (tests/golden/config6g_{self,pair}_digest.json),
which tests/test_full_size_gpu.py compares against."""
import argparse, json, os, re, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--mbp", type=float, default=6000.0)
ap.add_argument("--contigs", type=int, default=64)
ap.add_argument("--small-contigs", type=int, default=4)
ap.add_argument("--repeats", type=float, default=0.01)
ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 8))
ap.add_argument("--modes", default="pair,self")
ap.add_argument("--reference", action="store_true")
ap.add_argument("--golden-dir", default=None)
ap.add_argument("--workdir", default=None)
ap.add_argument("--no-gpu", action="store_true")
a = ap.parse_args()


if __name__ == "__main__":
    from fastga_amd import workload
    from oracle import harness as H
    d = a.workdir or tempfile.mkdtemp(prefix="fga_6g_")
    os.makedirs(d, exist_ok=True)
    t = time.time()
    rg, rs = workload.build_config6g(d, a.mbp, a.contigs, a.small_contigs, a.repeats, a.threads)
    print(f"genomes + GDBs: {time.time()-t:.1f} s", flush=True)
    oneview = H.ref_bin("ONEview")
    res = {}
    modes = a.modes.split(",")
    if not a.no_gpu:
        from fastga_amd import device as D
        for mode in modes:
            ours = os.path.join(d, f"ours_{mode}.1aln")
            t = time.time()
            ses = D.Session(rs, rg, nthreads=a.threads) if mode == "pair" else D.Session(rg, None, nthreads=a.threads)
            print(f"[{mode}] upload + device index build(s): {time.time()-t:.2f} s, tables {ses.table_bytes/1e9:.1f} GB", flush=True)
            t = time.time()
            st = ses.run(out_path=ours, nthreads=a.threads, reference_threads=a.threads)
            dt = time.time() - t
            print(f"[{mode}] fga_session_run: {dt:.2f} s | seeds {st['nseeds']} hits {st['nhits']} alns {st['nalns']} records {st['nlive']} "
                  f"parts {st['nparts']} peak HBM {st['hbm_peak_bytes']/2**30:.1f} GiB; kernels ms merge {st['merge_kernel_ms']:.1f} "
                  f"sort {st['sort_kernel_ms']:.1f} extend {st['extend_kernel_ms']:.1f}", flush=True)
            res[mode] = {"ours": {k: st[k] for k in ("nseeds", "nhits", "nalns", "nlive")}, "seconds": dt}
            ses.close()
            if os.path.exists(oneview):
                res[mode]["ours_digest"] = workload.digest_1aln_stream(ours, oneview)
                print(f"[{mode}] digest: {res[mode]['ours_digest']}", flush=True)
    if a.reference:
        T = a.threads
        t = time.time()
        for r in (rg, rs):
            H.run([H.ref_bin("GIXmake"), f"-T{T}", f"-P{d}", r], cwd=d)
        print(f"reference GIXmake -T{T} x 2: {time.time()-t:.0f} s", flush=True)
        for mode in modes:
            t = time.time()
            out = os.path.join(d, f"ref_{mode}")
            r, _ = H.ref_fastga(rs, rg, d, out, threads=T) if mode == "pair" else H.ref_fastga(rg, None, d, out, threads=T)
            secs = time.time() - t
            err = r.stderr.replace("\r", "\n")
            dg = workload.digest_1aln_stream(out + ".1aln", oneview)
            m = re.search(r"Total seeds = (\d+)", err)
            dg["total_seeds"] = int(m.group(1)) if m else None
            m = re.search(r"Total hits over \d+bp = (\d+), (\d+) aln's, (\d+) non-redundant", err)
            dg["hits"], dg["alignments"], dg["nonredundant"] = (int(m.group(k)) for k in (1, 2, 3)) if m else (None,) * 3
            dg["reference_seconds"], dg["reference_threads"] = round(secs, 1), T
            dg["generator"] = (f"fastga_amd.workload.build_config6g(mbp={a.mbp:g}, contigs={a.contigs}, small={a.small_contigs}, repeats={a.repeats:g}) "
                               f"+ oracle/_ref/GIXmake -T{T} + oracle/_ref/FastGA -T{T} ({mode})")
            print(f"[{mode}] reference FastGA -T{T}: {secs:.0f} s; digest {dg}", flush=True)
            res.setdefault(mode, {})["reference_digest"] = dg
            if "ours_digest" in res[mode]:
                same = all(res[mode]["ours_digest"][k] == dg[k] for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5"))
                print(f"[{mode}] ours == reference: {same}", flush=True)
                res[mode]["identical"] = same
            if a.golden_dir:
                json.dump(dg, open(os.path.join(a.golden_dir, f"config6g_{mode}_digest.json"), "w"), indent=1)
    print(json.dumps(res))
    if a.workdir is None:
        shutil.rmtree(d, ignore_errors=True)
