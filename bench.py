#!/usr/bin/env python3
"""bench.py -- throughput of the FastGA seed-and-extend hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU.  A *step* is one pass of the hot path over one synthetic genome pair
already resident in HBM (2-bit genomes uploaded and both GIX tables built on the device -- fga_dgix_build -- before the
timed region; no index files are involved).  Workload = BASELINE.json
configs[1]: synthetic 100 Mbp vs 100 Mbp, 2 % divergence, 40 contigs, repeats + rearrangements.
Weak scaling: every rank owns its own pair (different seed) -- contig-pair work units are independent, so
there is no data-path collective; only the per-rank record counts are gathered.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel = seed merge, algorithmic bytes / HIP-event time)
and `cpu_baseline` (the real reference FastGA from oracle/_ref, or the oracle port, timed on this box's cores on
a bounded sample).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mbp", type=float, default=100.0, help="size of each genome of the pair, Mbp")
    ap.add_argument("--div", type=float, default=0.02)
    ap.add_argument("--contigs", type=int, default=40)
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--verify", action="store_true", help="also compare our .1aln with the reference's (ONEview)")
    return ap.parse_args()


def cpu_baseline(args, ra, rb, workdir, verify_against=None):
    """The REAL reference FastGA (oracle/_ref, built from /root/reference by oracle/Makefile) on this box's host
    cores, on the bench's own pair (GDB from our FASTA producer, .gix/.ktab files written from the device index build, so
    the reference does not spend its wall time in GIXmake); falls back to a smaller pair if the bench pair is large, and
    to the oracle's seed-merge port if the reference binaries did not travel."""
    from oracle import harness as H
    ncores = os.cpu_count() or 1
    threads = max(1, min(32, ncores))
    mbp = args.mbp
    if H.have_reference():
        d = os.path.join(workdir, "cpu")
        os.makedirs(d, exist_ok=True)
        if mbp > 150:                       # keep the baseline leg to tens of seconds
            from fastga_amd import workload
            mbp = 100.0
            ra, rb = workload.build_pair(d, seed=4242, ncontig=max(threads, 40), total=int(mbp * 1e6),
                                         divergence=args.div, repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02,
                                         threads=threads)
        t = time.time()
        r, _ = H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=threads)
        dt = time.time() - t
        phases = [ln.strip() for ln in r.stderr.replace("\r", "\n").splitlines() if "Resources" in ln]
        out = {"value": mbp * 1e-3 / dt, "unit": "Gbp-pair/s", "cores": threads, "kind": "reference",
               "sample": f"oracle/_ref/FastGA -T{threads} -1:ref on the bench pair ({mbp:g} Mbp x {mbp:g} Mbp, "
                         f"prebuilt GDB/GIX, tmp on local disk): {dt:.2f} s wall; " + " | ".join(phases)}
        if verify_against is not None and mbp == args.mbp:
            a = H.oneview(verify_against)
            b = H.oneview(os.path.join(d, "ref.1aln"))
            out["identical_1aln"], out["identical_1aln_strict"] = same_1aln(a, b)
            out["records"] = sum(1 for ln in b if ln.startswith("A "))
        return out
    from fastga_amd.gixio import Gix
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    npre = 1 << 20                          # 1/16 of the prefix space
    t = time.time()
    H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte, pfirst=0, plast=npre)
    dt = time.time() - t
    return {"value": mbp * 1e-3 / 16 / dt, "unit": "Gbp-pair/s", "cores": 1, "kind": "port",
            "sample": f"oracle seed-merge restatement only, 1/16 of the k-mer prefix space, {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from fastga_amd import workload, device as D

    workdir = args.workdir or tempfile.mkdtemp(prefix=f"fga_bench_r{rank}_")
    os.makedirs(workdir, exist_ok=True)
    total = int(args.mbp * 1e6)
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    # FASTA -> GDB on the host; the two indices are built on the device when the session opens (no .gix files)
    ra, rb = workload.build_pair(workdir, seed=1 + rank, ncontig=args.contigs, total=total,
                                 divergence=args.div, repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02,
                                 threads=threads, gix=False)
    prep_s = time.time() - t0

    # inputs resident in HBM before the timed region: both GIX tables + both 2-bit genomes
    ses = D.Session(ra, rb, device=local)
    out1aln = os.path.join(workdir, f"bench_r{rank}.1aln")

    def step():
        return ses.run(out_path=out1aln, nthreads=threads, command_line="bench.py FastGA hot path")

    def barrier():
        ses.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # informational (not part of `value`): kernel time of one device index build (HIP events of the session's own build
    # of genome B); rank 0 also writes the .gix/.ktab files of both genomes from device builds, for the reference
    # FastGA of the cpu_baseline leg
    gix_ms = ses.dev_wrapper().stage_ms(5)
    do_cpu = (not args.no_cpu) and world == 1          # the CPU baseline leg runs on rank 0 at N=1 only
    if rank == 0 and do_cpu:
        from fastga_amd.gixio import Gdb
        for r in (ra, rb):
            g = Gdb(r + ".gdb")
            dgx, xg = D.build_gix_device(ses.dev_wrapper(), g, 8, host_copy=True)
            if ses.L.fga_gix_write_files(xg.h, r.encode()) != 0:
                raise RuntimeError("cannot write index files for the CPU baseline")
            dgx.free(); xg.close(); g.close()

    for _ in range(args.warmup):
        step()
    barrier()
    t = time.time()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    barrier()
    elapsed = time.time() - t

    nrec = stats[-1]["nlive"]
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the only cross-rank traffic of the path: gather the per-rank record counts (RCCL over xGMI)
        cnt = torch.tensor([nrec], device="cuda", dtype=torch.int64)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        nrec = int(sum(int(c.item()) for c in allc))

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        pair_gbp = 0.5 * (ses.bases[0] + ses.bases[1]) * 1e-9
        value = world * pair_gbp / (ms_per_step / 1000.0)
        last = stats[-1]
        nseeds = last["nseeds"]
        alg_bytes = ses.table_bytes + nseeds * ses.seed_bytes
        kavg = sum(s["merge_kernel_ms"] for s in stats) / len(stats)
        achieved = alg_bytes / (kavg * 1e-3) / 1e9
        stage_ms = {k: round(1000 * sum(s[k] for s in stats) / len(stats), 2)
                    for k in ("merge_s", "sort_s", "download_s", "chain_s", "extend_s", "filter_s", "write_s")}
        out = {
            "metric": "Gbp-pair aligned/sec", "value": value, "unit": "Gbp-pair/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32/u64",
            "data": "synthetic",
            "config": {"workload": f"synthetic {args.mbp:g} Mbp vs {args.mbp:g} Mbp, {args.div*100:g}% divergence, "
                                   f"{args.contigs} contigs, 5% repeats, 2% inversions/swaps (BASELINE configs[1]); "
                                   f"one pair per GPU",
                       "step": "seed merge -> sort -> chain scan -> wave extension -> redundancy filter -> .1aln "
                               "written; GIX tables + genomes resident in HBM",
                       "seeds": int(nseeds), "hits": int(last["nhits"]), "alignments": int(last["nalns"]),
                       "records": int(nrec), "la_calls": int(last["ncalls"]), "waves": int(last["nwaves"]),
                       "stage_ms": stage_ms,
                       "kernel_ms": {"merge": round(kavg, 3), "sort": round(last["sort_kernel_ms"], 3),
                                     "extend": round(last["extend_kernel_ms"], 3)},
                       "prep_s": round(prep_s, 1),
                       "gix_build_on_device_ms": None if gix_ms is None else round(gix_ms, 2)},
            "roofline": {"kernel": "seed_merge_wave_kernel (+ seed_merge_kernel on oversize tiles, hole closing; "
                                   "HIP events around the whole merge launch)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes": int(alg_bytes), "kernel_ms": kavg},
        }
        if abs(args.mbp - 100.0) < 1e-9 and abs(args.div - 0.02) < 1e-9:    # the profiled configuration
            tr, src = pmc_traffic()
            if tr is not None:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = src
        if do_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args, ra, rb, workdir,
                                                   verify_against=out1aln if args.verify else None)
            except Exception as e:      # the baseline leg must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "Gbp-pair/s", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)

    ses.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def same_1aln(a, b):
    """(same, strictly same) for two ONEview texts.  The reference's la_merge breaks ties on (aread, abpos) by the memory
    address of the record, i.e. by the thread slot whose file held it (MAPARE, FastGA.c:3906-3918), so its own output
    order on such ties changes with -T; `same` therefore means: identical header, identical records as a multiset and
    identical (aread, abpos) sequence.  `strict` is line-by-line equality."""
    keep = lambda lines: [ln for ln in lines if ln[0] not in "!<"]      # noqa: E731
    a, b = keep(a), keep(b)
    if a == b:
        return True, True

    def split(lines):
        first = next((i for i, ln in enumerate(lines) if ln.startswith("A ")), len(lines))
        recs, cur = [], []
        for ln in lines[first:]:
            if ln.startswith("A ") and cur:
                recs.append(tuple(cur))
                cur = []
            cur.append(ln)
        if cur:
            recs.append(tuple(cur))
        return lines[:first], recs

    ha, ra = split(a)
    hb, rb = split(b)
    if ha != hb or sorted(ra) != sorted(rb):
        return False, False
    ka = [tuple(int(v) for v in r[0].split()[1:3]) for r in ra]
    kb = [tuple(int(v) for v in r[0].split()[1:3]) for r in rb]
    return ka == kb, False


def pmc_traffic():
    """HBM bytes per seed_merge_kernel launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r*_pmc_summary.csv): FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on
    gfx950 (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE in KiB."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")))
    if not files:
        return None, None
    # the merge is two kernels per launch: seed_merge_wave_kernel (one wavefront per tile) and seed_merge_kernel
    # (workgroup per tile, only the oversize tiles the wave kernel queued); their traffic adds up
    fetch = write = None
    for r in csv.DictReader(open(files[-1])):
        if "seed_merge_wave_kernel" in r["kernel"] or "seed_merge_kernel" in r["kernel"]:
            if r["counter"] == "FETCH_SIZE":
                fetch = (fetch or 0.0) + float(r["avg_per_launch"])
            elif r["counter"] == "WRITE_SIZE":
                write = (write or 0.0) + float(r["avg_per_launch"])
    if fetch is None or write is None:
        return None, None
    return int((2.0 * fetch + write) * 1024), os.path.relpath(files[-1], ROOT)


def A_seqtot(root):
    from fastga_amd.gixio import Gdb
    g = Gdb(root + ".gdb")
    n = g.seqtot
    g.close()
    return n


if __name__ == "__main__":
    main()
