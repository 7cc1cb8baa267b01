#!/usr/bin/env python3
"""bench.py -- throughput of the FastGA seed-and-extend hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU.  A *step* is one pass of the hot path -- seed merge -> sort -> chain scan ->
wave extension -> redundancy filter -> .1aln written -- over ONE synthetic genome pair whose 2-bit genomes and GIX
tables are already resident in HBM (uploaded / built on the device before the timed region; no index files).

Workload = BASELINE.json configs[1]: synthetic pair, 2 % divergence, 2.5-Mbp contigs, 5 % repeats,
2 % rearranged blocks, 100 Mbp per genome PER GPU:
  N = 1   100 Mbp x 100 Mbp, 40 contigs -- the configuration the metric is quoted on.
  N > 1   ONE comparison of (N x 100 Mbp) x (N x 100 Mbp), 40 N contigs, cut over the N ranks the way the reference cuts
          it over parts and threads (fastga_amd/parallel.py): every rank merges one 12-mer prefix range of the two
          (replicated) tables, the seeds go to the rank that owns their A contig in one RCCL all-to-all-v, every rank
          sorts / chains / extends its A-contig part, rank 0 gathers the records, filters, orders and writes the .1aln.
          Per-GPU work is fixed as N grows ("scaling": "weak"); `value` = whole-comparison Gbp-pair per second.
`--strong-mbp M` instead fixes the comparison at M Mbp per genome for every N (strong scaling; the extension's critical
path -- the longest contig's serial wave chain -- bounds it).

Rank 0 prints ONE JSON line with, beside the contract's fields:
  roofline      dominant kernel = seed merge: algorithmic bytes (N1 E1 + N2 E2 + S x seed bytes) / HIP-event time of the
                launch, against 8 TB/s; `traffic` = HBM bytes per launch from the rocprofv3 PMC pass of this round
                (profiles/rNN_pmc_summary.csv, written by tools/profile_round.sh with the commit it was taken at)
  extend        the wave-extension kernel (87 % of the step): B_ext = 2 (bases compared + diagonal probes) + 2 trace
                elements, GB/s, wave steps / s, G cell updates / s, wavefronts busy on average
  cold          the same comparison on the span of the reference's "Total Resources" line: GDB files on disk -> genomes
                to HBM -> both indices built on the device -> one step -> .1aln closed (N = 1 only)
  cpu_baseline  the REAL reference FastGA (oracle/_ref) on this box's host cores on the same pair, and -- the parity
                gate -- whether its .1aln is identical to ours (`identical_1aln`; N = 1 only)
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
    ap.add_argument("--mbp", type=float, default=100.0, help="size of each genome per GPU, Mbp")
    ap.add_argument("--strong-mbp", type=float, default=0.0, help="fixed genome size for every N (strong scaling)")
    ap.add_argument("--div", type=float, default=0.02)
    ap.add_argument("--contigs", type=int, default=40, help="contigs per 100 Mbp")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the reference leg (cpu_baseline + parity)")
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="take the multi-GPU code path (process group, exchange, gather) even with one rank")
    ap.add_argument("--self", dest="self_", action="store_true",
                    help="BASELINE configs[2]'s shape instead of the pair: repeat-heavy genome of --mbp against itself, -M")
    ap.add_argument("--no-human-scale", action="store_true",
                    help="skip the 3 Gbp x 3 Gbp leg (BASELINE configs[3]: one comparison, N = 1 only, ~30 s)")
    return ap.parse_args()


def cpu_baseline(args, mbp, ra, rb, workdir, ours_1aln):
    """The REAL reference FastGA (oracle/_ref, built from /root/reference by oracle/Makefile) on this box's host
    cores, on the bench's own pair (GDB from our FASTA producer, .gix/.ktab files written from the device index build, so
    the reference does not spend its wall time in GIXmake), and the comparison of its .1aln with ours; falls back to the
    oracle's seed-merge port if the reference binaries did not travel."""
    from oracle import harness as H
    from fastga_amd import workload
    ncores = os.cpu_count() or 1
    threads = max(1, min(32, ncores))
    if H.have_reference():
        d = os.path.join(workdir, "cpu")
        os.makedirs(d, exist_ok=True)
        t = time.time()
        r, _ = H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=threads)
        dt = time.time() - t
        phases = [ln.strip() for ln in r.stderr.replace("\r", "\n").splitlines() if "Resources" in ln]
        out = {"value": mbp * 1e-3 / dt, "unit": "Gbp-pair/s", "cores": threads, "kind": "reference",
               "sample": f"oracle/_ref/FastGA -T{threads} -1:ref on the bench pair ({mbp:g} Mbp x {mbp:g} Mbp, "
                         f"prebuilt GDB/GIX, tmp on local disk): {dt:.2f} s wall; " + " | ".join(phases)}
        a = H.oneview(ours_1aln)
        b = H.oneview(os.path.join(d, "ref.1aln"))
        out["identical_1aln"] = workload.digest_1aln(a) == workload.digest_1aln(b)
        out["identical_1aln_strict"] = a == b
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
    if world > 1 or args.force_sharded:
        # torch FIRST: one HIP runtime per process (fastga_amd/lib.py::load_library)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if "MASTER_ADDR" not in os.environ:                  # --force-sharded without a launcher
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29517"
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from fastga_amd import workload, device as D

    shared = args.workdir or os.path.join(tempfile.gettempdir(), f"fga_bench_{os.environ.get('MASTER_PORT', os.getpid())}")
    os.makedirs(shared, exist_ok=True)
    mbp = args.strong_mbp if args.strong_mbp > 0 else args.mbp * world
    ncontig = max(world, int(round(args.contigs * mbp / 100.0)))
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    ra, rb = os.path.join(shared, "A"), os.path.join(shared, "B")
    if rank == 0:
        # FASTA -> GDB on the host, once; the two indices are built on the device when a session opens (no .gix files)
        if args.self_:
            ra = workload.build_config3(shared, mbp=mbp, threads=threads, gix=False, name="A")
        else:
            workload.build_config2(shared, mbp=mbp, seed=1, divergence=args.div, ncontig=ncontig, threads=threads, gix=False)
    if args.self_:
        rb = None
    if dist is not None:
        dist.barrier()
    prep_s = time.time() - t0

    # inputs resident in HBM before the timed region: both GIX tables + both 2-bit genomes (every rank holds all of them)
    # (with N ranks: every rank holds the genomes' bases and ITS 12-mer prefix range of the two tables only)
    ses = D.Session(ra, rb, device=local, rank=rank if world > 1 else 0, nranks=world if world > 1 else 1, nthreads=threads)
    out1aln = os.path.join(shared, "bench.1aln")
    kw = dict(out_path=out1aln, nthreads=threads, command_line="bench.py FastGA hot path")
    if args.self_:
        kw["soft_mask"] = True

    if dist is None:
        def step():
            return ses.run(**kw)
    else:
        from fastga_amd.parallel import run_sharded
        dev = f"cuda:{local}"

        def step():
            return run_sharded(ses, dist, kw, dev)

    def barrier():
        ses.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    gix_ms = ses.dev_wrapper().stage_ms(5)            # kernel time of the session's own device build of genome B's index
    do_cpu = (not args.no_cpu) and world == 1 and not args.self_     # the reference leg runs on rank 0 at N=1 only

    for _ in range(args.warmup):
        step()
    barrier()
    t = time.time()
    stats, cut_ms = [], []
    for _ in range(args.steps):
        stats.append(step())
        cut_ms.append(ses.dev_wrapper().stage_ms(0))      # FGA_STAGE_MERGE_PARTITION: range_cut_kernel of the step's merge launch
    barrier()
    elapsed = time.time() - t

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # totals over the ranks (counts only; the records themselves were gathered inside the step)
        keys = ("nseeds", "nhits", "nalns", "ncalls", "nwaves", "ext_cells", "ext_bases", "ext_trace", "part_seeds",
                "exchange_seeds_out")
        v = torch.tensor([int(stats[-1][k]) for k in keys], device="cuda", dtype=torch.int64)
        allv = [torch.zeros_like(v) for _ in range(world)]
        dist.all_gather(allv, v)
        tot = {k: [int(a[i].item()) for a in allv] for i, k in enumerate(keys)}
        km = torch.tensor([sum(s["merge_kernel_ms"] for s in stats) / len(stats),
                           sum(s["extend_kernel_ms"] for s in stats) / len(stats)], device="cuda")
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kavg, kext = float(km[0].item()), float(km[1].item())
    else:
        tot = None
        kavg = sum(s["merge_kernel_ms"] for s in stats) / len(stats)
        kext = sum(s["extend_kernel_ms"] for s in stats) / len(stats)

    if rank == 0:
        last = stats[-1]
        ms_per_step = 1000.0 * elapsed / args.steps
        pair_gbp = 0.5 * (ses.bases[0] + ses.bases[1]) * 1e-9
        value = pair_gbp / (ms_per_step / 1000.0)
        S = (lambda k: sum(tot[k])) if tot is not None else (lambda k: int(last[k]))
        nseeds = S("nseeds")
        # algorithmic bytes of the merge launches of one step: every table entry once in its on-disk width, every seed
        # once in the reference's record width (with N ranks each launch covers 1/N of the prefix space)
        seed_bytes = nseeds * ses.seed_bytes * (2 if args.self_ else 1)    # self: the reported total is halved (FastGA.c:1906)
        # a sliced session's table bytes are rank 0's share already (the ranges are cut for equal cost)
        alg_bytes = ses.table_bytes * (world if ses.nranks > 1 else 1) + seed_bytes
        achieved = alg_bytes / max(1, world) / (kavg * 1e-3) / 1e9          # per GPU, slowest rank's launch time
        stage_ms = {k: round(1000 * sum(s[k] for s in stats) / len(stats), 2)
                    for k in ("merge_s", "sort_s", "chain_s", "extend_s", "filter_s", "write_s")}
        b_ext = 2 * (S("ext_bases") + S("ext_cells")) + 2 * S("ext_trace")
        out = {
            "metric": "Gbp-pair aligned/sec", "value": value, "unit": "Gbp-pair/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong" if args.strong_mbp > 0 else "weak", "vs_baseline": None,
            "dtype": "u8/int32/u64", "data": "synthetic",
            "config": {"workload": (f"synthetic repeat-heavy {mbp:g} Mbp genome against itself, soft mask on (BASELINE configs[2]'s shape)"
                                    if args.self_ else
                                    f"synthetic {mbp:g} Mbp vs {mbp:g} Mbp, {args.div*100:g}% divergence, {ncontig} contigs, "
                                    f"5% repeats, 2% inversions/swaps (BASELINE configs[1]"
                                    + (")" if world == 1 else f" x {world}: ONE comparison over {world} GPUs, "
                                       f"{args.mbp:g} Mbp per genome per GPU)")),
                       "step": "seed merge -> sort -> chain scan -> wave extension -> redundancy filter -> .1aln "
                               "written; GIX tables + genomes resident in HBM"
                               + ("" if world == 1 else "; phase 1 by k-mer prefix range, seeds all-to-all-v by A-contig "
                                  "part (RCCL), phase 2 by part, records gathered to rank 0"),
                       "seeds": int(nseeds), "hits": S("nhits"), "alignments": S("nalns"),
                       "records": int(last["nlive"]), "la_calls": S("ncalls"), "waves": S("nwaves"),
                       "stage_ms": stage_ms,
                       "kernel_ms": {"merge": round(kavg, 3), "sort": round(last["sort_kernel_ms"], 3),
                                     "extend": round(kext, 3)},
                       "prep_s": round(prep_s, 1),
                       "gix_build_on_device_ms": None if gix_ms is None else round(gix_ms, 2)},
            "roofline": {"kernel": "seed merge launch (HIP events around fga_seed_merge's kernels on the library's stream)",
                         "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes": int(alg_bytes // max(1, world)), "kernel_ms": kavg,
                         # the walk kernel alone (the launch less its range_cut_kernel): what the rocprofv3 per-kernel average
                         # of seed_merge_walk_kernel corresponds to
                         "walk_kernel_ms": kavg - sum(cut_ms) / len(cut_ms),
                         "walk_kernel_frac": (alg_bytes / max(1, world)) / ((kavg - sum(cut_ms) / len(cut_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "extend": {"kernel": "extend_kernel", "bound": "latency (one wavefront per unit; longest unit's serial chain)",
                       "kernel_ms": kext, "algorithmic_bytes": int(b_ext),
                       "achieved_GBps": b_ext / (kext * 1e-3) / 1e9 if kext > 0 else None,
                       "frac_of_hbm_peak": b_ext / (kext * 1e-3) / 1e9 / HBM_PEAK_GBS / max(1, world) if kext > 0 else None,
                       "wave_steps_per_s": S("nwaves") / (kext * 1e-3) if kext > 0 else None,
                       "gcell_updates_per_s": S("ext_cells") / (kext * 1e-3) / 1e9 if kext > 0 else None,
                       "avg_wave_width": S("ext_cells") / max(1, S("nwaves")),
                       "avg_busy_wavefronts": round(float(last["ext_busy_waves"]), 1)},
        }
        if tot is not None:
            out["config"]["per_rank"] = {"part_seeds": tot["part_seeds"], "seeds_sent": tot["exchange_seeds_out"],
                                         "alignments": tot["nalns"], "waves": tot["nwaves"]}
        if world == 1 and not args.self_ and abs(mbp - 100.0) < 1e-9 and abs(args.div - 0.02) < 1e-9:    # the profiled configuration
            tr, src = pmc_traffic()
            if tr is not None:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = src
        if world == 1 and not args.no_cold and not args.self_:
            out["cold"] = cold_run(D, ra, rb, shared, threads, pair_gbp)
        if do_cpu:
            try:
                from fastga_amd.gixio import Gdb
                for r in (ra, rb):                    # index files for the reference, from device builds
                    g = Gdb(r + ".gdb")
                    dgx, xg = D.build_gix_device(ses.dev_wrapper(), g, 8, host_copy=True)
                    if ses.L.fga_gix_write_files(xg.h, r.encode()) != 0:
                        raise RuntimeError("cannot write index files for the CPU baseline")
                    dgx.free(); xg.close(); g.close()
                out["cpu_baseline"] = cpu_baseline(args, mbp, ra, rb, shared, out1aln)
            except Exception as e:      # the baseline leg must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "Gbp-pair/s", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {e}"}
        if world == 1 and not args.self_ and not args.no_human_scale:
            ses.close()                           # the 3 Gbp leg wants the whole device
            ses = None
            try:
                out["human_scale"] = human_scale_run(D, workload, shared, threads)
            except Exception as e:                # never takes the bench line down
                out["human_scale"] = {"error": str(e)}
        print(json.dumps(out), flush=True)

    if ses is not None:
        ses.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def human_scale_run(D, workload, workdir, threads):
    """BASELINE configs[3] at its stated size on this one GPU: 3 Gbp x 3 Gbp (32 contigs of ~94 Mbp, 1 % divergence, 45 %
    repeats; fastga_amd.workload.build_config4), both indices built on the device (2.4 G entries each), ONE comparison
    from resident inputs to the .1aln closed.  The counts are the ones the reference gives for this pair
    (tests/golden/config4_3000m_digest.json: made with oracle/_ref/GIXmake + FastGA -T32 in 74 s wall on the GPU box's 256
    host cores; the full record digest is compared in tests/test_full_size_gpu.py).  Seed-merge roofline at this size:
    algorithmic bytes N1 E1 + N2 E2 + S x seed bytes over the HIP-event time of the launch."""
    import shutil
    d = os.path.join(workdir, "human_scale")
    os.makedirs(d, exist_ok=True)
    try:
        t = time.time()
        ra, rb = workload.build_config4(d, mbp=3000.0, divergence=0.01, threads=threads)
        prep = time.time() - t
        t = time.time()
        ses = D.Session(ra, rb)
        opened = time.time() - t
        out = os.path.join(d, "c4.1aln")
        t = time.time()
        st = ses.run(out_path=out, nthreads=threads, command_line="bench.py FastGA 3 Gbp")
        dt = time.time() - t
        alg = ses.table_bytes + st["nseeds"] * ses.seed_bytes
        res = {"workload": "synthetic 3 Gbp vs 3 Gbp, 1% divergence, 32 contigs, 45% repeats (BASELINE configs[3]), 1 GPU",
               "value": 3.0 / dt, "unit": "Gbp-pair/s", "seconds": round(dt, 2), "parts": int(st["nparts"]),
               "seeds": int(st["nseeds"]), "hits": int(st["nhits"]), "alignments": int(st["nalns"]), "records": int(st["nlive"]),
               "stage_s": {k: round(st[k], 2) for k in ("merge_s", "sort_s", "chain_s", "extend_s", "filter_s", "write_s")},
               "kernel_ms": {"merge": round(st["merge_kernel_ms"], 1), "sort": round(st["sort_kernel_ms"], 1),
                             "extend": round(st["extend_kernel_ms"], 1)},
               "roofline": {"kernel": "seed merge launch", "bound": "hbm", "algorithmic_bytes": int(alg),
                            "kernel_ms": st["merge_kernel_ms"], "achieved": alg / (st["merge_kernel_ms"] * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / (st["merge_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
               "hbm_peak_gib": round(st["hbm_peak_bytes"] / 2**30, 1),
               "genomes_s": round(prep, 1), "upload_and_2_index_builds_s": round(opened, 2)}
        gold = os.path.join(ROOT, "tests", "golden", "config4_3000m_digest.json")
        if os.path.exists(gold):
            g = json.load(open(gold))
            res["reference"] = {"seconds": g.get("reference_seconds"), "threads": g.get("reference_threads"),
                                "where": "same box class (256 host cores), tests/golden/config4_3000m_digest.json"}
            res["counts_equal_reference"] = (st["nseeds"] == g["total_seeds"] and st["nhits"] == g["hits"] and
                                             st["nalns"] == g["alignments"] and st["nlive"] == g["records"])
        ses.close()
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cold_run(D, ra, rb, workdir, threads, pair_gbp):
    """One comparison from files on disk to the .1aln closed -- the span of the reference's "Total Resources" line
    (FastGA.c:4828-4829, 5263-5264), which starts with GDB (+ GIX) present: here the two GDBs are read, the bases go to
    HBM, both indices are BUILT on the device (no .gix files exist yet), one step runs, the .1aln is written and
    everything is released again (fga_run).  The files are in the page cache, as they are for the reference leg."""
    out = os.path.join(workdir, "cold.1aln")
    best = None
    for _ in range(2):                  # the second run no longer pays the one-off HIP module / allocator warm-up
        t = time.time()
        st = D.run(ra, rb, out, nthreads=threads, command_line="bench.py FastGA cold")
        dt = time.time() - t
        if best is None or dt < best[0]:
            best = (dt, st)
    dt, st = best
    return {"value": pair_gbp / dt, "unit": "Gbp-pair/s", "ms": round(1000 * dt, 1),
            "span": "GDB on disk -> genomes to HBM -> 2 index builds on the device -> merge/sort/chain/extend/filter -> "
                    ".1aln closed -> resources released (fga_run; best of 2)",
            "load_ms": round(1000 * st["load_s"], 1), "upload_and_index_ms": round(1000 * st["upload_s"], 1),
            "phases_ms": round(1000 * st["phase23_s"], 1)}


def pmc_traffic():
    """HBM bytes per merge launch from the committed rocprofv3 PMC passes of this same command
    (profiles/r*_pmc_summary.csv, regenerated each round by tools/profile_round.sh, which records the commit):
    FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section)
    -> x2; WRITE_SIZE in KiB.  All kernels of the merge launch are added up."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")))
    if not files:
        return None, None
    fetch = write = None
    commit = ""
    for row in csv.reader(ln for ln in open(files[-1]) if not ln.startswith("#")):
        if len(row) < 5 or row[0] == "pass":
            continue
        # pass, kernel, counter, launches, avg_per_launch -- a template kernel's name holds commas of its own
        k, counter, avg = ",".join(row[1:-3]), row[-3], row[-1]
        if "seed_merge" in k or "range_cut" in k:
            if counter == "FETCH_SIZE":
                fetch = (fetch or 0.0) + float(avg)
            elif counter == "WRITE_SIZE":
                write = (write or 0.0) + float(avg)
    for ln in open(files[-1]):
        if ln.startswith("# commit"):
            commit = ln.split(":", 1)[1].strip()
    if fetch is None or write is None:
        return None, None
    src = os.path.relpath(files[-1], ROOT) + (f" @ {commit}" if commit else "")
    return int((2.0 * fetch + write) * 1024), src


if __name__ == "__main__":
    main()
