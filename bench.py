#!/usr/bin/env python3
"""bench.py -- throughput of the FastGA seed-and-extend hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it under
torch.distributed.run, one rank per GPU -- and when it is started WITHOUT a launcher it starts its N ranks itself the same
way (WORLD_SIZE must equal --gpus, anything else is refused).  A *step* is one pass of the hot path -- seed merge -> sort -> chain scan ->
wave extension -> redundancy filter -> .1aln written -- over ONE synthetic genome pair whose 2-bit genomes and GIX
tables are already resident in HBM (uploaded / built on the device before the timed region; no index files).

Workload = BASELINE.json configs[1]: synthetic pair, 2 % divergence, 2.5-Mbp contigs, 5 % repeats,
2 % rearranged blocks, 100 Mbp per genome PER GPU:
  N = 1   100 Mbp x 100 Mbp, 40 contigs -- the configuration the metric is quoted on.
  N > 1   ONE comparison of (N x 100 Mbp) x (N x 100 Mbp), 40 N contigs, cut over the N GPUs the way the reference cuts
          it over parts and threads, by the C library itself (fga_multi_open / fga_multi_run, fastga_amd/csrc/fga_multi.c --
          what `FastGA -G<N>` runs): the launcher's rank 0 holds the session -- one host thread + HIP stream per device, every
          device its 12-mer prefix range of both tables, the seeds to the device that owns their A contig with
          hipMemcpyPeerAsync, every device sorts / chains / extends / filters its A-contig part, the .1aln written once -- and
          a step is ONE fga_multi_run; the launcher's other ranks only stand at the barriers (a gloo group: no kernel of theirs
          on the GPUs that are being timed).  No torch and no collective library in the timed path.
          Per-GPU work is fixed as N grows ("scaling": "weak"); `value` = whole-comparison Gbp-pair per second.
          `--devices 0,0` (one process, no launcher) takes the same path with ranks sharing a GPU: a code-path check.
`--strong-mbp M` instead fixes the comparison at M Mbp per genome for every N (strong scaling; the extension's critical
path -- the longest contig's serial wave chain -- bounds it).

Rank 0 prints ONE JSON line with, beside the contract's fields:
  roofline      dominant kernel = seed merge: algorithmic bytes (N1 E1 + N2 E2 + S x seed bytes) / HIP-event time of the
                launch, against 8 TB/s; `traffic` = HBM bytes per launch from the rocprofv3 PMC pass of this round
                (profiles/rNN_pmc_summary.csv, written by tools/profile_round.sh with the commit it was taken at)
  extend        the wave-extension kernel (87 % of the step): B_ext = 2 (bases compared + diagonal probes) + 2 trace
                elements, GB/s, wave steps / s, G cell updates / s, wavefronts busy on average
  sort          the radix sort of the step: 2 x 16 B x keys x passes over its HIP-event time, against 8 TB/s
  cold          the same comparison on the span of the reference's "Total Resources" line: GDB files on disk -> genomes
                to HBM -> both indices built on the device -> one step -> .1aln closed (N = 1 only)
  cpu_baseline  the REAL reference FastGA (oracle/_ref) on this box's host cores on the same pair (at -T32 and at the best of
                -T64 / -T128 where the box has the cores), and -- the parity gate -- whether its .1aln is identical to ours,
                line for line (`identical_1aln_strict`; the step asks for the reference's tie order, N = 1 only)
  human_scale / human_scale_10pct   BASELINE configs[3] / [4] at their stated 3 Gbp x 3 Gbp on this ONE GPU: warm comparison,
                the cold span beside the reference's wall time, roofline blocks of merge / sort / extension, and
                `projected_8gpu`: the 8-GPU wall time put together from the comparison run as 8 prefix ranges x 8 parts on
                this GPU (a projection, labelled so; the 8-GPU run itself is the driver's)
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
    ap.add_argument("--devices", default=None,
                    help="N = 1 only: run the step through fga_multi_run over these HIP devices (e.g. 0,0: two ranks sharing "
                         "GPU 0) -- the N > 1 code path on a one-GPU box")
    ap.add_argument("--self", dest="self_", action="store_true",
                    help="BASELINE configs[2]'s shape instead of the pair: repeat-heavy genome of --mbp against itself, -M")
    ap.add_argument("--batch", type=int, default=8,
                    help="independent comparisons in flight on the one GPU for the `batch` block (N = 1 only; 0/1: skip)")
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
        out["identical_1aln_strict"] = a == b              # the step ran with reference_threads = this -T
        out["records"] = sum(1 for ln in b if ln.startswith("A "))
        # the reference at more threads, where the box has them (its index files were made for -T<threads>: the search
        # phase takes any -T): the fastest is the figure to beat
        by_t = {threads: round(dt, 2)}
        for tt in (64, 128):
            if tt <= ncores and tt <= args.contigs * mbp / 100.0:      # the reference wants -T <= the number of contigs
                t = time.time()
                try:
                    H.ref_fastga(ra, rb, d, os.path.join(d, f"ref{tt}"), threads=tt)
                    by_t[tt] = round(time.time() - t, 2)
                except Exception:
                    by_t[tt] = None
        out["seconds_by_threads"] = by_t
        ok = {k: v for k, v in by_t.items() if v}
        bt = min(ok, key=ok.get)
        out["best"] = {"threads": bt, "seconds": ok[bt], "value": mbp * 1e-3 / ok[bt]}
        return out
    from fastga_amd.gixio import Gix
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    npre = 1 << 20                          # 1/16 of the prefix space
    t = time.time()
    H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte, pfirst=0, plast=npre)
    dt = time.time() - t
    return {"value": mbp * 1e-3 / 16 / dt, "unit": "Gbp-pair/s", "cores": 1, "kind": "port",
            "sample": f"oracle seed-merge restatement only, 1/16 of the k-mer prefix space, {dt:.1f} s"}


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly the way the
    driver does (one process per GPU under torch.distributed.run on 127.0.0.1), and hand its exit code back."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s): one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    dist = None
    multi_devs = None              # the step is fga_multi_run over these devices (rank 0 holds the session)
    if world > 1:
        # torch FIRST: one HIP runtime per process (fastga_amd/lib.py::load_library).  The process group is the launcher's
        # control plane only -- barriers around the timed region -- on gloo: a RCCL barrier is a kernel spinning on every
        # GPU, and the GPUs are rank 0's to time
        import torch
        import torch.distributed as dist
        multi_devs = tuple(range(world))
        if os.environ.get("FGA_BENCH_DEVICE_MAP"):       # a check of the launcher path on a box with fewer GPUs than ranks: "0,0"
            multi_devs = tuple(int(x) for x in os.environ["FGA_BENCH_DEVICE_MAP"].split(","))[:world]
        torch.cuda.set_device(multi_devs[local])
        dist.init_process_group("gloo")
    elif args.devices:
        multi_devs = tuple(int(x) for x in args.devices.split(","))
    ndev = len(multi_devs) if multi_devs else 1

    from fastga_amd import workload, device as D

    shared = args.workdir or os.path.join(tempfile.gettempdir(), f"fga_bench_{os.environ.get('MASTER_PORT', os.getpid())}")
    os.makedirs(shared, exist_ok=True)
    mbp = args.strong_mbp if args.strong_mbp > 0 else args.mbp * ndev
    ncontig = max(ndev, int(round(args.contigs * mbp / 100.0)))
    threads = max(1, min(32 * ndev, (os.cpu_count() or 8)))      # one process: the session shares them out among its ranks
    if multi_devs is None:
        threads = min(32, threads)
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
    # (with N devices: every device holds the genomes' bases and ITS 12-mer prefix range of the two tables only)
    ses = None
    if multi_devs is None:
        ses = D.Session(ra, rb, device=local, nthreads=threads)
    elif rank == 0:
        ses = D.Multi(ra, rb, devices=multi_devs, nthreads=threads)
    out1aln = os.path.join(shared, "bench.1aln")
    # reference_threads: records that tie on (aread, abpos) in the order FastGA -T<threads> writes them (the reference leg
    # below runs with that -T): the parity gate is line equality
    kw = dict(out_path=out1aln, nthreads=threads, command_line="bench.py FastGA hot path", reference_threads=min(32, threads))
    if args.self_:
        kw["soft_mask"] = True

    def step():
        return ses.run(**kw) if ses is not None else None      # (the launcher's other ranks: the barriers only)

    def barrier():
        if multi_devs is None:
            ses.sync()                                # (fga_multi_run returns with the .1aln closed: nothing in flight)
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    gix_ms = ses.dev_wrapper().stage_ms(5) if multi_devs is None else None   # kernel time of the session's device build of genome B's index
    do_cpu = (not args.no_cpu) and ndev == 1 and not args.self_     # the reference leg runs on rank 0 at N=1 only

    for _ in range(args.warmup):
        step()
    barrier()
    t = time.time()
    stats, cut_ms = [], []
    rank_stats = []
    for _ in range(args.steps):
        stats.append(step())
        if multi_devs is None:
            cut_ms.append(ses.dev_wrapper().stage_ms(0))      # FGA_STAGE_MERGE_PARTITION: range_cut_kernel of the step's merge launch
        elif ses is not None:
            rank_stats.append(ses.rank_stats())
    barrier()
    elapsed = time.time() - t

    if dist is not None:                              # the contract's MAX over the launcher's ranks
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    tot = None
    if rank == 0:
        # (fga_multi_run's stats are totals over its ranks already; its kernel times the slowest rank's)
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
        world = ndev                                     # (below: the GPUs the comparison ran on)
        alg_bytes = ses.table_bytes + seed_bytes         # (a multi-GPU session: its ranks' slices add up to the two tables)
        achieved = alg_bytes / max(1, world) / (kavg * 1e-3) / 1e9          # per GPU, slowest rank's launch time
        stage_ms = {k: round(1000 * sum(s[k] for s in stats) / len(stats), 2)
                    for k in ("merge_s", "sort_s", "chain_s", "extend_s", "filter_s", "write_s")}
        b_ext = 2 * (S("ext_bases") + S("ext_cells")) + 2 * S("ext_trace")
        out = {
            "metric": "Gbp-pair aligned/sec", "value": value, "unit": "Gbp-pair/s",
            "n_gpus": len(set(multi_devs)) if multi_devs else 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
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
                               + ("" if world == 1 else "; ONE fga_multi_run (C, one process, one host thread + stream per "
                                  "device): phase 1 by k-mer prefix range, seeds to the device that owns their A contig "
                                  "(hipMemcpyPeerAsync), phase 2 + filter by part, one .1aln"),
                       "devices": list(multi_devs) if multi_devs else [local], "ranks": ndev,
                       "seeds": int(nseeds), "hits": S("nhits"), "alignments": S("nalns"),
                       "records": int(last["nlive"]), "la_calls": S("ncalls"), "waves": S("nwaves"),
                       "stage_ms": stage_ms,
                       "kernel_ms": {"merge": round(kavg, 3), "sort": round(last["sort_kernel_ms"], 3),
                                     "chain": round(last["chain_kernel_ms"], 3), "extend": round(kext, 3)},
                       "prep_s": round(prep_s, 1),
                       "gix_build_on_device_ms": None if gix_ms is None else round(gix_ms, 2)},
            "roofline": {"kernel": "seed merge launch (HIP events around fga_seed_merge's kernels on the library's stream)",
                         "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes": int(alg_bytes // max(1, world)), "kernel_ms": kavg},
            "sort": sort_block(last["sort_keys"] if world == 1 else None, last["sort_passes"], last["sort_kernel_ms"]),
            "extend": {"kernel": "extend_kernel", "bound": "latency (one wavefront per unit; longest unit's serial chain)",
                       "kernel_ms": kext, "algorithmic_bytes": int(b_ext),
                       "achieved_GBps": b_ext / (kext * 1e-3) / 1e9 if kext > 0 else None,
                       "frac_of_hbm_peak": b_ext / (kext * 1e-3) / 1e9 / HBM_PEAK_GBS / max(1, world) if kext > 0 else None,
                       "wave_steps_per_s": S("nwaves") / (kext * 1e-3) if kext > 0 else None,
                       "gcell_updates_per_s": S("ext_cells") / (kext * 1e-3) / 1e9 if kext > 0 else None,
                       "avg_wave_width": S("ext_cells") / max(1, S("nwaves")),
                       "avg_busy_wavefronts": round(float(last["ext_busy_waves"]), 1)},
        }
        if cut_ms:
            # the walk kernel alone (the launch less its range_cut_kernel): what the rocprofv3 per-kernel average of
            # seed_merge_walk_kernel corresponds to
            out["roofline"]["walk_kernel_ms"] = kavg - sum(cut_ms) / len(cut_ms)
            out["roofline"]["walk_kernel_frac"] = alg_bytes / ((kavg - sum(cut_ms) / len(cut_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS
        if rank_stats:
            rs = rank_stats[-1]
            ext = [r["extend_kernel_ms"] for r in rs]
            out["config"]["per_rank"] = {k: [round(r[k], 4) for r in rs] for k in ("phase1_s", "exchange_s", "phase2_s")}
            out["config"]["per_rank"]["extend_kernel_ms"] = [round(x, 2) for x in ext]
            out["config"]["per_rank"]["wave_steps"] = [int(r["wave_steps"]) for r in rs]
            out["config"]["per_rank"]["part_imbalance_extend"] = max(ext) / (sum(ext) / len(ext)) if sum(ext) > 0 else None
            if len(rank_stats) > 1:               # the first step of the session deals the contigs by seed counts, the later ones by wave steps
                e0 = [r["extend_kernel_ms"] for r in rank_stats[0]]
                out["config"]["per_rank"]["part_imbalance_extend_first_timed_step"] = max(e0) / (sum(e0) / len(e0)) if sum(e0) > 0 else None
        if multi_devs is None and not args.self_ and abs(mbp - 100.0) < 1e-9 and abs(args.div - 0.02) < 1e-9:    # the profiled configuration
            tr, src = pmc_traffic()
            if tr is not None:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = src
        if multi_devs is None and not args.no_cold and not args.self_:
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
        if multi_devs is None and not args.self_ and args.batch > 1:
            try:
                out["batch"] = batch_leg(D, ra, rb, shared, threads, pair_gbp, args.batch, max(2, min(args.steps, 5)), ms_per_step)
            except Exception as e:                # never takes the bench line down
                out["batch"] = {"error": str(e)}
        if multi_devs is None and not args.self_ and not args.no_human_scale:
            ses.close()                           # the 3 Gbp leg wants the whole device
            ses = None
            for key, div in (("human_scale", 0.01), ("human_scale_10pct", 0.10)):
                try:
                    out[key] = human_scale_run(D, workload, shared, threads, div, project=(div == 0.01))
                except Exception as e:            # never takes the bench line down
                    out[key] = {"error": str(e)}
            if os.environ.get("FGA_BENCH_SELF_1G", "1") != "0":
                try:
                    out["self_scale"] = self_scale_run(D, workload, shared, threads)
                except Exception as e:
                    out["self_scale"] = {"error": str(e)}
    # N > 1: parity of the step that was timed (its .1aln against the same comparison on one GPU, record lines in sequence),
    # the cold form of the same entry (fga_run_multi = open + run + close), and the comparison north_star names -- ONE 3 Gbp x
    # 3 Gbp pair over the N GPUs (strong scaling) beside the weak-scaled `value` above.  All on rank 0; the others wait.
    if multi_devs is not None and rank == 0:
        if os.environ.get("FGA_BENCH_VERIFY", "1") != "0":
            try:
                ses.close()
                ses = None
                got = file_digest(workload, out1aln)
                out["parity"] = verify_multi_step(D, workload, kw, ra, rb, got, shared, min(32, threads))
                out["c_abi_multi_cold"] = multi_cold_leg(D, workload, ra, rb, shared, threads, multi_devs, mbp * 1e-3, min(32, threads), compare_with=got)
            except Exception as e:                    # never takes the bench line down
                out["parity"] = {"error": str(e)}
        want3g = os.environ.get("FGA_BENCH_SHARDED_3G", "1")          # "force": also over a --devices list (a code-path check)
        if (world > 1 or want3g == "force") and not args.no_human_scale and args.strong_mbp <= 0 and want3g != "0":
            try:
                if ses is not None:
                    ses.close()
                    ses = None
                out["human_scale_sharded"] = multi_human_scale(D, workload, shared, threads, multi_devs)
            except Exception as e:                    # never takes the bench line down
                out["human_scale_sharded"] = {"error": str(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)

    if ses is not None:
        ses.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def batch_leg(D, ra, rb, workdir, threads, pair_gbp, k, steps, single_ms):
    """K independent comparisons of the bench pair in flight on ONE GPU: K host threads, each with a session of its own (its
    own HIP stream, buffers from the shared device pool), each running `steps` comparisons back to back.  One comparison
    alone keeps ~130 of the device's 4,096 wavefront slots busy (the extension is one wavefront's serial chain);
    independent comparisons fill the rest -- what the C-ABI's re-entrancy (include/fastga_amd.h) is for, and what the
    reference's file-static tables and thread team cannot do (RSDsort.c:26-33).  ctypes releases the GIL during the calls."""
    import threading
    sessions = [D.Session(ra, rb, nthreads=max(1, threads // k)) for _ in range(k)]
    outs = [os.path.join(workdir, f"batch{i}.1aln") for i in range(k)]
    errs, recs = [], [None] * k

    def work(i, n):
        try:
            for _ in range(n):
                st = sessions[i].run(out_path=outs[i], nthreads=max(1, threads // k), command_line="bench.py FastGA batch",
                                     reference_threads=threads)
            recs[i] = int(st["nlive"])
        except Exception as e:                           # noqa: BLE001
            errs.append(repr(e))

    def run_all(n):
        th = [threading.Thread(target=work, args=(i, n)) for i in range(k)]
        t = time.time()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return time.time() - t

    run_all(1)                                           # every session's buffers in place
    dt = run_all(steps)
    for s in sessions:
        s.close()
    res = {"comparisons_in_flight": k, "steps_each": steps, "seconds": round(dt, 3),
           "value": k * steps * pair_gbp / dt, "unit": "Gbp-pair/s",
           "ms_per_comparison_amortised": round(1000.0 * dt / (k * steps), 2),
           "vs_one_at_a_time": (single_ms / 1000.0) * k * steps / dt,
           "records_each": recs, "what": f"{k} host threads x own session on cuda:0, the bench pair each, .1aln written by each"}
    if errs:
        res["error"] = errs[0]
    return res


def file_digest(workload, path, golden_keys=False):
    """digest of a .1aln for the parity checks of the N > 1 legs: ONEview's text (comparable with tests/golden/*_digest.json,
    `lines_md5` = the record lines in sequence) when the reference's viewer travelled, this library's own reader otherwise"""
    from oracle import harness as H
    if os.path.exists(H.ref_bin("ONEview")):
        return workload.digest_1aln_stream(path, H.ref_bin("ONEview"))
    if golden_keys:
        return None
    return workload.digest_1aln_records(path)


def verify_multi_step(D, workload, ses_kw, ra, rb, got, workdir, threads):
    """Parity of the N-GPU step: the SAME comparison once more on GPU 0 alone (whole tables, fga_session_run) must give the
    file the step wrote, record line for record line.  (The one-GPU path is what the test suite pins to the reference; at
    N x 100 Mbp no golden digest exists.)"""
    one = os.path.join(workdir, "one_gpu.1aln")
    ses = D.Session(ra, rb, device=0, nthreads=threads)
    kw = dict(ses_kw); kw["out_path"] = one; kw["nthreads"] = threads
    t = time.time()
    st = ses.run(**kw)
    dt = time.time() - t
    ses.close()
    exp = file_digest(workload, one)
    return {"multi_equals_one_gpu_run": bool(got == exp), "records": int(st["nlive"]),
            "one_gpu_seconds_first_run": round(dt, 3),
            "digest": {k: got[k] for k in got if k in ("records", "lines_md5", "order_md5", "fields_md5", "trace_md5")},
            "how": "the step's .1aln vs fga_session_run of the same pair on one GPU: record lines in sequence"}


def multi_cold_leg(D, workload, ra, rb, workdir, threads, devs, gbp, ref_threads, compare_with=None, golden=None):
    """The same comparison through fga_run_multi -- what `FastGA -G<N>` runs from the command line: open + run + close in one
    call, so the span includes reading the GDBs, the genomes to every device and every rank's slice of both index builds."""
    out = os.path.join(workdir, "c_abi_multi.1aln")
    t = time.time()
    st = D.run_multi(ra, rb, out, devices=devs, nthreads=threads, reference_threads=ref_threads,
                     command_line="bench.py FastGA -G%d" % len(devs))
    dt = time.time() - t
    res = {"entry": "fga_run_multi (one process, devices %s, hipMemcpyPeerAsync exchange)" % (list(devs),),
           "seconds_cold": round(dt, 3), "value_cold": gbp / dt, "unit": "Gbp-pair/s", "records": int(st["nlive"]),
           "stage_s_max_over_ranks": {k: round(st[k], 3) for k in ("merge_s", "sort_s", "chain_s", "extend_s", "filter_s", "write_s")},
           "open_s": round(st["upload_s"], 3)}
    got = file_digest(workload, out, golden_keys=golden is not None)
    if golden is not None and got is not None:
        res["digest_equals_reference"] = all(got[k] == golden[k] for k in ("records", "header_md5", "records_sum128", "order_md5", "lines_md5"))
    elif compare_with is not None:
        res["equals_timed_step"] = bool(got == compare_with)
    os.unlink(out)
    return res


def multi_human_scale(D, workload, workdir, threads, devs):
    """BASELINE configs[3] over the N GPUs of the node: 3 Gbp x 3 Gbp, 1 %, ONE comparison at a time on a fga_multi session
    (every device its slice of both indices, built on it).  The cold span is the open + the first run; three runs follow
    on the warm session -- from the second on the contigs are dealt to the devices by the wave steps of the run before."""
    import shutil
    d = os.path.join(workdir, "human_scale_sharded")
    gbp = float(os.environ.get("FGA_BENCH_SHARDED_MBP", "3000")) * 1e-3        # 3 Gbp unless a code-path check asks for less
    os.makedirs(d, exist_ok=True)
    try:
        t = time.time()
        ra, rb = workload.build_config4(d, mbp=1000.0 * gbp, divergence=0.01, threads=min(32, threads))
        prep = time.time() - t
        t = time.time()
        M = D.Multi(ra, rb, devices=devs, nthreads=threads)
        opened = time.time() - t
        out = os.path.join(d, "sharded.1aln")
        kw = dict(out_path=out, nthreads=threads, command_line="bench.py FastGA 3 Gbp over %d GPUs" % len(devs), reference_threads=32)
        times, ranks, last = [], [], None
        for _ in range(4):
            t = time.time()
            last = M.run(**kw)
            times.append(time.time() - t)
            ranks.append(M.rank_stats())
        M.close()
        dt = min(times[1:])
        ext = [[r["extend_kernel_ms"] for r in rs] for rs in ranks]
        res = {"workload": f"synthetic {gbp:g} Gbp vs {gbp:g} Gbp, 1% divergence, 32 contigs, 45% repeats (BASELINE configs[3]): "
                           f"ONE comparison over {len(devs)} GPUs (fga_multi_run: prefix ranges -> seeds by A-contig part over "
                           f"xGMI -> phase 2 + filter by part -> one .1aln)",
               "n_gpus": len(devs), "devices": list(devs), "scaling": "strong", "value": gbp / dt, "unit": "Gbp-pair/s",
               "seconds": round(dt, 3), "seconds_runs": [round(x, 3) for x in times], "records": int(last["nlive"]),
               "open_s": round(opened, 2), "genomes_s": round(prep, 1),
               "per_rank_last_run": {k: [round(r[k], 3) for r in ranks[-1]] for k in ("phase1_s", "exchange_s", "phase2_s", "extend_kernel_ms")},
               "part_imbalance_extend_by_run": [round(max(e) / (sum(e) / len(e)), 3) if sum(e) > 0 else None for e in ext],
               "cold": {"seconds": round(opened + times[0], 2),
                        "span": "GDBs on disk -> every device's slice of both indices built on it -> the first comparison -> "
                                ".1aln closed"}}
        gold = os.path.join(ROOT, "tests", "golden", "config4_3000m_digest.json")
        if os.path.exists(gold) and abs(gbp - 3.0) < 1e-9:
            g = json.load(open(gold))
            res["records_equal_reference"] = bool(last["nlive"] == g["records"])
            # the parity gate of the leg: header, records as a multiset, (aread, abpos) order and the record LINES in
            # sequence against the digest the real reference's file gave (tests/golden/make_golden_config4.py)
            try:
                got = file_digest(workload, out, golden_keys=True)
                if got is not None:
                    res["digest_equals_reference"] = all(got[k] == g[k] for k in
                                                          ("records", "header_md5", "records_sum128", "order_md5", "lines_md5"))
                    res["lines_md5"] = got["lines_md5"]
                else:
                    res["digest_equals_reference"] = None      # oracle/_ref/ONEview did not travel
            except Exception as e:
                res["digest_error"] = str(e)
            if g.get("reference_seconds"):
                res["reference_seconds"] = g["reference_seconds"]
                res["vs_reference_warm"] = g["reference_seconds"] / dt
                res["cold"]["vs_reference"] = g["reference_seconds"] / (opened + times[0])
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


# What a wave step of the throughput extension costs a SIMD's vector ALU (rocprofv3 PMC of round 6 over the 150 Mbp self
# comparison, profiles/r06_throughput_pmc_summary.csv, 5.65e8 wave steps per launch): 128 VALU wave-instructions per step (round 5:
# 224) that keep the vector ALU busy for SQ_ACTIVE_INST_VALU = 134.6 quad-cycles = 538 cycles (4.2 per instruction: selects, compares,
# shifts, lane moves and anything with an SGPR operand take four cycles, plain two-operand integer ops two --
# profiles/r05_valu_issue_rates.txt); beside them 138 SALU, 12 LDS, 2 SMEM and 25 branch instructions.  The chip has 1024 SIMDs at 2.4 GHz
EXT_VALU_PER_STEP = 128.0
EXT_VALU_CYCLES_PER_STEP = 538.0
SIMD_CYCLES_PER_S = 1024 * 2.4e9
XGMI_LINK_GBS = 153.0          # one xGMI link (point to point, 7 per GPU): MI355X_MICROARCH.md


def sort_block(keys, passes, kernel_ms):
    """2 x 16 B x keys x passes (every pass reads and writes every 128-bit record once) over the sort's HIP-event time"""
    if not keys or not passes or not kernel_ms:
        return None
    b = 2.0 * 16.0 * keys * passes
    return {"kernel": "os_pass_kernel x passes (+ first histogram)", "bound": "hbm", "keys": int(keys), "passes": int(passes),
            "kernel_ms": round(kernel_ms, 3), "algorithmic_bytes": int(b), "achieved": b / (kernel_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def extend_block(st):
    k = st["extend_kernel_ms"]
    if not k:
        return None
    return {"kernel": "ext_mid::extend_kernel (throughput regime: twenty wavefronts per CU, issue-bound)", "bound": "valu + scalar issue",
            "kernel_ms": round(k, 1), "wave_steps": int(st["nwaves"]), "wave_steps_per_s": st["nwaves"] / (k * 1e-3),
            "avg_wave_width": st["ext_cells"] / max(1, st["nwaves"]),
            "valu_issue_frac_est": st["nwaves"] * EXT_VALU_CYCLES_PER_STEP / (k * 1e-3) / SIMD_CYCLES_PER_S,
            "note": f"{EXT_VALU_PER_STEP:g} VALU wave-instructions per step (PMC) = {EXT_VALU_CYCLES_PER_STEP:.0f} cycles of a SIMD's vector "
                    f"ALU x steps / kernel time / {SIMD_CYCLES_PER_S:.3g} SIMD cycles per s at the nominal 2.4 GHz; PMC: SQ_ACTIVE_INST_VALU = "
                    f"16 % of the wave cycles with five wavefronts per SIMD = 0.8 busy, the scalar unit ~ 0.7 "
                    f"(profiles/r06_throughput_pmc_summary.csv)"}


def project_8gpu(st8, nparts):
    """wall time of the comparison on `nparts` GPUs, put together from its run as nparts prefix ranges x nparts parts on ONE
    GPU (parallel.run_parts_on_one_gpu): slowest phase 1 + the all-to-all-v at one xGMI link per directed pair + slowest
    phase 2 (+ its filter) + the gather of the surviving records + the merge by A contig and the write on rank 0"""
    pr = st8["per_rank"]
    # (what a rank of the emulation waited for the driver is not part of a real rank's phase 1: the emulation holds all
    # ranks' buffers on one device)
    drv = pr.get("merge_driver_alloc_s", [0.0] * nparts)
    phase1 = [m - w + s for m, w, s in zip(pr["merge_s"], drv, pr["split_s"])]
    sent = st8["sent_bytes"]
    link = max(max(sent[r][p] for p in range(nparts) if p != r) for r in range(nparts)) if nparts > 1 else 0
    phase2 = [a + f for a, f in zip(pr["align_s"], pr["filter_s"])]
    gather = sum(st8["gather_bytes"][1:]) / (XGMI_LINK_GBS * 1e9)        # into rank 0: its links in parallel at best, one at worst
    ext = pr["extend_kernel_ms"]
    seconds = max(phase1) + link / (XGMI_LINK_GBS * 1e9) + max(phase2) + gather + st8["finish_s"]
    tail = None
    if st8.get("streamed"):
        # fga_multi_run's streamed finish: rank p orders its own records when its filter is through and appends them when the
        # ranks before it have (all ranks are threads of one process: no gather); what follows the last append is the footer
        done = 0.0
        for p in range(nparts):
            ready = max(phase1) + link / (XGMI_LINK_GBS * 1e9) + phase2[p] + pr["order_s"][p] + pr["format_s"][p]
            done = max(ready, done) + pr["append_s"][p]
        tail = done - (max(phase1) + link / (XGMI_LINK_GBS * 1e9) + max(phase2))
        seconds = done + st8["finish_s"]
        gather = 0.0
    return {"seconds": seconds,
            "streamed_finish": bool(st8.get("streamed")), "appends_after_slowest_rank_s": None if tail is None else round(tail, 3),
            "order_s": [round(x, 3) for x in pr.get("order_s", [])], "format_s": [round(x, 3) for x in pr.get("format_s", [])], "append_s": [round(x, 3) for x in pr.get("append_s", [])],
            "is": "a PROJECTION from 8 prefix ranges x 8 parts run one after the other on this GPU (index builds excluded: "
                  "each rank builds 1/8 of both tables); not a measurement on 8 GPUs",
            "phase1_s_max": round(max(phase1), 3), "phase1_s": [round(x, 3) for x in phase1],
            "phase1_driver_alloc_s_excluded": [round(x, 3) for x in drv],
            "exchange_s": round(link / (XGMI_LINK_GBS * 1e9), 4), "largest_directed_pair_bytes": int(link),
            "phase2_s_max": round(max(phase2), 3), "phase2_s": [round(x, 3) for x in phase2],
            "gather_s": round(gather, 4), "finish_s": round(st8["finish_s"], 3),
            "extend_kernel_ms": [round(x, 1) for x in ext], "wave_steps": pr["wave_steps"],
            "part_imbalance_extend": max(ext) / (sum(ext) / len(ext)) if sum(ext) > 0 else None,
            "part_imbalance_seeds": max(st8["part_seed_counts"]) / (sum(st8["part_seed_counts"]) / nparts)}


def self_scale_run(D, workload, workdir, threads, mbp=1000.0):
    """BASELINE configs[2]'s shape at its stated size: a repeat-heavy 1 Gbp genome (30 % repeats, 2 % tandem arrays, 85 % of the
    repeats soft-masked) against ITSELF with -M, index built on the device; the warm comparison of a session, the seed merge's
    roofline block in self mode (every table entry once in its on-disk width; every seed once in the reference's record width --
    a self run reports half of the seeds it writes, FastGA.c:1906)."""
    d = os.path.join(workdir, "self_scale")
    os.makedirs(d, exist_ok=True)
    t = time.time()
    ra = workload.build_config3(d, mbp=mbp, threads=threads, gix=False, name="A")
    prep = time.time() - t
    t = time.time()
    ses = D.Session(ra, None, nthreads=threads)
    opened = time.time() - t
    try:
        kw = dict(out_path=os.path.join(d, "self.1aln"), nthreads=threads, command_line="bench.py FastGA self", soft_mask=True,
                  reference_threads=min(32, threads))
        t = time.time()
        st1 = ses.run(**kw)
        first = time.time() - t
        t = time.time()
        st = ses.run(**kw)
        dt = time.time() - t
        alg = ses.table_bytes + 2 * st["nseeds"] * ses.seed_bytes
        ach = alg / (st["merge_kernel_ms"] * 1e-3) / 1e9
        res = {"workload": f"synthetic repeat-heavy {mbp:g} Mbp genome against itself, soft mask on (BASELINE configs[2]), 1 GPU",
               "value": mbp * 1e-3 / dt, "unit": "Gbp-pair/s", "seconds": round(dt, 2), "first_run_seconds": round(first, 2),
               "parts": int(st["nparts"]), "seeds": int(st["nseeds"]), "hits": int(st["nhits"]), "alignments": int(st["nalns"]),
               "records": int(st["nlive"]),
               "stage_s": {k: round(st[k], 2) for k in ("merge_s", "sort_s", "chain_s", "extend_s", "filter_s", "write_s")},
               "kernel_ms": {"merge": round(st["merge_kernel_ms"], 2), "sort": round(st["sort_kernel_ms"], 1),
                             "chain": round(st["chain_kernel_ms"], 1), "extend": round(st["extend_kernel_ms"], 1)},
               "roofline": {"kernel": "seed merge launch, self mode", "bound": "hbm", "algorithmic_bytes": int(alg),
                            "stored_bytes": int(ses.table_bytes + 2 * st["nseeds"] * 16),
                            "kernel_ms": st["merge_kernel_ms"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS,
                            "note": "the kernel stores 16-byte seeds (stored_bytes); the reference's record is seed_bytes wide"},
               "sort": sort_block(st["sort_keys"], st["sort_passes"], st["sort_kernel_ms"]),
               "extend": extend_block(st),
               "hbm_peak_gib": round(st["hbm_peak_bytes"] / 2**30, 1),
               "genome_s": round(prep, 1), "upload_and_index_build_s": round(opened, 2)}
        if st1["nlive"] != st["nlive"]:
            res["error"] = "the two comparisons of the session differ"
        return res
    finally:
        ses.close()


def human_scale_run(D, workload, workdir, threads, div=0.01, project=False):
    """BASELINE configs[3] (1 %) / configs[4] (10 %) at their stated size on this one GPU: 3 Gbp x 3 Gbp (32 contigs of ~94
    Mbp, 45 % repeats; fastga_amd.workload.build_config4), both indices built on the device (2.4 G entries each), ONE
    comparison from resident inputs to the .1aln closed, and the COLD span beside it (GDB files on disk -> genomes to HBM ->
    both indices built -> the comparison -> .1aln closed: the span of the reference's "Total Resources" line, which its
    golden wall time covers).  The counts are the ones the reference gives for this pair (tests/golden/config4|5_3000m_digest
    .json: made with oracle/_ref/GIXmake + FastGA -T32 on the GPU box's 256 host cores; FGA_BENCH_REF_3G=1 runs the reference
    again here; the full record digest is compared in tests/test_full_size_gpu.py)."""
    import shutil
    name = "config4" if div < 0.05 else "config5"
    d = os.path.join(workdir, "human_scale_" + name)
    os.makedirs(d, exist_ok=True)
    try:
        t = time.time()
        ra, rb = workload.build_config4(d, mbp=3000.0, divergence=div, threads=threads)
        prep = time.time() - t
        from fastga_amd.lib import load_library
        L = load_library()
        w0 = L.fga_dev_driver_seconds()
        t = time.time()
        ses = D.Session(ra, rb, nthreads=threads)
        opened = time.time() - t
        w1 = L.fga_dev_driver_seconds()
        out = os.path.join(d, "c4.1aln")
        kw = dict(out_path=out, nthreads=threads, command_line="bench.py FastGA 3 Gbp", reference_threads=32)
        t = time.time()
        st1 = ses.run(**kw)                       # the first comparison of the session also takes its work buffers from the driver
        first = time.time() - t
        w2 = L.fga_dev_driver_seconds()
        t = time.time()
        st = ses.run(**kw)                        # `value`: ONE comparison from resident inputs, like a step of the 100 Mbp bench
        dt = time.time() - t
        w3 = L.fga_dev_driver_seconds()
        alg = ses.table_bytes + st["nseeds"] * ses.seed_bytes
        res = {"workload": f"synthetic 3 Gbp vs 3 Gbp, {div*100:g}% divergence, 32 contigs, 45% repeats (BASELINE "
                           f"configs[{3 if div < 0.05 else 4}]), 1 GPU",
               "value": 3.0 / dt, "unit": "Gbp-pair/s", "seconds": round(dt, 2), "parts": int(st["nparts"]),
               "seeds": int(st["nseeds"]), "hits": int(st["nhits"]), "alignments": int(st["nalns"]), "records": int(st["nlive"]),
               "stage_s": {k: round(st[k], 2) for k in ("merge_s", "sort_s", "chain_s", "extend_s", "filter_s", "write_s")},
               "kernel_ms": {"merge": round(st["merge_kernel_ms"], 1), "sort": round(st["sort_kernel_ms"], 1),
                             "chain": round(st["chain_kernel_ms"], 1), "extend": round(st["extend_kernel_ms"], 1)},
               "roofline": {"kernel": "seed merge launch", "bound": "hbm", "algorithmic_bytes": int(alg),
                            "kernel_ms": st["merge_kernel_ms"], "achieved": alg / (st["merge_kernel_ms"] * 1e-3) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / (st["merge_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
               "sort": sort_block(st["sort_keys"], st["sort_passes"], st["sort_kernel_ms"]),
               "extend": extend_block(st),
               "hbm_peak_gib": round(st["hbm_peak_bytes"] / 2**30, 1),
               "genomes_s": round(prep, 1), "upload_and_2_index_builds_s": round(opened, 2),
               "first_run_seconds": round(first, 2),
               # seconds inside hipMalloc / hipFree (fga_dev_driver_seconds): 0.3 ms a call on a clean device, ~1 s per 40 GB
               # while the driver is still clearing memory a process before this one released
               "driver_alloc_s": {"open": round(w1 - w0, 2), "first_run": round(w2 - w1, 2), "run": round(w3 - w2, 2)},
               "cold": {"seconds": round(opened + first, 2), "value": 3.0 / (opened + first), "unit": "Gbp-pair/s",
                        "span": "GDB on disk (page cache) -> genomes to HBM -> 2 index builds on the device -> the "
                                "first comparison of the session -> .1aln closed",
                        "of_which_driver_alloc_s": round(w2 - w0, 2)}}
        if st1["nlive"] != st["nlive"]:
            res["error"] = "the two comparisons of the session differ"
        gold = os.path.join(ROOT, "tests", "golden", f"{name}_3000m_digest.json")
        if os.path.exists(gold):
            g = json.load(open(gold))
            res["reference"] = {"seconds": g.get("reference_seconds"), "threads": g.get("reference_threads"),
                                "where": f"same box class (256 host cores), tests/golden/{name}_3000m_digest.json; its own "
                                         f"GIXmake -T32 on both genomes comes on top (not in the span)"}
            res["counts_equal_reference"] = (st["nseeds"] == g["total_seeds"] and st["nhits"] == g["hits"] and
                                             st["nalns"] == g["alignments"] and st["nlive"] == g["records"])
            if g.get("reference_seconds"):
                res["cold"]["vs_reference"] = g["reference_seconds"] / (opened + first)
                res["vs_reference_warm"] = g["reference_seconds"] / dt
        if project:
            from fastga_amd import parallel
            out8 = os.path.join(d, "parts8.1aln")
            # twice, like two runs of a fga_multi session: the first deals the contigs to the parts by seed counts, the second
            # by the wave steps the first one counted per A contig (fga_alns.ctg_waves) -- the warm comparison is the second
            st8a = parallel.run_parts_on_one_gpu(ses, 8, out_path=out8, nthreads=threads, reference_threads=32)
            st8 = parallel.run_parts_on_one_gpu(ses, 8, weights=st8a["contig_waves"], out_path=out8, nthreads=threads,
                                                reference_threads=32)
            res["projected_8gpu"] = project_8gpu(st8, 8)
            res["projected_8gpu"]["records_equal_one_gpu_run"] = bool(st8["nlive"] == st["nlive"] and st8a["nlive"] == st["nlive"])
            proj1 = project_8gpu(st8a, 8)
            res["projected_8gpu"]["first_run_of_a_session"] = {k: proj1[k] for k in ("seconds", "phase2_s_max", "part_imbalance_extend",
                                                                                      "part_imbalance_seeds")}
        ses.close()
        # The denominator of the 3 Gbp ratios measured in THIS run, on this box's host cores: the real reference's GIXmake -T32 on
        # both genomes (not in the span, like its own "Total Resources" line) and FastGA -T32 on the pair.  By default for
        # configs[3] where the box has the cores (>= 64; four to six minutes); configs[4] keeps the stored figure, labelled.
        # FGA_BENCH_REF_3G=0 skips it, =1 forces it for both.
        want = os.environ.get("FGA_BENCH_REF_3G", "")
        if want != "0" and (want == "1" or (div < 0.05 and (os.cpu_count() or 1) >= 64)):
            from oracle import harness as H
            if H.have_reference():
                rd = None
                try:
                    # the reference's index files (2 x 34 GB) and seed files (15 GB) want ~100 GB of scratch: /dev/shm where it has
                    # the room (the GPU boxes: 1.5 TB of it, 74 GB of /tmp), else beside the genomes
                    import tempfile
                    rd = d
                    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > 200e9:
                        rd = tempfile.mkdtemp(prefix="fga_ref3g_", dir="/dev/shm")
                        for r in (ra, rb):
                            b = os.path.basename(r)
                            for f in os.listdir(d):
                                if f in (b + ".gdb", b + ".1gdb", "." + b + ".bps"):
                                    shutil.copy(os.path.join(d, f), os.path.join(rd, f))
                    elif shutil.disk_usage(d).free < 130e9:
                        raise RuntimeError(f"skipped: {shutil.disk_usage(d).free/1e9:.0f} GB of scratch, the reference needs ~130")
                    rra, rrb = os.path.join(rd, os.path.basename(ra)), os.path.join(rd, os.path.basename(rb))
                    t = time.time()
                    for r in (rra, rrb):
                        H.run([H.ref_bin("GIXmake"), "-T32", f"-P{rd}", r], cwd=rd)
                    gt = time.time() - t
                    t = time.time()
                    H.ref_fastga(rra, rrb, rd, os.path.join(rd, "ref"), threads=32)
                    ft = time.time() - t
                    res["reference"] = {"seconds": round(ft, 1), "threads": 32, "measured_in_this_run": True, "gixmake_s": round(gt, 1),
                                        "where": f"oracle/_ref/FastGA -T32 on this box ({os.cpu_count()} host cores), index files by "
                                                 f"oracle/_ref/GIXmake -T32 (not in the span)"}
                    res["vs_reference_warm"] = ft / dt
                    res["cold"]["vs_reference"] = ft / (opened + first)
                    refd = os.path.join(rd, "ref.1aln")
                    if os.path.exists(refd) and os.path.exists(H.ref_bin("ONEview")):
                        a = workload.digest_1aln_stream(out, H.ref_bin("ONEview"))
                        b = workload.digest_1aln_stream(refd, H.ref_bin("ONEview"))
                        res["identical_to_reference_here"] = bool(a["lines_md5"] == b["lines_md5"] and a["records"] == b["records"])
                except Exception as e:
                    res["reference_error"] = str(e)[-400:]
                finally:
                    if rd is not None and rd != d:
                        shutil.rmtree(rd, ignore_errors=True)
        if "reference" in res and not res["reference"].get("measured_in_this_run"):
            res["reference"]["measured_in_this_run"] = False
        return res
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cold_run(D, ra, rb, workdir, threads, pair_gbp):
    """One comparison from files on disk to the .1aln closed -- the span of the reference's "Total Resources" line
    (FastGA.c:4828-4829, 5263-5264), which starts with GDB (+ GIX) present: here the two GDBs are read, the bases go to
    HBM, both indices are BUILT on the device (no .gix files exist yet), one step runs, the .1aln is written and
    everything is released again (fga_run).  The files are in the page cache, as they are for the reference leg."""
    from fastga_amd.lib import load_library
    L = load_library()
    out = os.path.join(workdir, "cold.1aln")
    best = None
    for _ in range(2):                  # the second run no longer pays the one-off HIP module / allocator warm-up
        w = L.fga_dev_driver_seconds()
        t = time.time()
        st = D.run(ra, rb, out, nthreads=threads, command_line="bench.py FastGA cold")
        dt = time.time() - t
        if best is None or dt < best[0]:
            best = (dt, st, L.fga_dev_driver_seconds() - w)
    dt, st, wait = best
    return {"value": pair_gbp / dt, "unit": "Gbp-pair/s", "ms": round(1000 * dt, 1),
            "of_which_driver_alloc_ms": round(1000 * wait, 1),        # inside hipMalloc / hipFree (see human_scale.driver_alloc_s)
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
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv"))
                   if re.fullmatch(r"r\d+_pmc_summary\.csv", os.path.basename(f)))     # the bench pair's, not another shape's
    if not files:
        return None, None
    fetch = write = None
    commit = ""
    rows = []
    for row in csv.reader(ln for ln in open(files[-1]) if not ln.startswith("#")):
        if len(row) < 5 or row[0] == "pass":
            continue
        # pass, kernel, counter, launches, avg_per_launch -- a template kernel's name holds commas of its own
        rows.append((",".join(row[1:-3]), row[-3], float(row[-2]), float(row[-1])))
    # per MERGE launch: the range cuts are made once per session since round 5 (fewer launches than the walk kernel)
    walks = max([n for k, c, n, a in rows if "seed_merge" in k and c == "FETCH_SIZE"] or [1.0])
    for k, counter, n, avg in rows:
        if "seed_merge" in k or "range_cut" in k:
            share = avg * (min(n, walks) / walks if "range_cut" in k else 1.0)
            if counter == "FETCH_SIZE":
                fetch = (fetch or 0.0) + share
            elif counter == "WRITE_SIZE":
                write = (write or 0.0) + share
    for ln in open(files[-1]):
        if ln.startswith("# commit"):
            commit = ln.split(":", 1)[1].strip()
    if fetch is None or write is None:
        return None, None
    src = os.path.relpath(files[-1], ROOT) + (f" @ {commit}" if commit else "")
    return int((2.0 * fetch + write) * 1024), src


if __name__ == "__main__":
    main()
