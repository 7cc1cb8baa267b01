#!/usr/bin/env python3
"""bench.py -- throughput of the FastGA seed-and-extend hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU.  A *step* is one pass of the hot path over one synthetic genome pair
already resident in HBM (GIX tables + 2-bit genomes uploaded before the timed region).  Workload = BASELINE.json
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
    ap.add_argument("--cpu-mbp", type=float, default=8.0, help="size of the bounded CPU-baseline sample, Mbp")
    ap.add_argument("--no-cpu", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, workdir):
    """Reference FastGA (oracle/_ref) on a bounded sample pair built by our own producers; falls back to the
    oracle's seed-merge port if the reference binaries did not travel."""
    from fastga_amd import workload
    from oracle import harness as H
    ncores = os.cpu_count() or 1
    threads = max(1, min(32, ncores))
    d = os.path.join(workdir, "cpu")
    os.makedirs(d, exist_ok=True)
    mbp = args.cpu_mbp
    ra, rb = workload.build_pair(d, seed=4242, ncontig=max(threads, 32), total=int(mbp * 1e6),
                                 divergence=args.div, repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02,
                                 threads=threads)
    if H.have_reference():
        t = time.time()
        H.ref_fastga(ra, rb, d, os.path.join(d, "ref"), threads=threads)
        dt = time.time() - t
        return {"value": mbp * 1e-3 / dt, "unit": "Gbp-pair/s", "cores": threads, "kind": "reference",
                "sample": f"oracle/_ref/FastGA -T{threads} on a synthetic {mbp:g} Mbp x {mbp:g} Mbp pair "
                          f"(same generator, {dt:.1f} s wall, prebuilt GDB/GIX)"}
    from fastga_amd.gixio import Gix
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    t = time.time()
    H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte)
    dt = time.time() - t
    return {"value": mbp * 1e-3 / dt, "unit": "Gbp-pair/s", "cores": 1, "kind": "port",
            "sample": f"oracle seed-merge restatement only, {mbp:g} Mbp pair, {dt:.1f} s"}


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
    from fastga_amd.gixio import Gix

    workdir = args.workdir or tempfile.mkdtemp(prefix=f"fga_bench_r{rank}_")
    os.makedirs(workdir, exist_ok=True)
    total = int(args.mbp * 1e6)
    threads = max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    ra, rb = workload.build_pair(workdir, seed=1 + rank, ncontig=args.contigs, total=total,
                                 divergence=args.div, repeat_frac=0.05, inv_frac=0.02, swap_frac=0.02,
                                 threads=threads)
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    prep_s = time.time() - t0

    dev = D.Device(local)
    dA, dB = dev.upload(A), dev.upload(B)

    def step():
        s = D.seed_merge(dev, dA, dB)
        n = s.count
        ms = dev.stage_ms(D.STAGE_MERGE)
        s.free()
        return n, ms

    def barrier():
        dev.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t = time.time()
    kms, nseeds = [], 0
    for _ in range(args.steps):
        nseeds, ms = step()
        kms.append(ms)
    barrier()
    elapsed = time.time() - t

    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = 1000.0 * elapsed / args.steps
        pair_gbp = 0.5 * (A_seqtot(ra) + A_seqtot(rb)) * 1e-9
        value = world * pair_gbp / (ms_per_step / 1000.0)
        seed_bytes = 1 + A.pbyte + B.pbyte
        alg_bytes = A.nents * A.ebytes + B.nents * B.ebytes + nseeds * seed_bytes
        kavg = sum(kms) / len(kms)
        achieved = alg_bytes / (kavg * 1e-3) / 1e9
        out = {
            "metric": "Gbp-pair aligned/sec", "value": value, "unit": "Gbp-pair/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int64",
            "data": "synthetic",
            "config": {"workload": f"synthetic {args.mbp:g} Mbp vs {args.mbp:g} Mbp, {args.div*100:g}% divergence, "
                                   f"{args.contigs} contigs, 5% repeats, 2% inversions/swaps (BASELINE configs[1])",
                       "stages": "seed-merge (GPU); sort/chain/extend not yet in the timed step",
                       "seeds": int(nseeds), "entries": [int(A.nents), int(B.nents)],
                       "prep_s": round(prep_s, 1)},
            "roofline": {"kernel": "seed_merge_kernel", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes": int(alg_bytes), "kernel_ms": kavg},
        }
        if not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args, workdir)
            except Exception as e:      # the baseline leg must never take the bench line down
                out["cpu_baseline"] = {"value": None, "unit": "Gbp-pair/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)

    dA.free(); dB.free(); dev.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def A_seqtot(root):
    from fastga_amd.gixio import Gdb
    g = Gdb(root + ".gdb")
    n = g.seqtot
    g.close()
    return n


if __name__ == "__main__":
    main()
