"""One comparison over several GPUs (one process per GPU, torch.distributed: "nccl" = RCCL over xGMI; "gloo" on CPU).

The cut follows the reference's own two partitions:

  phase 1  every rank merges ONE 12-mer prefix range of the two tables (equal merge cost per range; the reference splits
           its merge threads the same way, FastGA.c:2291-2321).  Panels are independent: the union of the ranges' seeds
           is the seed set.
  exchange seeds per A contig are counted (the reference's buck[]), all-reduced, and every rank derives the SAME map
           A contig -> part from the totals (fga_partition_contigs: heaviest contig to the lightest part).  Each rank
           regroups its seeds by part on the device (fga_seeds_split_to, straight into the send buffer) and ONE
           all-to-all-v moves the 16-byte records: what the N_Units / C_Units file matrix and its transpose do in the
           reference (FastGA.c:5097-5134, 5160-5184).
  phase 2  every rank sorts / chain-scans / extends the seeds of its part: contig pairs are independent work units.
  filter   every rank runs the redundancy filter on ITS records (all records of a contig pair come from the part owning
           the A contig, so the filter needs nothing from other ranks) and orders them.
  gather   the surviving alignments (56-byte records + trace bytes, tens of MB) go to rank 0, which lays the ranks' runs
           per A contig out in contig order and writes the .1aln once: the reference's la_merge (FastGA.c:3991-4133).

Everything on the data path is the C-ABI of include/fastga_amd.h; torch provides the exchange buffers and the
collectives, nothing else.  `run_parts_on_one_gpu` drives the same C-ABI calls with the transport replaced by local
slicing, so that results can be checked for any number of parts on a single GPU.
"""
import ctypes as C

import numpy as np

NPREFIX = 1 << 24


# ------------------------------------------------------------------------------------------------ host-side pieces

def prefix_shards(idx1, idx2, nshards):
    """[(begin,end)] * nshards covering [0, 2^24), equal cumulative entry count of both tables per shard (host copy of
    the prefix indices; fga_session_prefix_cuts does the same on the device-resident indices)."""
    total = int(idx1[-1]) + (int(idx2[-1]) if idx2 is not None else 0)
    cuts = [0]
    for s in range(1, nshards):
        target = (total * s) // nshards
        if idx2 is None:
            p = int(np.searchsorted(idx1, target, side="left"))
        else:
            lo, hi = 0, NPREFIX
            while lo < hi:
                m = (lo + hi) >> 1
                if int(idx1[m]) + int(idx2[m]) >= target:
                    hi = m
                else:
                    lo = m + 1
            p = lo
        cuts.append(max(cuts[-1], min(NPREFIX, p + 1)))
    cuts.append(NPREFIX)
    return [(cuts[i], cuts[i + 1]) for i in range(nshards)]


def partition_contigs(weights, nparts):
    """fga_partition_contigs: part of every A contig from its weight (seed count); same result on every rank"""
    from .lib import load_library, check
    L = load_library()
    w = np.ascontiguousarray(weights, dtype=np.int64)
    sel = np.zeros(len(w), dtype=np.int32)
    check(L.fga_partition_contigs(w.ctypes.data_as(C.POINTER(C.c_int64)), len(w), nparts,
                                  sel.ctypes.data_as(C.POINTER(C.c_int))), "partition")
    return sel


def partition_contigs_in_order(weights, perm, nparts):
    """fga_partition_contigs_in_order: the A contigs dealt out contiguously in their ORIGINAL order (perm[j] = original
    index of contig j of the index order), stretches of about equal weight -- what fga_session_run deals when it streams the
    parts' records to the .1aln"""
    from .lib import load_library, check
    L = load_library()
    w = np.ascontiguousarray(weights, dtype=np.int64)
    pm = np.ascontiguousarray(perm, dtype=np.int32)
    sel = np.zeros(len(w), dtype=np.int32)
    check(L.fga_partition_contigs_in_order(w.ctypes.data_as(C.POINTER(C.c_int64)), pm.ctypes.data_as(C.POINTER(C.c_int)),
                                           len(w), nparts, sel.ctypes.data_as(C.POINTER(C.c_int))), "partition in order")
    return sel


def alns_to_arrays(ptr):
    """POINTER(Alns) -> (records as ALN_DTYPE array, trace bytes, (ncalls, nwaves)); copies"""
    from .device import ALN_DTYPE
    a = ptr.contents
    n, nt = a.naln, a.ntrace
    recs = np.frombuffer((C.c_char * (n * ALN_DTYPE.itemsize)).from_address(a.alns), dtype=ALN_DTYPE).copy() \
        if n > 0 else np.zeros(0, dtype=ALN_DTYPE)
    tb = np.frombuffer((C.c_char * nt).from_address(a.tbytes), dtype=np.uint8).copy() if nt > 0 \
        else np.zeros(0, dtype=np.uint8)
    return recs, tb, (a.ncalls, a.nwaves)


def arrays_to_alns(recs, tb, counts=(0, 0)):
    """(records, trace bytes) -> (Alns struct, keep-alive tuple); the struct points into the arrays"""
    from .lib import Alns
    recs = np.ascontiguousarray(recs)
    tb = np.ascontiguousarray(tb, dtype=np.uint8)
    a = Alns(len(recs), len(tb), counts[0], counts[1], recs.ctypes.data, tb.ctypes.data)
    return a, (recs, tb)


def all_reduce_counts(dist, counts, device):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64)).to(device)
    dist.all_reduce(t)
    return t.cpu().numpy()


def gather_records(dist, recs, tb, counts, device, dst=0):
    """all ranks' (records, trace bytes) on rank `dst` as a list indexed by rank (None elsewhere).  Sizes first (one
    small all-gather), then one padded gather per array: tens of MB over xGMI."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    from .device import ALN_DTYPE
    meta = torch.tensor([len(recs), len(tb), int(counts[0]), int(counts[1])], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    out = None
    for which, arr, unit in ((0, recs.view(np.uint8).reshape(-1), ALN_DTYPE.itemsize), (1, tb, 1)):
        cap = max(max(m[which] for m in metas) * unit, 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=device)
        if arr.size:
            buf[:arr.size] = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
        got = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, got, dst=dst)
        if rank == dst:
            if out is None:
                out = [[None, None, (m[2], m[3])] for m in metas]
            for r in range(world):
                raw = got[r][:metas[r][which] * unit].cpu().numpy()
                out[r][which] = raw.view(ALN_DTYPE).copy() if which == 0 else raw.copy()
    return out


# ------------------------------------------------------------------------------------------------ the sharded run

def run_sharded(ses, dist, prm_kwargs, device):
    """One pass of the hot path over the session's resident inputs, cut over dist's ranks.  Every rank holds both
    genomes' bases (phase 2 needs them) and, with a sliced session (Session(rank=, nranks=)), only its own 12-mer prefix
    range of the two tables; an unsliced session holds the whole tables and merges its range of them.  Returns the stats
    dict on every rank; rank 0's has the totals of the finished run (nlive, cover) and wrote the output.
    device = "cuda:<n>" with backend nccl (RCCL: the exchange buffers are device tensors), or "cpu" with backend gloo
    (records staged through the host; how tests/test_parts_gpu.py runs this very function with two ranks on one GPU)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    out_path = prm_kwargs.get("out_path")
    kw = dict(prm_kwargs)
    if rank != 0:
        kw["out_path"] = None
        kw["paf_path"] = None
    prm = ses.params(**kw)
    st = ses.new_stats()
    cuts = prefix_cuts(ses, world)
    ses.clear_strand_counts()
    seeds = ses.merge(prm, st, int(cuts[rank]), int(cuts[rank + 1]))
    n = seeds.count
    # the library keeps the device memory it is done with for reuse; what RCCL and torch allocate on their first
    # collective comes from the driver, so idle regions go back first (only regions that are entirely free are returned)
    ses.trim()
    hist = all_reduce_counts(dist, ses.contig_histogram(seeds), device)
    if kw.get("reference_threads", 0) > 0:        # the reference's tie order needs the per-strand counts of ALL ranges
        ses.set_strand_counts(all_reduce_counts(dist, ses.strand_counts(), device))
    select = partition_contigs(hist, world)
    host = str(device) == "cpu"          # gloo: the collectives move host tensors, the records are staged through the host
    if host:
        sbuf = ses.dev_malloc(16 * max(n, 1))
        off = ses.split_to(seeds, select, world, sbuf)
        send = torch.from_numpy(ses.dev_download(sbuf, 16 * n).view(np.int32).reshape(-1, 4).copy()) if n > 0 \
            else torch.zeros((0, 4), dtype=torch.int32)
        ses.dev_free(sbuf)
    else:
        send = _torch_empty(ses, torch, (max(n, 1), 4), device)
        off = ses.split_to(seeds, select, world, send.data_ptr())
    seeds.free()
    in_splits = np.diff(off).astype(np.int64)
    t_in = torch.from_numpy(in_splits).to(device)
    t_out = torch.zeros_like(t_in)
    dist.all_to_all_single(t_out, t_in)
    out_splits = t_out.cpu().numpy()
    total = int(out_splits.sum())
    recv = _torch_empty(ses, torch, (max(total, 1), 4), device)
    dist.all_to_all_single(recv[:total], send[:n], output_split_sizes=out_splits.tolist(),
                           input_split_sizes=in_splits.tolist())
    if host:
        rbuf = ses.dev_malloc(16 * max(total, 1))
        if total > 0:
            ses.dev_upload(rbuf, recv[:total].numpy())
        part = ses.import_seeds([(rbuf, total)])
        ses.dev_free(rbuf)
    else:
        torch.cuda.synchronize()
        part = ses.import_seeds([(recv.data_ptr(), total)])
    del send, recv
    raw = ses.align(prm, st, part)
    fil = ses.filter(raw, nthreads=kw.get("nthreads", 8))        # this rank's contig pairs are complete: filter here
    ses.free_alns(raw)
    recs, tb, counts = alns_to_arrays(fil)
    ses.free_alns(fil)
    allr = gather_records(dist, recs, tb, counts, device)
    if rank == 0:
        keep, structs = [], []
        for r in range(world):
            a, k = arrays_to_alns(allr[r][0], allr[r][1], allr[r][2])
            keep.append(k)
            structs.append(C.pointer(a))
        ses.finish_filtered(prm, st, structs)
    d = ses.stats_dict(st)
    d["exchange_seeds_out"] = int(n - in_splits[rank])
    d["part_seeds"] = total
    d["out_path"] = out_path if rank == 0 else None
    return d


def _torch_empty(ses, torch, shape, device):
    """an int32 exchange buffer from torch's allocator; when the device is full of regions the library keeps for reuse,
    those nobody uses go back to the driver first (fga_dev_trim) and the allocation is tried once more"""
    try:
        return torch.empty(shape, dtype=torch.int32, device=device)
    except RuntimeError:
        ses.trim()
        return torch.empty(shape, dtype=torch.int32, device=device)


def prefix_cuts(ses, nshards):
    cuts = np.zeros(nshards + 1, dtype=np.int64)
    from .lib import check
    check(ses.L.fga_session_prefix_cuts(ses.h, nshards, cuts.ctypes.data_as(C.POINTER(C.c_int64))), "prefix cuts")
    return cuts


def run_parts_on_one_gpu(ses, nparts, weights=None, stream=True, **prm_kwargs):
    """The sharded run's C-ABI calls on ONE GPU, rank by rank in sequence (merge of each prefix range, histogram,
    partition, split, import of each part's pieces, align, finish over all parts): the parity check of the multi-GPU
    path for any number of parts.  The staging buffers come from fga_dev_malloc (no torch in this process).
    weights: what the contigs are dealt to the parts by (per A contig, index order) -- None: the seed counts, as the first run of
    a fga_multi session does; the `contig_waves` a previous call returned: the wave steps of every contig's units, as the
    session's later runs do.
    stream: like fga_multi_run, deal the contigs in ORIGINAL order when that deal's heaviest part is within 1.25 x of the balanced
    one's, and have every "rank" put its own records into the reference's tie order and append them to the .1aln
    (fga_aln_stream_*); what is left for the finish is the footer."""
    import time
    from .lib import check
    prm = ses.params(**prm_kwargs)
    st = ses.new_stats()
    cuts = prefix_cuts(ses, nparts)
    ses.clear_strand_counts()                                 # the "ranks" share the session: their counts add up in it
    sends, offs, hist = [], [], np.zeros(ses.nctg, dtype=np.int64)
    merged = []
    per_rank = {"merge_s": [], "split_s": [], "align_s": [], "sort_s": [], "chain_s": [], "extend_s": [], "filter_s": [],
                "extend_kernel_ms": [], "wave_steps": [], "records": [], "merge_driver_alloc_s": []}
    for r in range(nparts):                                   # "rank r": phase 1 on its prefix range
        t = time.time()
        w = ses.L.fga_dev_driver_seconds()
        seeds = ses.merge(prm, st, int(cuts[r]), int(cuts[r + 1]))
        hist += ses.contig_histogram(seeds)
        per_rank["merge_s"].append(time.time() - t)
        # seconds of this rank's phase 1 spent inside hipMalloc: on ONE GPU the emulation keeps every rank's seed and send
        # buffers alive side by side, so late ranks may have to ask the driver for a fresh region (a real rank never does)
        per_rank["merge_driver_alloc_s"].append(ses.L.fga_dev_driver_seconds() - w)
        merged.append((seeds, ses.dev_malloc(16 * seeds.count)))
    wts = hist if weights is None else np.asarray(weights, dtype=np.int64)
    select = partition_contigs(wts, nparts)
    streamed = False
    if stream and prm.out_path and not prm.paf_path:
        inord = partition_contigs_in_order(wts, ses.contig_perm(), nparts)
        lmax = np.bincount(select, weights=wts, minlength=nparts).max()
        omax = np.bincount(inord, weights=wts, minlength=nparts).max()
        if omax <= lmax + lmax // 4 + 1:
            select, streamed = inord, True
    per_rank["order_s"], per_rank["format_s"], per_rank["append_s"] = [], [], []
    out_stream = ses.stream_open(prm) if streamed else None
    contig_waves = np.zeros(ses.nctg, dtype=np.int64)
    for seeds, buf in merged:
        t = time.time()
        offs.append(ses.split_to(seeds, select, nparts, buf))
        per_rank["split_s"].append(time.time() - t)
        seeds.free()
        sends.append(buf)
    raws = []
    for p in range(nparts):                                   # "rank p": its part's pieces from every rank, phase 2
        pieces = [(sends[r] + 16 * int(offs[r][p]), int(offs[r][p + 1] - offs[r][p])) for r in range(nparts)]
        before = ses.stats_dict(st)
        t = time.time()
        part = ses.import_seeds(pieces)
        raw = ses.align(prm, st, part)
        per_rank["align_s"].append(time.time() - t)
        if raw.contents.ctg_waves and raw.contents.nctg_waves == ses.nctg:     # wave steps per A contig of this part's units
            contig_waves += np.frombuffer((C.c_int64 * ses.nctg).from_address(raw.contents.ctg_waves), dtype=np.int64)
        after = ses.stats_dict(st)
        for k in ("sort_s", "chain_s", "extend_s", "extend_kernel_ms"):
            per_rank[k].append(after[k] - before[k])
        per_rank["wave_steps"].append(int(after["nwaves"] - before["nwaves"]))
        t = time.time()
        raws.append(ses.filter(raw, nthreads=prm_kwargs.get("nthreads", 8)))      # "rank p" filters its own records
        per_rank["filter_s"].append(time.time() - t)
        per_rank["records"].append(int(raws[-1].contents.naln))
        ses.free_alns(raw)
        if streamed:                                          # "rank p": the tie order of its records, its stretch of the file
            t = time.time()
            ses.reference_order(prm, raws[-1])
            per_rank["order_s"].append(time.time() - t)
            t = time.time()                                   # formatted on the rank's own threads before its turn ..
            blk = C.c_void_p()
            pre = bool(ses.L.fga_aln_stream_preformats(out_stream)) and raws[-1].contents.naln > 0
            if pre:
                check(ses.L.fga_aln_stream_format(out_stream, raws[-1], C.byref(blk)), "stream format")
            per_rank["format_s"].append(time.time() - t)
            t = time.time()                                   # .. written when the ranks before it have
            if pre:
                check(ses.L.fga_aln_stream_commit(out_stream, blk), "stream commit")
            else:
                check(ses.L.fga_aln_stream_append(out_stream, raws[-1]), "stream append")
            per_rank["append_s"].append(time.time() - t)
    for buf in sends:
        ses.dev_free(buf)
    t = time.time()
    if streamed:
        check(ses.L.fga_aln_stream_close(out_stream, 1), "stream close")
        st.nlive = sum(int(r.contents.naln) for r in raws)
    else:
        ses.finish_filtered(prm, st, raws)
    finish_s = time.time() - t
    gather_bytes = [int(r.contents.naln) * 56 + int(r.contents.ntrace) for r in raws]
    for r in raws:
        ses.free_alns(r)
    d = ses.stats_dict(st)
    d["part_seed_counts"] = [int(sum(offs[r][p + 1] - offs[r][p] for r in range(nparts))) for p in range(nparts)]
    # what an N-GPU run moves and waits for, rank by rank (bench.py projects the N-GPU wall time from these)
    d["per_rank"] = per_rank
    d["sent_bytes"] = [[16 * int(offs[r][p + 1] - offs[r][p]) for p in range(nparts)] for r in range(nparts)]   # [from][to]
    d["gather_bytes"] = gather_bytes
    d["finish_s"] = finish_s
    d["streamed"] = streamed
    d["contig_waves"] = contig_waves
    return d
