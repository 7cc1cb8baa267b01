"""N>1 path on CPU (torch.distributed, gloo, world_size 2), driving the PRODUCT's sharding functions:

 * phase-1 shards: contiguous 12-mer prefix ranges of equal cost; the union of the per-range seed sets is the seed set
   (data source on CPU: the pinned seed oracle -- the GPU entry point takes the same prefix range);
 * the A-contig partition (fga_partition_contigs) from all-reduced per-contig counts -- the same map on every rank;
 * the record gather (fastga_amd.parallel.gather_records) + fga_alns_concat + the redundancy filter on rank 0 give
   exactly what the filter gives on the undivided record set.
The device side of the same path (fga_seeds_contig_histogram / split_to / import, fga_session_merge / align / finish) is
covered on the GPU by tests/test_parts_gpu.py."""
import ctypes as C
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(target, args, world=2):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda x: x[0])
    return res


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


# ------------------------------------------------------------------------------------------------ phase-1 shards

def _merge_worker(rank, world, port, ra, rb, q):
    dist = _init(rank, world, port)
    from fastga_amd.gixio import Gix
    from fastga_amd.parallel import prefix_shards, all_reduce_counts
    from oracle import harness as H
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    shards = prefix_shards(A.index, B.index, world)
    b, e = shards[rank]
    n, c, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte, pfirst=b, plast=e)
    mine = np.zeros(world, dtype=np.int64)
    mine[rank] = nh
    counts = all_reduce_counts(dist, mine, "cpu").tolist()
    q.put((rank, shards, counts, n, c))
    dist.barrier()
    dist.destroy_process_group()


def test_prefix_sharded_merge_world2(toy_pair):
    from fastga_amd.gixio import Gix
    from oracle import harness as H
    d, ra, rb = toy_pair
    res = _spawn(_merge_worker, (ra, rb))
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    fn, fc, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte)
    w = 1 + A.pbyte + B.pbyte
    shards = res[0][1]
    assert shards[0][0] == 0 and shards[-1][1] == 1 << 24 and shards[0][1] == shards[1][0]
    assert res[0][2] == res[1][2] and sum(res[0][2]) == nh           # same reduced vector on both ranks
    assert min(res[0][2]) > 0.25 * nh                                  # roughly balanced
    un = b"".join(r[3] for r in res)
    uc = b"".join(r[4] for r in res)
    assert np.array_equal(H.sorted_records(un, w), H.sorted_records(fn, w))
    assert np.array_equal(H.sorted_records(uc, w), H.sorted_records(fc, w))


def test_prefix_shards_cover_and_balance(toy_pair):
    from fastga_amd.gixio import Gix
    from fastga_amd.parallel import prefix_shards
    d, ra, rb = toy_pair
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    for n in (1, 2, 3, 8):
        sh = prefix_shards(A.index, B.index, n)
        assert len(sh) == n and sh[0][0] == 0 and sh[-1][1] == 1 << 24
        assert all(sh[i][1] == sh[i + 1][0] for i in range(n - 1))
        tot = int(A.index[-1]) + int(B.index[-1])
        for b, e in sh:
            cnt = int(A.index[e - 1]) + int(B.index[e - 1]) - (int(A.index[b - 1]) + int(B.index[b - 1]) if b else 0)
            assert abs(cnt - tot / n) < 0.05 * tot + 100


# ------------------------------------------------------------------------------------------------ A-contig partition

def test_partition_is_balanced_and_a_pure_function(built_library):
    from fastga_amd.parallel import partition_contigs
    rng = np.random.default_rng(4)
    for nctg, nparts in ((1, 1), (5, 8), (40, 2), (40, 8), (1000, 8), (32, 8)):
        w = (rng.pareto(1.5, nctg) * 1e6).astype(np.int64) + rng.integers(0, 3, nctg)
        sel = partition_contigs(w, nparts)
        assert sel.min() >= 0 and sel.max() < nparts
        assert np.array_equal(sel, partition_contigs(w.copy(), nparts))
        load = np.bincount(sel, weights=w, minlength=nparts)
        # longest-processing-time bound: no part exceeds the mean by more than the heaviest contig
        assert load.max() <= w.sum() / nparts + w.max() + nctg
        if nctg >= nparts:
            assert len(set(sel.tolist())) == nparts                    # every part owns a contig
    # equal weights (e.g. a genome without seeds): contigs still spread over the parts
    assert len(set(partition_contigs(np.zeros(16, np.int64), 4).tolist())) == 4


def test_partition_in_original_order_is_contiguous(built_library):
    """the deal fga_session_run uses when the parts' records stream to the .1aln: part numbers do not decrease along the
    contigs' original order, so part p's records all come before part p+1's"""
    from fastga_amd.parallel import partition_contigs_in_order
    rng = np.random.default_rng(5)
    for nctg, nparts in ((1, 1), (5, 8), (40, 2), (40, 8), (1000, 8), (24, 3)):
        w = (rng.pareto(1.5, nctg) * 1e6).astype(np.int64)
        perm = rng.permutation(nctg).astype(np.int32)
        sel = partition_contigs_in_order(w, perm, nparts)
        assert sel.min() >= 0 and sel.max() < nparts
        inv = np.empty(nctg, np.int64); inv[perm] = np.arange(nctg)
        along = sel[inv]                                               # part of original contig 0, 1, ..
        assert np.all(np.diff(along) >= 0)
        load = np.bincount(sel, weights=w, minlength=nparts)
        assert load.max() <= w.sum() / nparts + w.max() + nparts       # a stretch ends with the contig that fills its share
    # contigs an index pads a short genome with (perm beyond the genome's own, no seeds) go along at the end
    sel = partition_contigs_in_order(np.array([5, 5, 0, 0], np.int64), np.array([1, 0, 2, 3], np.int32), 2)
    assert sel.tolist() == [1, 0, 1, 1]


# ------------------------------------------------------------------------------------------------ record gather

def _raw_records(seed=11, ngroups=60):
    """a synthetic set of accepted alignments: `ngroups` contig pairs x strands of overlapping records (the stress
    generator of tests/test_filter_oracle.py), units numbered in discovery order"""
    from tests.test_filter_oracle import _random_group
    from fastga_amd.device import ALN_DTYPE
    rng = np.random.default_rng(seed)
    recs, tbs, off, unit = [], [], 0, 0
    for g in range(ngroups):
        r, tb = _random_group(rng)
        r["aread"] = g % 23
        r["bread"] = (g * 7) % 11
        r["flags"] = (g // 23) & 1
        r["unit"] += unit
        unit = int(r["unit"].max()) + 1
        r["toff"] += off
        off += len(tb)
        recs.append(r)
        tbs.append(tb)
    # discovery order is (unit, seq) with units in key order: (strand, aread, bread)
    allr = np.concatenate(recs)
    key = np.lexsort((allr["seq"], allr["unit"]))
    return allr[key], np.concatenate(tbs)


def _filter(L, recs, tb):
    from fastga_amd.lib import Alns
    from fastga_amd.device import ALN_DTYPE
    a = Alns(len(recs), len(tb), 0, 0, recs.ctypes.data, tb.ctypes.data)
    out = C.POINTER(Alns)()
    assert L.fga_filter_alignments_mt(C.byref(a), 2, C.byref(out)) == 0
    o = out.contents
    got = np.frombuffer((C.c_char * (o.naln * ALN_DTYPE.itemsize)).from_address(o.alns), dtype=ALN_DTYPE).copy()
    gt = np.frombuffer((C.c_char * max(o.ntrace, 1)).from_address(o.tbytes), dtype=np.uint8)[:o.ntrace].copy()
    L.fga_alns_free(out)
    return got, gt


def _subset(recs, tb, keep):
    """the records `keep` selects, with their trace bytes repacked and units renumbered from 0 (a rank only knows its own)"""
    sub = recs[keep].copy()
    pieces, off = [], 0
    for i in range(len(sub)):
        t0, tl = int(sub["toff"][i]), int(sub["tlen"][i])
        pieces.append(tb[t0:t0 + tl])
        sub["toff"][i] = off
        off += tl
    _, inv = np.unique(sub["unit"], return_inverse=True)
    sub["unit"] = inv.astype(np.int32)
    return sub, (np.concatenate(pieces) if pieces else np.zeros(0, np.uint8))


def _gather_worker(rank, world, port, q):
    dist = _init(rank, world, port)
    from fastga_amd.lib import load_library, Alns
    from fastga_amd.parallel import (partition_contigs, all_reduce_counts, gather_records, arrays_to_alns)
    L = load_library()
    recs, tb = _raw_records()
    nctg = int(recs["aread"].max()) + 1
    # per-contig weights: every rank counts the records of "its half of the prefix space", then all-reduce
    half = np.arange(len(recs)) % world == rank
    w = all_reduce_counts(dist, np.bincount(recs["aread"][half], minlength=nctg), "cpu")
    select = partition_contigs(w, world)
    mine, mtb = _subset(recs, tb, select[recs["aread"]] == rank)
    allr = gather_records(dist, mine, mtb, (len(mine), 7 * len(mine)), "cpu")
    res = None
    if rank == 0:
        keep, ptrs = [], (C.POINTER(Alns) * world)()
        for r in range(world):
            a, k = arrays_to_alns(allr[r][0], allr[r][1], allr[r][2])
            keep.append((a, k))
            ptrs[r] = C.pointer(a)
        cat = C.POINTER(Alns)()
        assert L.fga_alns_concat(ptrs, world, C.byref(cat)) == 0
        c = cat.contents
        assert c.naln == len(recs) and c.ncalls == len(recs) and c.nwaves == 7 * len(recs)
        from fastga_amd.device import ALN_DTYPE
        crecs = np.frombuffer((C.c_char * (c.naln * ALN_DTYPE.itemsize)).from_address(c.alns), dtype=ALN_DTYPE).copy()
        ctb = np.frombuffer((C.c_char * max(c.ntrace, 1)).from_address(c.tbytes), dtype=np.uint8)[:c.ntrace].copy()
        L.fga_alns_free(cat)
        got = _filter(L, crecs, ctb)
        exp = _filter(L, recs, tb)
        res = (got[0].tobytes() == exp[0].tobytes(), got[1].tobytes() == exp[1].tobytes(), len(exp[0]), len(recs),
               select.tolist())
    else:
        res = (select.tolist(),)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gathered_parts_filter_like_the_whole_world2(built_library):
    res = _spawn(_gather_worker, ())
    same_recs, same_trace, nlive, nraw, sel0 = res[0][1]
    assert same_recs and same_trace and 0 < nlive < nraw
    assert res[1][1][0] == sel0 and len(set(sel0)) == 2          # both ranks derived the same two-part map


# ------------------------------------------------------------------------------------------------ filter per part, merge

def _merge_filtered(L, sets, nthreads=None):
    """fga_alns_merge_filtered[_mt] over [(records, trace bytes)]"""
    from fastga_amd.lib import Alns
    from fastga_amd.device import ALN_DTYPE
    keep, ptrs = [], (C.POINTER(Alns) * max(len(sets), 1))()
    for k, (r, t) in enumerate(sets):
        r, t = np.ascontiguousarray(r), np.ascontiguousarray(t)
        a = Alns(len(r), len(t), 0, 0, r.ctypes.data, t.ctypes.data)
        keep.append((a, r, t))
        ptrs[k] = C.pointer(a)
    out = C.POINTER(Alns)()
    if nthreads is None:
        assert L.fga_alns_merge_filtered(ptrs, len(sets), C.byref(out)) == 0
    else:
        assert L.fga_alns_merge_filtered_mt(ptrs, len(sets), nthreads, C.byref(out)) == 0
    o = out.contents
    got = np.frombuffer((C.c_char * (o.naln * ALN_DTYPE.itemsize)).from_address(o.alns), dtype=ALN_DTYPE).copy()
    gt = np.frombuffer((C.c_char * max(o.ntrace, 1)).from_address(o.tbytes), dtype=np.uint8)[:o.ntrace].copy()
    L.fga_alns_free(out)
    return got, gt


@pytest.mark.parametrize("nparts", [1, 2, 3, 8])
def test_parts_filtered_on_their_own_merge_to_the_filter_of_the_whole(built_library, nparts):
    """what fga_session_run does over A-contig parts (a background filter per part) and what every rank does before the
    gather: filter the part's records, then lay the parts' runs per A contig out in contig order"""
    from fastga_amd.lib import load_library
    from fastga_amd.parallel import partition_contigs
    L = load_library()
    recs, tb = _raw_records(seed=5, ngroups=90)
    exp = _filter(L, recs, tb)
    nctg = int(recs["aread"].max()) + 1
    select = partition_contigs(np.bincount(recs["aread"], minlength=nctg), nparts)
    sets = []
    for p in range(nparts):
        sub, stb = _subset(recs, tb, select[recs["aread"]] == p)
        sets.append(_filter(L, sub, stb))
    got = _merge_filtered(L, sets)
    assert len(exp[0]) > 0
    assert got[0].tobytes() == exp[0].tobytes() and got[1].tobytes() == exp[1].tobytes()
    # inputs that share an A contig (not what the pipeline produces) still merge by the rest of the key
    half = np.arange(len(exp[0])) % 2 == 0
    a, b = _subset(exp[0], exp[1], half), _subset(exp[0], exp[1], ~half)
    m = _merge_filtered(L, [a, b])
    for f in ("aread", "abpos", "bread", "aepos", "bepos", "diffs", "tlen"):
        assert np.array_equal(np.sort(m[0][f]), np.sort(exp[0][f]))
    key = m[0]["aread"].astype(np.int64) << 32 | m[0]["abpos"]
    assert np.all(np.diff(key) >= 0)


def test_merge_of_filtered_parts_on_all_threads(built_library):
    """fga_alns_merge_filtered_mt: > 50 k records so that the copies are made by a team, runs longer than a copy task
    (64 k records), same bytes as on one thread"""
    from tests.test_filter_threads import _random_set, _run
    L = built_library
    rng = np.random.default_rng(3)
    alns, tb = _random_set(rng, 40000, 8, nctg=3)
    fa, ft = _run(L, alns, tb, 8)
    assert len(fa) > 130000
    sets = [_subset(fa, ft, fa["aread"] == c) for c in (2, 0, 1)]
    one = _merge_filtered(L, sets)
    assert one[0].tobytes() == fa.tobytes() and one[1].tobytes() == ft.tobytes()
    for nt in (2, 8, 32):
        got = _merge_filtered(L, sets, nthreads=nt)
        assert got[0].tobytes() == one[0].tobytes() and got[1].tobytes() == one[1].tobytes()


def _filtered_gather_worker(rank, world, port, q):
    dist = _init(rank, world, port)
    from fastga_amd.lib import load_library
    from fastga_amd.parallel import partition_contigs, all_reduce_counts, gather_records
    L = load_library()
    recs, tb = _raw_records(seed=8, ngroups=70)
    nctg = int(recs["aread"].max()) + 1
    half = np.arange(len(recs)) % world == rank
    select = partition_contigs(all_reduce_counts(dist, np.bincount(recs["aread"][half], minlength=nctg), "cpu"), world)
    mine, mtb = _subset(recs, tb, select[recs["aread"]] == rank)
    fil = _filter(L, mine, mtb)                                     # on the rank that owns the contig pairs
    allr = gather_records(dist, fil[0], fil[1], (0, 0), "cpu")
    res = None
    if rank == 0:
        got = _merge_filtered(L, [(allr[r][0], allr[r][1]) for r in range(world)])
        exp = _filter(L, recs, tb)
        res = (got[0].tobytes() == exp[0].tobytes(), got[1].tobytes() == exp[1].tobytes(), len(exp[0]))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_filter_before_the_gather_world2(built_library):
    res = _spawn(_filtered_gather_worker, ())
    same_recs, same_trace, nlive = res[0][1]
    assert same_recs and same_trace and nlive > 0
