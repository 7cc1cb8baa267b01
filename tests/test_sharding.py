"""N>1 path on CPU: prefix-range sharding of the seed merge with torch.distributed (gloo, world_size 2).
Each rank runs its shard (here through the CPU oracle -- the GPU entry point takes the same prefix range), the
per-rank seed counts are all-gathered, and the union of the shards is the full seed multiset."""
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


def _worker(rank, world, port, ra, rb, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastga_amd.gixio import Gix
    from fastga_amd.parallel import prefix_shards, gather_counts
    from oracle import harness as H
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    shards = prefix_shards(A.index, B.index, world)
    b, e = shards[rank]
    n, c, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte, pfirst=b, plast=e)
    counts = gather_counts(dist, nh)
    q.put((rank, shards, counts, n, c))
    dist.barrier()
    dist.destroy_process_group()


def test_prefix_sharded_merge_world2(toy_pair):
    import torch.multiprocessing as mp
    from fastga_amd.gixio import Gix
    from oracle import harness as H
    d, ra, rb = toy_pair
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ra, rb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    A, B = Gix(ra + ".gix"), Gix(rb + ".gix")
    fn, fc, nh, ts = H.oracle_seed_merge(A.table, A.index, A.pbyte, B.table, B.index, B.pbyte)
    w = 1 + A.pbyte + B.pbyte
    shards = res[0][1]
    assert shards[0][0] == 0 and shards[-1][1] == 1 << 24 and shards[0][1] == shards[1][0]
    assert res[0][2] == res[1][2] and sum(res[0][2]) == nh           # same gathered vector on both ranks
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
