"""GPU: one comparison cut into A-contig parts (the reference's Select[] / seed file matrix / la_merge,
FastGA.c:5057-5134, 5160-5184, 3991-4133).  The multi-GPU run's C-ABI calls are driven on ONE GPU for 1, 2, 4 and 8 parts
(fastga_amd.parallel.run_parts_on_one_gpu: the RCCL all-to-all-v replaced by local slicing) and must give the
reference's .1aln every time; fga_session_run's own multi-pass mode (pass_seeds) likewise; the routing kernels are
checked record for record."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _view(path):
    from oracle import harness as H
    return H.oneview(path)


def _sans_date(path):
    """the file's bytes with the provenance line's time stamp blanked (two runs may straddle a second)"""
    import re
    return re.sub(rb"\d{4}-\d\d-\d\d_\d\d:\d\d:\d\d", b"D", open(path, "rb").read())


def test_routing_kernels_histogram_split_import(toy_pair):
    from fastga_amd import device as D
    from fastga_amd.parallel import partition_contigs
    d, ra, rb = toy_pair
    ses = D.Session(ra, rb)
    prm, st = ses.params(), ses.new_stats()
    seeds = ses.merge(prm, st)
    host = seeds.download()
    actg = (host["actg"] >> 8).astype(np.int64)
    hist = ses.contig_histogram(seeds)
    assert np.array_equal(hist, np.bincount(actg, minlength=ses.nctg))
    for nparts in (1, 3, 8):
        sel = partition_contigs(hist, nparts)
        buf = ses.dev_malloc(16 * len(host))
        off = ses.split_to(seeds, sel, nparts, buf)
        got = ses.dev_download(buf, 16 * len(host)).view(D.SEED_DTYPE).reshape(-1)
        assert off[0] == 0 and off[-1] == len(host)
        void = np.dtype((np.void, 16))
        for p in range(nparts):
            piece = got[off[p]:off[p + 1]]
            assert np.all(sel[(piece["actg"] >> 8).astype(np.int64)] == p)
            exp = host[sel[actg] == p]
            assert np.array_equal(np.sort(piece.view(void)), np.sort(exp.view(void)))
        # import: pieces from two "ranks" become one seed buffer
        mid = len(host) // 2
        imp = ses.import_seeds([(buf, mid), (buf + 16 * mid, len(host) - mid)])
        assert np.array_equal(imp.download().view(void), got.view(void))
        imp.free()
        ses.dev_free(buf)
    seeds.free()
    ses.close()


def _reference(ra, rb, w, flags=()):
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=8, flags=flags)
    return H.oneview(os.path.join(w, "ref.1aln"))


@pytest.mark.parametrize("mode", ["pair", "symmetric", "self"])
def test_any_number_of_parts_gives_the_reference_1aln(toy_pair, tmp_path, mode):
    from fastga_amd import device as D, workload
    from fastga_amd.parallel import run_parts_on_one_gpu
    d, ra, rb = toy_pair
    w = str(tmp_path)
    b = None if mode == "self" else rb
    kw = dict(symmetric=True, freq=6) if mode == "symmetric" else {}
    ref = _reference(ra, b, w, flags=("-S", "-f6") if mode == "symmetric" else ())
    ses = D.Session(ra, b)
    kw["reference_threads"] = 8                # ties on (aread, abpos) as FastGA -T8 orders them
    whole = ses.run(out_path=os.path.join(w, "whole.1aln"), nthreads=8, **kw)
    base = _view(os.path.join(w, "whole.1aln"))
    assert workload.digest_1aln(base) == workload.digest_1aln(ref)
    assert base == ref                          # line for line
    for nparts in (1, 2, 4, 8):
        out = os.path.join(w, f"parts{nparts}.1aln")
        st = run_parts_on_one_gpu(ses, nparts, out_path=out, nthreads=8, **kw)
        assert _view(out) == base, nparts                         # identical to the undivided run, line for line
        assert st["nalns"] == whole["nalns"] and st["nlive"] == whole["nlive"]
        if mode == "self":      # self totals are halved per merge launch (FastGA.c:1906): floors add up differently
            assert 0 <= whole["nseeds"] - st["nseeds"] <= nparts
            assert 0 <= sum(st["part_seed_counts"]) - 2 * whole["nseeds"] <= 1
        else:
            assert st["nseeds"] == whole["nseeds"] and sum(st["part_seed_counts"]) == whole["nseeds"]
        if nparts > 1:
            assert min(st["part_seed_counts"]) > 0
    ses.close()


def test_session_run_in_several_passes(toy_pair, tmp_path):
    """fga_session_run cuts phase 2 into A-contig parts by itself when the merge finds more seeds than one pass should
    take (pass_seeds): the reference's NPARTS loop (FastGA.c:5186-5204)"""
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    w = str(tmp_path)
    one = D.run(ra, rb, os.path.join(w, "one.1aln"), nthreads=8)
    many = D.run(ra, rb, os.path.join(w, "many.1aln"), nthreads=8, pass_seeds=max(1, one["nseeds"] // 5))
    assert one["nparts"] == 1 and many["nparts"] >= 5
    assert _view(os.path.join(w, "many.1aln")) == _view(os.path.join(w, "one.1aln"))
    assert many["nseeds"] == one["nseeds"] and many["nhits"] == one["nhits"] and many["nlive"] == one["nlive"]


@pytest.mark.parametrize("self_mode", [False, True])
def test_parts_streamed_to_the_1aln_while_later_parts_run(toy_pair, tmp_path, self_mode, monkeypatch):
    """with the A contigs dealt to the parts in original order, each part's records are filtered, put in the reference's
    tie order and appended to the .1aln while the next part's kernels run (fga_aln_stream_*): the same bytes as the file
    written after the last part, and as the one-pass run's"""
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    w = str(tmp_path)
    b = None if self_mode else rb
    kw = dict(nthreads=8, reference_threads=8)
    one = D.run(ra, b, os.path.join(w, "one.1aln"), **kw)
    lim = max(1, one["nseeds"] // 3)
    streamed = D.run(ra, b, os.path.join(w, "streamed.1aln"), pass_seeds=lim, **kw)
    monkeypatch.setenv("FGA_STREAM_PARTS", "0")
    after = D.run(ra, b, os.path.join(w, "after.1aln"), pass_seeds=lim, **kw)
    assert streamed["nparts"] == after["nparts"] >= 3 and after["streamed_parts"] == 0
    assert streamed["streamed_parts"] == streamed["nparts"], "the contiguous deal was judged too uneven: nothing streamed"
    ref = _sans_date(os.path.join(w, "one.1aln"))
    assert _sans_date(os.path.join(w, "streamed.1aln")) == ref
    assert _sans_date(os.path.join(w, "after.1aln")) == ref
    for k in ("nseeds", "nhits", "nalns", "nlive", "cover"):
        assert streamed[k] == after[k] == one[k], k


# ---- run_sharded itself with N = 2: two processes on cuda:0, backend gloo (the exchange staged through the host) ----------

def _sharded_worker(rank, world, port, ra, rb, out, q, self_mode):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fastga_amd import device as D
        from fastga_amd.parallel import run_sharded
        ses = D.Session(ra, None if self_mode else rb, rank=rank, nranks=world)      # holds its own slice of the tables only
        st = run_sharded(ses, dist, dict(out_path=out, nthreads=4, reference_threads=4), "cpu")
        ses.close()
        q.put((rank, {k: st[k] for k in ("nseeds", "nalns", "nlive", "part_seeds", "exchange_seeds_out")}))
    except Exception as e:                                  # report instead of hanging the other rank's collective
        q.put((rank, {"error": repr(e)}))
        raise
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("self_mode", [False, True])
def test_run_sharded_with_two_ranks_on_one_gpu(toy_pair, tmp_path, self_mode):
    """the real multi-GPU function -- prefix ranges, all-reduced contig histogram, partition, all-to-all-v of the seeds,
    phase 2 and the filter on the owning rank, gather, merge by A contig on rank 0 -- with world size 2.  RCCL refuses two
    ranks on one device, so the collectives are gloo's and the records are staged through the host; every C-ABI call and
    all of parallel.run_sharded's control flow are the ones an 8-GPU run executes.  Result: the reference's .1aln."""
    import socket
    import torch.multiprocessing as mp
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    d, ra, rb = toy_pair
    out = str(tmp_path / "sharded.1aln")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, ra, rb, out, q, self_mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert all("error" not in v for v in res.values()), res
    assert all(p.exitcode == 0 for p in procs)
    assert res[0]["part_seeds"] > 0 and res[1]["part_seeds"] > 0                 # both ranks owned contigs
    assert res[0]["exchange_seeds_out"] + res[1]["exchange_seeds_out"] > 0       # and seeds really changed ranks
    rd = str(tmp_path / "ref")
    os.makedirs(rd)
    H.ref_fastga(ra, None if self_mode else rb, rd, os.path.join(rd, "ref"), threads=4)
    assert _view(out) == _view(os.path.join(rd, "ref.1aln"))           # line for line: the tie order is FastGA -T4's


@pytest.mark.parametrize("mode", ["pair", "self", "files"])
def test_sliced_sessions_hold_their_prefix_range_only(toy_pair, tmp_path, mode):
    """every rank of an N-GPU run opens the session with ITS 12-mer prefix range of both tables (fga_session_open_sliced:
    the ranges are cut from the per-prefix counts on every rank alike; index built on the device with 1/N of the sort, or
    the slice of the index files uploaded): the slices' seeds are the whole table's seeds, nothing else can be merged,
    and the comparison put together from the slices is the undivided one"""
    import shutil
    from fastga_amd import device as D
    from fastga_amd.parallel import partition_contigs
    d, ra, rb = toy_pair
    w = str(tmp_path)
    if mode == "files":
        a, b = ra, rb                                              # toy_pair carries .gix files
    else:                                                          # GDBs only: indices are built on the device
        for r in (ra, rb):
            for f in (os.path.basename(r) + ".gdb", "." + os.path.basename(r) + ".bps"):
                shutil.copy(os.path.join(os.path.dirname(r), f), os.path.join(w, f))
        a, b = os.path.join(w, os.path.basename(ra)), os.path.join(w, os.path.basename(rb))
    if mode == "self":
        b = None
    whole = D.Session(a, b)
    prm, st = whole.params(), whole.new_stats()
    full = whole.merge(prm, st)
    void = np.dtype((np.void, 16))
    allseeds = np.sort(full.download().view(void))
    full.free()
    base_out = os.path.join(w, "whole.1aln")
    whole.run(out_path=base_out, nthreads=4)
    N = 3
    sessions = [D.Session(a, b, rank=r, nranks=N) for r in range(N)]
    cuts = [np.zeros(N + 1, dtype=np.int64) for _ in range(N)]
    for r, s in enumerate(sessions):
        from fastga_amd.lib import check
        import ctypes as C
        check(s.L.fga_session_prefix_cuts(s.h, N, cuts[r].ctypes.data_as(C.POINTER(C.c_int64))), "cuts")
        assert s.table_bytes < whole.table_bytes                    # a slice, not the table
    assert all(np.array_equal(cuts[0], c) for c in cuts) and cuts[0][0] == 0 and cuts[0][-1] == 1 << 24
    assert sum(s.table_bytes for s in sessions) == whole.table_bytes
    parts, hist = [], np.zeros(whole.nctg, dtype=np.int64)
    for r, s in enumerate(sessions):
        sd = s.merge(s.params(), s.new_stats(), int(cuts[0][r]), int(cuts[0][r + 1]))
        parts.append(sd.download())
        hist += s.contig_histogram(sd)
        sd.free()
        with pytest.raises(Exception):                              # another rank's range is not in this session
            o = (r + 1) % N
            s.merge(s.params(), s.new_stats(), int(cuts[0][o]), int(cuts[0][o + 1]))
    got = np.sort(np.concatenate(parts).view(void))
    if mode == "self":                                              # the halved totals of a self merge aside, the same records
        assert np.array_equal(np.unique(got), np.unique(allseeds)) and abs(len(got) - len(allseeds)) <= N
    else:
        assert np.array_equal(got, allseeds)
    # the comparison from the slices: every "rank" aligns the seeds of its A-contig part on its own session
    select = partition_contigs(hist, N)
    raws = []
    for p, s in enumerate(sessions):
        mine = np.concatenate([x[select[(x["actg"] >> 8).astype(np.int64)] == p] for x in parts])
        buf = s.dev_malloc(16 * max(len(mine), 1))
        if len(mine):
            s.dev_upload(buf, mine)
        part = s.import_seeds([(buf, len(mine))])
        s.dev_free(buf)
        stp = s.new_stats()
        raw = s.align(s.params(nthreads=4), stp, part)
        raws.append((s, s.filter(raw, nthreads=4)))
        s.free_alns(raw)
    out = os.path.join(w, "sliced.1aln")
    s0 = sessions[0]
    s0.finish_filtered(s0.params(out_path=out, nthreads=4), s0.new_stats(), [f for _, f in raws])
    for s, f in raws:
        s.free_alns(f)
    assert _view(out) == _view(base_out)
    for s in sessions:
        s.close()
    whole.close()
