"""fga_run_multi / `FastGA -G<n>`: ONE comparison cut over several GPUs from one process, behind the C-ABI (the reference's
parts machinery inside one process: Select[] / unit matrix FastGA.c:5057-5134, transpose + NPARTS loop 5160-5204, la_merge
3991-4133).  On the GPU box the device list names GPU 0 several times -- virtual ranks on one GPU: every rank has its own
host thread, HIP stream and sliced session, the seeds move between the ranks' buffers exactly as between devices -- and
the result must be the reference's .1aln line for line whatever the number of ranks.  The argument handling is tested
on the CPU."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fastga_amd", "bin", "FastGA")


def _view(path):
    from oracle import harness as H
    return H.oneview(path)


def _sans_date(path):
    """the file's bytes with the provenance line's time stamp blanked (two runs may straddle a second)"""
    import re
    return re.sub(rb"\d{4}-\d\d-\d\d_\d\d:\d\d:\d\d", b"D", open(path, "rb").read())


def _keep(lines):
    return [ln for ln in lines if ln[0] not in "!<"]


# ------------------------------------------------------------------------------------------------ CPU: arguments

def test_run_multi_argument_checks(built_library, toy_pair):
    from fastga_amd.lib import RunParams, RunStats
    L = built_library
    d, ra, rb = toy_pair
    prm, st = RunParams(), RunStats()
    prm.freq, prm.nthreads, prm.align_min, prm.align_rate = 10, 4, 100, 0.3
    prm.chain_break, prm.chain_min = 2000, 170
    dev2 = (C.c_int * 2)(0, 0)
    for ndev, devs in ((0, dev2), (65, dev2), (2, None), (-1, dev2)):
        rc = L.fga_run_multi(ra.encode(), rb.encode(), C.byref(prm), ndev, devs, C.byref(st))
        assert rc == 1 and b"fga_run_multi: bad argument" in L.fga_last_error(), (ndev, L.fga_last_error())
    rc = L.fga_run_multi(None, rb.encode(), C.byref(prm), 2, dev2, C.byref(st))
    assert rc == 1 and b"bad argument" in L.fga_last_error()
    import torch
    if not torch.cuda.is_available():                     # the product path has no CPU fallback, with any number of devices
        rc = L.fga_run_multi(ra.encode(), rb.encode(), C.byref(prm), 2, dev2, C.byref(st))
        assert rc == 1 and b"no CPU fallback" in L.fga_last_error()
        assert L.fga_dev_device_count() == 0
    else:
        n = L.fga_dev_device_count()
        assert n >= 1
        bad = (C.c_int * 2)(0, n)                         # a device the node does not have
        rc = L.fga_run_multi(ra.encode(), rb.encode(), C.byref(prm), 2, bad, C.byref(st))
        assert rc == 1 and b"out of range" in L.fga_last_error()


def test_multi_session_argument_checks(built_library, toy_pair):
    from fastga_amd.lib import RunParams, RunStats
    L = built_library
    d, ra, rb = toy_pair
    prm, st = RunParams(), RunStats()
    prm.freq, prm.nthreads, prm.align_min, prm.align_rate = 10, 4, 100, 0.3
    prm.chain_break, prm.chain_min = 2000, 170
    dev2 = (C.c_int * 2)(0, 0)
    h = C.c_void_p()
    for ndev, devs in ((0, dev2), (65, dev2), (2, None)):
        rc = L.fga_multi_open(ra.encode(), rb.encode(), C.byref(prm), ndev, devs, C.byref(h))
        assert rc == 1 and b"fga_multi_open: bad argument" in L.fga_last_error() and not h.value
    assert L.fga_multi_open(ra.encode(), rb.encode(), C.byref(prm), 2, dev2, None) == 1
    assert L.fga_multi_run(None, C.byref(prm), C.byref(st)) == 1 and b"null argument" in L.fga_last_error()
    L.fga_multi_close(None)                               # a no-op
    assert L.fga_multi_ndev(None) == 0
    import torch
    if not torch.cuda.is_available():                     # no CPU fallback behind the session form either
        rc = L.fga_multi_open(ra.encode(), rb.encode(), C.byref(prm), 2, dev2, C.byref(h))
        assert rc == 1 and b"no CPU fallback" in L.fga_last_error() and not h.value
        rc = L.fga_multi_open(os.path.join(d, "nothing_here").encode(), None, C.byref(prm), 2, dev2, C.byref(h))
        assert rc == 1 and not h.value                    # a missing genome is reported before the device


def test_cli_gpu_option_grammar(built_library, toy_pair, tmp_path):
    d, ra, rb = toy_pair
    def run(args, env=None):
        e = dict(os.environ)
        e.pop("FGA_DEVICES", None)
        if env:
            e.update(env)
        return subprocess.run([EXE, *args], cwd=str(tmp_path), capture_output=True, text=True, env=e)
    r = run(["-G0", "-1:x", ra, rb])
    assert r.returncode == 1 and "-G number of GPUs must be in [1,64]" in r.stderr
    r = run(["-G65", "-1:x", ra, rb])
    assert r.returncode == 1 and "[1,64]" in r.stderr
    r = run(["-Gx", "-1:x", ra, rb])
    assert r.returncode == 1 and "not an integer" in r.stderr
    r = run(["-1:x", ra, rb], env={"FGA_DEVICES": "0,a"})
    assert r.returncode == 1 and "FGA_DEVICES must be a comma-separated list" in r.stderr
    r = run(["-1:x", ra, rb], env={"FGA_DEVICES": "0;1"})
    assert r.returncode == 1 and "FGA_DEVICES" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = run(["-v", "-G2", "-1:x", ra, rb])
        assert r.returncode == 1 and "no CPU fallback" in r.stderr and "Using 2 GPUs (0,1)" in r.stderr
        r = run(["-v", "-1:x", ra, rb], env={"FGA_DEVICES": "0,0,0"})
        assert r.returncode == 1 and "no CPU fallback" in r.stderr and "Using 3 GPUs (0,0,0)" in r.stderr
        assert not os.path.exists(os.path.join(str(tmp_path), "x.1aln"))
    assert "-G<int>" in run([]).stderr                     # the usage text names the option


# ------------------------------------------------------------------------------------------------ GPU: parity

def _ngpu():
    from fastga_amd.lib import load_library
    return load_library().fga_dev_device_count()


def _device_lists():
    """virtual ranks on GPU 0 everywhere; distinct devices -- the hipMemcpyPeerAsync branch of fga_seeds_import_peer, peer
    access, buffers released on another device's thread -- on a node that has them"""
    lists = [(0,), (0, 0), (0, 0, 0, 0), (0, 0, 0)]
    n = _ngpu()
    if n >= 2:
        lists += [(0, 1), (1, 0), (1, 1, 0)]
    if n >= 4:
        lists += [(0, 1, 2, 3)]
    if n >= 8:
        lists += [tuple(range(8))]
    return lists


def _reference(ra, rb, w, flags=(), threads=8):
    from oracle import harness as H
    if not H.have_reference():
        pytest.skip("oracle/_ref did not travel")
    H.ref_fastga(ra, rb, w, os.path.join(w, "ref"), threads=threads, flags=flags)
    return _keep(H.oneview(os.path.join(w, "ref.1aln")))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pair", "symmetric", "self"])
def test_run_multi_with_virtual_ranks_is_the_reference_line_for_line(toy_pair, tmp_path, mode):
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    w = str(tmp_path)
    b = None if mode == "self" else rb
    kw = dict(symmetric=True, freq=6) if mode == "symmetric" else {}
    ref = _reference(ra, b, w, flags=("-S", "-f6") if mode == "symmetric" else ())
    one = D.run(ra, b, os.path.join(w, "one.1aln"), nthreads=8, reference_threads=8, **kw)
    assert _keep(_view(os.path.join(w, "one.1aln"))) == ref
    for devices in _device_lists():
        out = os.path.join(w, "multi%s.1aln" % "".join(map(str, devices)))
        st = D.run_multi(ra, b, out, devices=devices, nthreads=8, reference_threads=8, **kw)
        assert _keep(_view(out)) == ref, devices
        assert st["nalns"] == one["nalns"] and st["nlive"] == one["nlive"] and st["nhits"] == one["nhits"]
        assert st["nparts"] == len(devices)
        if mode == "self":       # self totals are halved per merge launch (FastGA.c:1906): the floors add up differently
            assert 0 <= one["nseeds"] - st["nseeds"] <= len(devices)
        else:
            assert st["nseeds"] == one["nseeds"]


@pytest.mark.gpu
def test_run_multi_builds_its_index_slices_on_the_devices(toy_pair, tmp_path):
    """no index files: every rank counts the 12-mer prefixes of both genomes on its device, cuts the same ranges and builds
    only its slice (fga_dgix_build_range); PAF with CIGARs comes from rank 0's device"""
    from fastga_amd import device as D
    from oracle import harness as H
    d, ra, rb = toy_pair
    w = str(tmp_path)
    for root in (ra, rb):                                  # the GDBs alone
        base = os.path.basename(root)
        for f in (base + ".gdb", "." + base + ".bps"):
            shutil.copy(os.path.join(os.path.dirname(root), f), os.path.join(w, f))
    a, b = os.path.join(w, os.path.basename(ra)), os.path.join(w, os.path.basename(rb))
    ref = _reference(ra, rb, w)
    paf = os.path.join(w, "m.paf")
    st = D.run_multi(a, b, os.path.join(w, "m.1aln"), devices=(0, 0, 0), nthreads=8, reference_threads=8,
                     paf_path=paf, paf_flags=2)
    assert _keep(_view(os.path.join(w, "m.1aln"))) == ref
    assert not [f for f in os.listdir(w) if f.endswith(".gix") or ".ktab." in f]
    exp = H.run([H.ref_bin("ALNtoPAF"), "-T4", "-x", os.path.join(w, "ref.1aln")], cwd=w).stdout
    assert open(paf).read() == exp and st["trace_kernel_ms"] > 0


@pytest.mark.gpu
def test_cli_with_gpu_list_is_the_reference(toy_pair, tmp_path):
    d, ra, rb = toy_pair
    w = str(tmp_path)
    ref = _reference(ra, rb, w, threads=4)
    for args, env in ((["-G1"], {}), ([], {"FGA_DEVICES": "0,0"}), ([], {"FGA_DEVICES": "0,0,0,0"})):
        e = dict(os.environ)
        e.pop("FGA_DEVICES", None)
        e.update(env)
        out = os.path.join(w, "cli%d" % len(env.get("FGA_DEVICES", "0")))
        r = subprocess.run([EXE, "-v", "-T4", *args, "-1:" + out, ra, rb], cwd=w, capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stderr
        assert _keep(_view(out + ".1aln")) == ref, (args, env)
        if env:
            assert "Using %d GPUs" % len(env["FGA_DEVICES"].split(",")) in r.stderr


@pytest.mark.gpu
def test_run_multi_again_and_again(toy_pair, tmp_path):
    """three virtual ranks = three host threads opening sessions, uploading genomes and launching kernels at once on their own
    (non-blocking) streams, thirty times over: every run is the single-session run's file.  (Pins a race this flow found:
    hipMemset runs on the legacy default stream, is not ordered with a non-blocking stream and returns before it has happened
    -- with several contexts at work the zeroing of the complement genome image landed after revcomp_kernel had written it in
    one run of thirty, and alignments of that strand went missing.  All fills are on the contexts' streams now.)"""
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    w = str(tmp_path)
    D.run(ra, rb, os.path.join(w, "one.1aln"), nthreads=8, reference_threads=8)
    ref = _keep(_view(os.path.join(w, "one.1aln")))
    out = os.path.join(w, "m.1aln")
    for k in range(30):
        D.run_multi(ra, rb, out, devices=(0, 0, 0), nthreads=8, reference_threads=8)
        assert _keep(_view(out)) == ref, k


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pair", "self"])
def test_multi_session_runs_warm_and_reweighs_its_parts(toy_pair, tmp_path, mode):
    """fga_multi_open / run / run / run / close: the inputs stay on the devices, the rank threads stay alive, and from the
    second run on the contigs are dealt to the ranks by the wave steps their units took the run before (the first: by seed
    counts) -- every run is the reference's file line for line, with any parameters of the run"""
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    w = str(tmp_path)
    b = None if mode == "self" else rb
    ref = _reference(ra, b, w)
    ref6 = None
    for devices in [dl for dl in _device_lists() if len(dl) > 1][:4]:
        M = D.Multi(ra, b, devices=devices, nthreads=8)
        try:
            for k in range(3):
                out = os.path.join(w, "s%d.1aln" % k)
                st = M.run(out_path=out, nthreads=8, reference_threads=8)
                assert _keep(_view(out)) == ref, (devices, k)
                assert st["nparts"] == len(devices)
                rs = M.rank_stats()
                assert len(rs) == len(devices) and sum(r["wave_steps"] for r in rs) == st["nwaves"]
            if mode == "pair":                             # other parameters on the same session
                if ref6 is None:
                    ref6 = _reference(ra, b, os.path.join(w), flags=("-S", "-f6"))
                out = os.path.join(w, "s6.1aln")
                M.run(out_path=out, nthreads=8, reference_threads=8, symmetric=True, freq=6)
                assert _keep(_view(out)) == ref6, devices
        finally:
            M.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["pair", "self"])
def test_multi_ranks_stream_their_stretches_of_the_1aln(toy_pair, tmp_path, mode, monkeypatch):
    """with the A contigs dealt to the ranks in original order every rank puts its own records into the reference's tie order
    and appends them to the .1aln when the ranks before it have (fga_aln_stream_*): the file is the one rank 0 writes from
    the gathered records, byte for byte, and the reference's line for line"""
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    w = str(tmp_path)
    b = None if mode == "self" else rb
    ref = _reference(ra, b, w)
    for devices in [dl for dl in _device_lists() if len(dl) > 1][:4]:
        M = D.Multi(ra, b, devices=devices, nthreads=8)
        try:
            for k, (env, want) in enumerate((("2", len(devices)), ("0", 0), ("2", len(devices)))):
                monkeypatch.setenv("FGA_STREAM_PARTS", env)       # 2: whatever the balance of the contiguous deal
                out = os.path.join(w, "t%d.1aln" % k)
                st = M.run(out_path=out, nthreads=8, reference_threads=8, command_line="FastGA test")
                assert st["streamed_parts"] == want, (devices, env)
                assert _keep(_view(out)) == ref, (devices, env)
            one = [_sans_date(os.path.join(w, "t%d.1aln" % k)) for k in range(3)]
            assert one[0] == one[1] == one[2], devices
        finally:
            M.close()


@pytest.mark.gpu
@pytest.mark.skipif("_ngpu() < 2")
def test_two_real_devices_leave_the_callers_device_alone(toy_pair, tmp_path):
    """the ranks' threads visit their devices; the calling thread's current device is what it was"""
    import torch
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    torch.cuda.set_device(1)
    D.run_multi(ra, rb, os.path.join(str(tmp_path), "m.1aln"), devices=(0, 1), nthreads=8)
    assert torch.cuda.current_device() == 1
    torch.cuda.set_device(0)
