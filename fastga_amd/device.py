"""Python view of the device-side C-ABI (include/fastga_amd.h): Device, DeviceGix, seed merge."""
import ctypes as C

import numpy as np

from .lib import (load_library, check, MergeParams, SortParams, ChainParams, Hits, ExtendParams, Alns,  # noqa: F401
                  FgaError, STAGE_MERGE,
                  STAGE_MERGE_PARTITION, STAGE_SORT, STAGE_CHAIN, STAGE_EXTEND)

SEED_DTYPE = np.dtype([("apos", "<u4"), ("bpos", "<u4"), ("actg", "<u4"), ("bctg", "<u4")])


class Device:
    def __init__(self, index=0):
        self.L = load_library()
        self.h = C.c_void_p()
        check(self.L.fga_dev_open(index, C.byref(self.h)), "open device")

    def sync(self):
        check(self.L.fga_dev_sync(self.h), "sync")

    def stage_ms(self, stage):
        return float(self.L.fga_dev_stage_ms(self.h, stage))

    def upload(self, gix):
        return DeviceGix(self, gix)

    def close(self):
        if self.h:
            self.L.fga_dev_close(self.h)
            self.h = C.c_void_p()


class DeviceGix:
    def __init__(self, dev, gix):
        self.dev = dev
        self.h = C.c_void_p()
        self.nents = gix.nents
        self.ebytes = gix.ebytes
        self.pbyte = gix.pbyte
        self.postbytes = gix.postbytes
        self.contbytes = gix.contbytes
        check(dev.L.fga_dgix_upload(dev.h, gix.h, C.byref(self.h)), "upload GIX")

    def free(self):
        if self.h:
            self.dev.L.fga_dgix_free(self.h)
            self.h = C.c_void_p()


def build_gix_device(dev, gdb, nthreads=8, host_copy=False, use_mask=False):
    """GDB -> index on the device (fga_dgix_build); returns (DeviceGix, Gix descriptor)."""
    from .gixio import Gix
    dh, xh = C.c_void_p(), C.c_void_p()
    flags = (1 if host_copy else 0) | (2 if use_mask else 0)
    check(dev.L.fga_dgix_build(dev.h, gdb.h, nthreads, flags, C.byref(dh), C.byref(xh)), "device GIX build")
    x = Gix("<device>", handle=xh)
    d = DeviceGix.__new__(DeviceGix)
    d.dev, d.h = dev, dh
    d.nents, d.ebytes, d.pbyte, d.postbytes, d.contbytes = x.nents, x.ebytes, x.pbyte, x.postbytes, x.contbytes
    return d, x


class Seeds:
    """device-resident seeds of one fga_seed_merge call."""

    def __init__(self, dev, h):
        self.dev = dev
        self.h = h
        self.count = dev.L.fga_seeds_count(h)
        self.plen_sum = dev.L.fga_seeds_plen_sum(h)

    def download(self):
        out = np.empty(self.count, dtype=SEED_DTYPE)
        check(self.dev.L.fga_seeds_download(self.h, out.ctypes.data_as(C.c_void_p), self.count), "download seeds")
        return out

    def free(self):
        if self.h:
            self.dev.L.fga_seeds_free(self.h)
            self.h = C.c_void_p()


def seed_merge(dev, t1, t2=None, freq=10, soft_mask=False, flip=False, prefix_begin=0, prefix_end=0,
               capacity=0):
    """Adaptive seed merge on the GPU (t2=None: self comparison)."""
    prm = MergeParams(freq, int(soft_mask), int(flip), prefix_begin, prefix_end)
    h = C.c_void_p()
    st = dev.L.fga_seed_merge(dev.h, t1.h, t2.h if t2 is not None else None, C.byref(prm), capacity, C.byref(h))
    if st == 2:
        need = dev.L.fga_seeds_count(h)
        dev.L.fga_seeds_free(h)
        return seed_merge(dev, t1, t2, freq, soft_mask, flip, prefix_begin, prefix_end, capacity=need + 1024)
    check(st, "seed merge")
    return Seeds(dev, h)


def seeds_to_reference_bytes(seeds, ipost, icont, jpost, jcont):
    """Re-encode fga_seed records as the reference's seed temp records (FastGA.c:961-966):
    u8 plen; A post (IPOST) + contig (ICONT); B post (JPOST) + contig|sign (JCONT), little endian.
    Returns (N stream bytes, C stream bytes)."""
    n = len(seeds)
    w = 1 + ipost + icont + jpost + jcont
    out = np.zeros((n, w), dtype=np.uint8)
    out[:, 0] = seeds["actg"] & 0xff
    col = 1
    for k in range(ipost):
        out[:, col] = (seeds["apos"] >> (8 * k)) & 0xff
        col += 1
    actg = seeds["actg"] >> 8
    for k in range(icont):
        out[:, col] = (actg >> (8 * k)) & 0xff
        col += 1
    for k in range(jpost):
        out[:, col] = (seeds["bpos"] >> (8 * k)) & 0xff
        col += 1
    bctg = (seeds["bctg"] & 0x3fffffff).astype(np.uint32)
    bsign = ((seeds["bctg"] >> 30) & 1).astype(np.uint32)
    bval = bctg | (bsign << np.uint32(8 * jcont - 1))
    for k in range(jcont):
        out[:, col] = (bval >> (8 * k)) & 0xff
        col += 1
    comp = (seeds["bctg"] >> 31).astype(bool)
    return out[~comp].tobytes(), out[comp].tobytes()


KEY_DTYPE = np.dtype([("lo", "<u8"), ("hi", "<u8")])


class Keys:
    """device-resident sorted diagonal records (128-bit keys)."""

    def __init__(self, dev, h):
        self.dev = dev
        self.h = h
        self.count = dev.L.fga_keys_count(h)
        w = [C.c_int() for _ in range(4)]
        dev.L.fga_keys_layout(h, *[C.byref(x) for x in w])
        self.wa, self.wb, self.wd, self.wt = [x.value for x in w]

    def download(self):
        out = np.empty(self.count, dtype=KEY_DTYPE)
        check(self.dev.L.fga_keys_download(self.h, out.ctypes.data_as(C.c_void_p), self.count), "download keys")
        return out

    def fields(self, keys=None):
        """dict of the packed fields as int64 arrays (host side decode, for tests)."""
        k = self.download() if keys is None else keys
        lo = k["lo"].astype(object)
        hi = k["hi"].astype(object)
        big = [(int(h) << 64) | int(l) for l, h in zip(lo, hi)]
        out = {}
        sh = 0
        for name, w in (("lcp", 6), ("drem", 6), ("anti", self.wt), ("bucket", self.wd),
                        ("bctg", self.wb), ("actg", self.wa), ("strand", 1)):
            m = (1 << w) - 1
            out[name] = np.array([(v >> sh) & m for v in big], dtype=np.int64)
            sh += w
        return out

    def free(self):
        if self.h:
            self.dev.L.fga_keys_free(self.h)
            self.h = C.c_void_p()


def seed_sort(dev, seeds, amxpos, bmxpos, nctg_a, nctg_b, anti_order_only=False):
    prm = SortParams(amxpos, bmxpos, nctg_a, nctg_b, int(anti_order_only))
    h = C.c_void_p()
    check(dev.L.fga_seed_sort(dev.h, seeds.h, C.byref(prm), C.byref(h)), "seed sort")
    return Keys(dev, h)


HIT_DTYPE = np.dtype([("dgmin", "<i4"), ("dgmax", "<i4"), ("alow", "<i8"), ("ahgh", "<i8"),
                      ("cov", "<i4"), ("pad", "<i4")])
UNIT_DTYPE = np.dtype([("actg", "<i4"), ("bctg", "<i4"), ("comp", "<i4"), ("nhits", "<i4"),
                       ("first_hit", "<i8"), ("bucket", "<i8")])
ALN_DTYPE = np.dtype([("tlen", "<i4"), ("diffs", "<i4"), ("abpos", "<i4"), ("bbpos", "<i4"),
                      ("aepos", "<i4"), ("bepos", "<i4"), ("flags", "<u4"), ("aread", "<i4"),
                      ("bread", "<i4"), ("unit", "<i4"), ("seq", "<i4"), ("pad", "<i4"), ("toff", "<i8")])


class HitList:
    """host-side hits + units (fga_hits)."""

    def __init__(self, L, ptr):
        self.L = L
        self.ptr = ptr
        self.nhits = L.fga_hits_count(ptr)
        self.nunits = L.fga_hits_nunits(ptr)

    @property
    def hits(self):
        if self.nhits == 0:
            return np.zeros(0, dtype=HIT_DTYPE)
        buf = (C.c_char * (self.nhits * HIT_DTYPE.itemsize)).from_address(self.L.fga_hits_array(self.ptr))
        return np.frombuffer(buf, dtype=HIT_DTYPE).copy()

    @property
    def units(self):
        if self.nunits == 0:
            return np.zeros(0, dtype=UNIT_DTYPE)
        buf = (C.c_char * (self.nunits * UNIT_DTYPE.itemsize)).from_address(self.L.fga_hits_units(self.ptr))
        return np.frombuffer(buf, dtype=UNIT_DTYPE).copy()

    def free(self):
        if self.ptr:
            self.L.fga_hits_free(self.ptr)
            self.ptr = None


def chain_scan(keys_host, layout, chain_break, chain_min, amxpos, bmxpos, alen_sorted, nthreads=8):
    """keys_host: structured (lo,hi) array; layout = (wa,wb,wd,wt); alen_sorted: A contig lengths by sorted index."""
    L = load_library()
    al = np.ascontiguousarray(alen_sorted, dtype=np.int64)
    prm = ChainParams(chain_break, chain_min, amxpos, bmxpos, al.ctypes.data, len(al))
    out = C.POINTER(Hits)()
    k = np.ascontiguousarray(keys_host)
    check(L.fga_chain_scan(k.ctypes.data_as(C.c_void_p), len(k), *layout, C.byref(prm), nthreads, C.byref(out)),
          "chain scan")
    return HitList(L, out)


def chain_scan_device(dev, keys, chain_break, chain_min, amxpos, bmxpos, alen_sorted):
    """The chain scan on the device-resident sorted keys (`keys`: a Keys object of seed_sort)."""
    al = np.ascontiguousarray(alen_sorted, dtype=np.int64)
    prm = ChainParams(chain_break, chain_min, amxpos, bmxpos, al.ctypes.data, len(al))
    out = C.POINTER(Hits)()
    check(dev.L.fga_chain_scan_device(dev.h, keys.h, C.byref(prm), C.byref(out)), "device chain scan")
    return HitList(dev.L, out)


def hits_from_arrays(units, hits):
    L = load_library()
    u = np.ascontiguousarray(units, dtype=UNIT_DTYPE)
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    out = C.POINTER(Hits)()
    check(L.fga_hits_create(u.ctypes.data_as(C.c_void_p), len(u), h.ctypes.data_as(C.c_void_p), len(h),
                            C.byref(out)), "hits create")
    return HitList(L, out)


def align_spec(ave_corr, tspace, freq):
    L = load_library()
    table = np.zeros(32768, dtype=np.int16)
    score = np.zeros(32768, dtype=np.int16)
    pa = C.c_int()
    f = (C.c_float * 4)(*freq)
    check(L.fga_align_spec(ave_corr, tspace, f, C.byref(pa), table.ctypes.data_as(C.c_void_p),
                           score.ctypes.data_as(C.c_void_p)), "align spec")
    return pa.value, table, score


class DeviceGenome:
    def __init__(self, dev, gdb, perm, want_revcomp):
        self.dev = dev
        self.h = C.c_void_p()
        p = np.ascontiguousarray(perm, dtype=np.int32)
        check(dev.L.fga_dgenome_upload(dev.h, gdb.h, p.ctypes.data_as(C.c_void_p), len(p), int(want_revcomp),
                                       C.byref(self.h)), "upload genome")

    def free(self):
        if self.h:
            self.dev.L.fga_dgenome_free(self.h)
            self.h = C.c_void_p()


def extend(dev, ga, gb, hitlist, path_ave, table, score, tspace=100, self_cmp=False, aln_min=50,
           aln_rate=0.35, cell_cap=0):
    prm = ExtendParams(tspace, path_ave, table.ctypes.data, score.ctypes.data, int(self_cmp), aln_min,
                       aln_rate, cell_cap, 0, 0)
    out = C.POINTER(Alns)()
    check(dev.L.fga_extend(dev.h, ga.h, gb.h, hitlist.ptr, C.byref(prm), C.byref(out)), "extend")
    a = out.contents
    n, nt = a.naln, a.ntrace
    alns = np.frombuffer((C.c_char * (n * ALN_DTYPE.itemsize)).from_address(a.alns), dtype=ALN_DTYPE).copy() \
        if n > 0 else np.zeros(0, dtype=ALN_DTYPE)
    tb = np.frombuffer((C.c_char * nt).from_address(a.tbytes), dtype=np.uint8).copy() if nt > 0 \
        else np.zeros(0, dtype=np.uint8)
    stats = {"calls": a.ncalls, "waves": a.nwaves}
    dev.L.fga_alns_free(out)
    return alns, tb, stats


def trace_pts(dev, ga, gb, alns, tb, tspace=100, self_cmp=False, regrouped=False):
    """fga_trace_pts: Compute_Trace_PTS (align.c:6171, GREEDIEST) of every alignment of a set on the device.
    alns: ALN_DTYPE records, tb: their trace bytes.  Returns (toff[n+1], tlen[n], diffs[n], ints, info).
    regrouped: fga_trace_pts_regrouped (Gap_Improver applied on the device), info["resume"] = where the host has to go on
    per alignment (-1: nowhere)."""
    from .lib import Traces
    alns = np.ascontiguousarray(alns)
    tb = np.ascontiguousarray(tb, dtype=np.uint8)
    a = Alns(len(alns), len(tb), 0, 0, alns.ctypes.data, tb.ctypes.data)
    out = C.POINTER(Traces)()
    fn = dev.L.fga_trace_pts_regrouped if regrouped else dev.L.fga_trace_pts
    check(fn(dev.h, ga.h, gb.h, C.byref(a), tspace, int(self_cmp), C.byref(out)), "trace_pts")
    t = out.contents
    n, nt = t.naln, t.ntrace

    def take(ptr, count, dt):
        if count == 0:
            return np.zeros(0, dtype=dt)
        return np.frombuffer((C.c_char * (count * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt).copy()
    res = (take(t.toff, n + 1 if n else 0, np.int64), take(t.tlen, n, np.int32), take(t.diffs, n, np.int32),
           take(t.trace, nt, np.int32), {"panels": t.npanels})
    if regrouped:
        res[4]["resume"] = take(t.resume, n, np.int32)
    dev.L.fga_traces_free(out)
    return res


def run(root1, root2=None, out_path=None, device=0, freq=10, soft_mask=False, symmetric=False, chain_break=1000,
        chain_min=85, align_min=100, identity=0.7, nthreads=8, command_line="FastGA", paf_path=None, paf_flags=0,
        pass_seeds=0, reference_threads=0, build_index=False, masks1=None, masks2=None):
    """The whole hot path (fga_run): prebuilt GDB/GIX roots in, .1aln (and PAF) out.  Returns the stats as a dict.
    reference_threads = n > 0: records that tie on (aread, abpos) in the order `FastGA -T<n>` writes them."""
    from .lib import RunParams, RunStats
    L = load_library()
    def cstrs(paths):
        if not paths:
            return None, 0
        return (C.c_char_p * len(paths))(*[p.encode() for p in paths]), len(paths)
    m1, n1 = cstrs(masks1)
    m2, n2 = cstrs(masks2)
    prm = RunParams(device, freq, int(soft_mask), int(symmetric), 2 * chain_break, 2 * chain_min, align_min,
                    1.0 - identity, nthreads, out_path.encode() if out_path else None, command_line.encode(),
                    paf_path.encode() if paf_path else None, paf_flags, pass_seeds, int(build_index), m1, n1, m2, n2,
                    reference_threads)
    st = RunStats()
    check(L.fga_run(root1.encode(), root2.encode() if root2 else None, C.byref(prm), C.byref(st)), "fga_run")
    return {n: getattr(st, n) for n, _ in RunStats._fields_}


def run_multi(root1, root2=None, out_path=None, devices=(0,), freq=10, soft_mask=False, symmetric=False,
              chain_break=1000, chain_min=85, align_min=100, identity=0.7, nthreads=8, command_line="FastGA",
              paf_path=None, paf_flags=0, reference_threads=0, build_index=False, masks1=None, masks2=None):
    """fga_run_multi: ONE comparison cut over `devices` (HIP device numbers; they may repeat -- ranks sharing a GPU) from this
    one process: one host thread + stream per device, seeds exchanged by A-contig part with hipMemcpyPeerAsync, one .1aln.
    The result does not depend on the number of devices.  Returns the stats as a dict."""
    from .lib import RunParams, RunStats
    L = load_library()
    def cstrs(paths):
        if not paths:
            return None, 0
        return (C.c_char_p * len(paths))(*[p.encode() for p in paths]), len(paths)
    m1, n1 = cstrs(masks1)
    m2, n2 = cstrs(masks2)
    prm = RunParams(0, freq, int(soft_mask), int(symmetric), 2 * chain_break, 2 * chain_min, align_min,
                    1.0 - identity, nthreads, out_path.encode() if out_path else None, command_line.encode(),
                    paf_path.encode() if paf_path else None, paf_flags, 0, int(build_index), m1, n1, m2, n2,
                    reference_threads)
    devs = (C.c_int * max(len(devices), 1))(*[int(d) for d in devices])
    st = RunStats()
    check(L.fga_run_multi(root1.encode(), root2.encode() if root2 else None, C.byref(prm), len(devices), devs,
                          C.byref(st)), "fga_run_multi")
    return {n: getattr(st, n) for n, _ in RunStats._fields_}


class Multi:
    """fga_multi_open / fga_multi_run / fga_multi_close: ONE comparison at a time over `devices` from this one process, the
    inputs resident between the runs (every rank's slice of both tables on its device, one host thread per device)."""

    def __init__(self, root1, root2=None, devices=(0,), nthreads=8, build_index=False, masks1=None, masks2=None):
        from .lib import RunParams
        self.L = load_library()
        self.devices = tuple(int(d) for d in devices)
        def cstrs(paths):
            if not paths:
                return None, 0
            return (C.c_char_p * len(paths))(*[p.encode() for p in paths]), len(paths)
        m1, n1 = cstrs(masks1)
        m2, n2 = cstrs(masks2)
        prm = RunParams(0, 10, 0, 0, 2000, 170, 100, 0.3, nthreads, None, b"FastGA", None, 0, 0, int(build_index), m1, n1, m2, n2, 0)
        devs = (C.c_int * max(len(self.devices), 1))(*self.devices)
        h = C.c_void_p()
        check(self.L.fga_multi_open(root1.encode(), root2.encode() if root2 else None, C.byref(prm), len(self.devices), devs,
                                    C.byref(h)), "fga_multi_open")
        self.h = h
        tb, sb, b1, b2 = C.c_int64(), C.c_int(), C.c_int64(), C.c_int64()
        check(self.L.fga_multi_info(h, C.byref(tb), C.byref(sb), C.byref(b1), C.byref(b2)), "fga_multi_info")
        self.table_bytes, self.seed_bytes, self.bases = tb.value, sb.value, (b1.value, b2.value)

    def run(self, out_path=None, freq=10, soft_mask=False, symmetric=False, chain_break=1000, chain_min=85, align_min=100,
            identity=0.7, nthreads=8, command_line="FastGA", paf_path=None, paf_flags=0, reference_threads=0):
        from .lib import RunParams, RunStats
        prm = RunParams(0, freq, int(soft_mask), int(symmetric), 2 * chain_break, 2 * chain_min, align_min,
                        1.0 - identity, nthreads, out_path.encode() if out_path else None, command_line.encode(),
                        paf_path.encode() if paf_path else None, paf_flags, 0, 0, None, 0, None, 0, reference_threads)
        st = RunStats()
        check(self.L.fga_multi_run(self.h, C.byref(prm), C.byref(st)), "fga_multi_run")
        return {n: getattr(st, n) for n, _ in RunStats._fields_}

    def rank_stats(self):
        """per rank of the last run: seconds of phase 1 / exchange / phase 2 (+ filter), extension kernel ms, wave steps"""
        out = []
        if len(self.devices) < 2:
            return out
        for r in range(len(self.devices)):
            s3, ms, ws = (C.c_double * 3)(), C.c_double(), C.c_int64()
            check(self.L.fga_multi_rank_stats(self.h, r, s3, C.byref(ms), C.byref(ws)), "fga_multi_rank_stats")
            out.append({"phase1_s": s3[0], "exchange_s": s3[1], "phase2_s": s3[2], "extend_kernel_ms": ms.value,
                        "wave_steps": ws.value})
        return out

    def close(self):
        if self.h:
            self.L.fga_multi_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session:
    """inputs resident in HBM; run() is one pass of the hot path (fga_session_*)."""

    def __init__(self, root1, root2=None, device=0, rank=0, nranks=1, nthreads=8, build_index=False):
        """nranks > 1: the session of rank `rank` of one comparison over nranks GPUs -- it holds only its own 12-mer prefix
        range of both tables (fga_session_open_sliced)"""
        self.L = load_library()
        self.h = C.c_void_p()
        if nranks > 1:
            check(self.L.fga_session_open_sliced(root1.encode(), root2.encode() if root2 else None, device, nthreads,
                                                 rank, nranks, C.byref(self.h)), "open sliced session")
        else:
            check(self.L.fga_session_open_flags(root1.encode(), root2.encode() if root2 else None, device, nthreads,
                                                1 if build_index else 0, C.byref(self.h)), "open session")
        self.rank, self.nranks = rank, nranks
        self.table_bytes = self.L.fga_session_table_bytes(self.h)
        self.seed_bytes = self.L.fga_session_seed_bytes(self.h)
        self.bases = (self.L.fga_session_bases(self.h, 0), self.L.fga_session_bases(self.h, 1))

    def run(self, out_path=None, freq=10, soft_mask=False, symmetric=False, chain_break=1000, chain_min=85,
            align_min=100, identity=0.7, nthreads=8, command_line="FastGA", paf_path=None, paf_flags=0,
            pass_seeds=0, reference_threads=0, build_index=False):
        from .lib import RunParams, RunStats
        prm = RunParams(0, freq, int(soft_mask), int(symmetric), 2 * chain_break, 2 * chain_min, align_min,
                        1.0 - identity, nthreads, out_path.encode() if out_path else None, command_line.encode(),
                        paf_path.encode() if paf_path else None, paf_flags, pass_seeds, int(build_index), None, 0, None, 0,
                    reference_threads)
        st = RunStats()
        check(self.L.fga_session_run(self.h, C.byref(prm), C.byref(st)), "session run")
        return {n: getattr(st, n) for n, _ in RunStats._fields_}

    # ---- the three stages of run(), for one comparison cut into A-contig parts (fastga_amd/parallel.py) ----
    def params(self, out_path=None, freq=10, soft_mask=False, symmetric=False, chain_break=1000, chain_min=85,
               align_min=100, identity=0.7, nthreads=8, command_line="FastGA", paf_path=None, paf_flags=0,
               pass_seeds=0, reference_threads=0, build_index=False):
        from .lib import RunParams
        return RunParams(0, freq, int(soft_mask), int(symmetric), 2 * chain_break, 2 * chain_min, align_min,
                         1.0 - identity, nthreads, out_path.encode() if out_path else None, command_line.encode(),
                         paf_path.encode() if paf_path else None, paf_flags, pass_seeds, int(build_index), None, 0, None, 0,
                    reference_threads)

    def new_stats(self):
        from .lib import RunStats
        return RunStats()

    @staticmethod
    def stats_dict(st):
        from .lib import RunStats
        return {n: getattr(st, n) for n, _ in RunStats._fields_}

    @property
    def nctg(self):
        return self.L.fga_session_nctg(self.h)

    def contig_perm(self):
        """original index of every A contig of the (length-sorted) index order"""
        p = self.L.fga_session_contig_perm(self.h)
        return np.array([p[j] for j in range(self.nctg)], dtype=np.int32)

    def merge(self, prm, stats, prefix_begin=0, prefix_end=0):
        """phase 1 over a 12-mer prefix range -> Seeds (device resident)"""
        h = C.c_void_p()
        check(self.L.fga_session_merge(self.h, C.byref(prm), prefix_begin, prefix_end, C.byref(h), C.byref(stats)),
              "session merge")
        return Seeds(self.dev_wrapper(), h)

    def strand_counts(self):
        """seeds per (strand, A contig) the merges of this session have counted (reference order only)"""
        cnt = np.zeros(2 * self.nctg, dtype=np.int64)
        check(self.L.fga_session_strand_counts(self.h, cnt.ctypes.data_as(C.POINTER(C.c_int64))), "strand counts")
        return cnt

    def set_strand_counts(self, cnt):
        cnt = np.ascontiguousarray(cnt, dtype=np.int64)
        assert len(cnt) == 2 * self.nctg
        check(self.L.fga_session_set_strand_counts(self.h, cnt.ctypes.data_as(C.POINTER(C.c_int64))), "strand counts")

    def clear_strand_counts(self):
        self.L.fga_session_clear_strand_counts(self.h)

    def contig_histogram(self, seeds):
        cnt = np.zeros(self.nctg, dtype=np.int64)
        check(self.L.fga_seeds_contig_histogram(self.L.fga_session_device(self.h), seeds.h, self.nctg,
                                                cnt.ctypes.data_as(C.POINTER(C.c_int64))), "contig histogram")
        return cnt

    def split_to(self, seeds, select, nparts, dst_device_ptr):
        """regroup `seeds` by select[A contig] into the device buffer at dst_device_ptr; returns part offsets"""
        sel = np.ascontiguousarray(select, dtype=np.int32)
        off = np.zeros(nparts + 1, dtype=np.int64)
        check(self.L.fga_seeds_split_to(self.L.fga_session_device(self.h), seeds.h,
                                        sel.ctypes.data_as(C.POINTER(C.c_int)), len(sel), nparts,
                                        C.c_void_p(dst_device_ptr), off.ctypes.data_as(C.POINTER(C.c_int64))),
              "seed split")
        return off

    def import_seeds(self, pieces):
        """pieces: [(device pointer, record count)] -> Seeds"""
        n = len(pieces)
        ptrs = (C.c_void_p * max(n, 1))(*[C.c_void_p(p) for p, _ in pieces])
        cnts = (C.c_int64 * max(n, 1))(*[int(c) for _, c in pieces])
        h = C.c_void_p()
        check(self.L.fga_seeds_import(self.L.fga_session_device(self.h), ptrs, cnts, n, C.byref(h)), "seed import")
        return Seeds(self.dev_wrapper(), h)

    def align(self, prm, stats, seeds):
        """phase 2 on `seeds` (consumed) -> pointer to the raw fga_alns (free with free_alns)"""
        from .lib import Alns
        out = C.POINTER(Alns)()
        h, seeds.h = seeds.h, C.c_void_p()
        check(self.L.fga_session_align(self.h, C.byref(prm), h, C.byref(out), C.byref(stats)), "session align")
        return out

    def free_alns(self, a):
        self.L.fga_alns_free(a)

    def finish(self, prm, stats, raws):
        """redundancy filter + phase 3 over the record sets `raws` (POINTER(Alns) each)"""
        from .lib import Alns
        arr = (C.POINTER(Alns) * max(len(raws), 1))(*raws)
        check(self.L.fga_session_finish(self.h, C.byref(prm), arr, len(raws), C.byref(stats)), "session finish")

    def filter(self, raw, nthreads=8):
        """the redundancy filter on ONE part's raw records (every contig pair's records come from the part that owns the
        A contig) -> pointer to the filtered set in final order (free with free_alns)"""
        from .lib import Alns
        out = C.POINTER(Alns)()
        check(self.L.fga_filter_alignments_mt(raw, nthreads, C.byref(out)), "filter")
        return out

    def finish_filtered(self, prm, stats, filtered):
        """phase 3 over sets that were filtered part by part: merge by A contig, .1aln / PAF"""
        from .lib import Alns
        arr = (C.POINTER(Alns) * max(len(filtered), 1))(*filtered)
        check(self.L.fga_session_finish_filtered(self.h, C.byref(prm), arr, len(filtered), C.byref(stats)),
              "session finish (filtered sets)")

    def stream_open(self, prm):
        """fga_session_stream_open: the .1aln of this session's genomes as a stream (prm: out_path, command_line)"""
        h = C.c_void_p()
        check(self.L.fga_session_stream_open(self.h, C.byref(prm), C.byref(h)), "stream open")
        return h

    def reference_order(self, prm, filtered):
        """one part's filtered set into the reference's tie order (in place)"""
        check(self.L.fga_session_reference_order(self.h, C.byref(prm), filtered), "reference order")

    def sync(self):
        check(self.L.fga_dev_sync(self.L.fga_session_device(self.h)), "sync")

    def dev_malloc(self, nbytes):
        p = C.c_void_p()
        check(self.L.fga_dev_malloc(self.L.fga_session_device(self.h), max(int(nbytes), 16), C.byref(p)), "device malloc")
        return p.value

    def trim(self):
        """fga_dev_trim: regions of the device pool that nothing uses go back to the driver"""
        self.L.fga_dev_trim(self.L.fga_session_device(self.h))

    def dev_free(self, ptr):
        self.L.fga_dev_free(self.L.fga_session_device(self.h), C.c_void_p(ptr))

    def dev_download(self, ptr, nbytes):
        out = np.empty(int(nbytes), dtype=np.uint8)
        check(self.L.fga_dev_download(self.L.fga_session_device(self.h), out.ctypes.data_as(C.c_void_p),
                                      C.c_void_p(ptr), int(nbytes)), "device download")
        return out

    def dev_upload(self, ptr, arr):
        a = np.ascontiguousarray(arr)
        check(self.L.fga_dev_upload(self.L.fga_session_device(self.h), C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p),
                                    a.nbytes), "device upload")

    def dev_wrapper(self):
        """the session's device context as a (non-owning) Device object"""
        d = Device.__new__(Device)
        d.L = self.L
        d.h = C.c_void_p(self.L.fga_session_device(self.h))
        d.close = lambda: None
        return d

    def close(self):
        if self.h:
            self.L.fga_session_close(self.h)
            self.h = C.c_void_p()
