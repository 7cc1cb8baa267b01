"""oracle/harness.py -- TEST INFRASTRUCTURE: drive the real reference binaries in oracle/_ref and the CPU
restatement in oracle/liboracle.so.  Imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product (fastga_amd/)."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")


def ref_bin(name):
    return os.path.join(REF, name)


def have_reference():
    return all(os.path.exists(ref_bin(b)) for b in ("FAtoGDB", "GIXmake", "FastGA", "ONEview"))


def run(cmd, cwd=None, env=None, check=True):
    e = dict(os.environ)
    e["PATH"] = REF + os.pathsep + e.get("PATH", "")
    if env:
        e.update(env)
    r = subprocess.run(cmd, cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if check and r.returncode != 0:
        raise RuntimeError(f"{cmd} failed ({r.returncode}):\n{r.stdout}\n{r.stderr}")
    return r


def ref_build_index(fasta, workdir, threads=8):
    """reference FAtoGDB + GIXmake on a FASTA; returns the root path (workdir/<root>)."""
    root = os.path.splitext(os.path.basename(fasta))[0]
    run([ref_bin("FAtoGDB"), fasta, os.path.join(workdir, root + ".1gdb")], cwd=workdir)
    run([ref_bin("GIXmake"), f"-T{threads}", f"-P{workdir}", os.path.join(workdir, root)], cwd=workdir)
    return os.path.join(workdir, root)


def ref_fastga(a, b, workdir, out, threads=8, flags=(), capture_seeds=False):
    """Run reference FastGA -1:<out> a [b]; with capture_seeds the seed temp files survive (unlink shim)
    and are returned as (N bytes, C bytes)."""
    env = {}
    if capture_seeds:
        env["LD_PRELOAD"] = os.path.join(REF, "unlink_shim.so")
        for f in glob.glob(os.path.join(workdir, "_pair.*")):
            os.remove(f)
    cmd = [ref_bin("FastGA"), "-v", "-k", f"-T{threads}", f"-P{workdir}", f"-1:{out}", *flags, a]
    if b is not None:
        cmd.append(b)
    r = run(cmd, cwd=workdir, env=env)
    seeds = None
    if capture_seeds:
        nb, cb = [], []
        for f in sorted(glob.glob(os.path.join(workdir, "_pair.*"))):
            (nb if f.endswith(".N") else cb).append(open(f, "rb").read())
            os.remove(f)
        seeds = (b"".join(nb), b"".join(cb))
    return r, seeds


def oneview(path, strip_provenance=True):
    r = run([ref_bin("ONEview"), path])
    lines = r.stdout.splitlines()
    if strip_provenance:
        lines = [ln for ln in lines if not ln.startswith(("!", "<"))]
    return lines


# ------------------------------------------------------------------ CPU restatement (liboracle.so)

class _Seeds(C.Structure):
    _fields_ = [("nbuf", C.POINTER(C.c_uint8)), ("nlen", C.c_int64),
                ("cbuf", C.POINTER(C.c_uint8)), ("clen", C.c_int64),
                ("nhits", C.c_int64), ("tseed", C.c_int64)]


_ORACLE = None


def oracle_lib():
    global _ORACLE
    if _ORACLE is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", HERE, "port"], check=True, stdout=subprocess.DEVNULL)
        _ORACLE = C.CDLL(path)
    return _ORACLE


def _take(ptr, n):
    if n == 0:
        return b""
    return bytes(np.ctypeslib.as_array(ptr, shape=(n,)))


def oracle_seed_merge(tab1, idx1, pbyte1, tab2, idx2, pbyte2, freq=10, soft_mask=False, flip=False,
                      pfirst=0, plast=1 << 24):
    """tab*: uint8 numpy arrays of raw entries; idx*: int64[2^24].  Returns (N bytes, C bytes, nhits, tseed)."""
    L = oracle_lib()
    o = _Seeds()
    t1 = np.ascontiguousarray(tab1, dtype=np.uint8)
    t2 = np.ascontiguousarray(tab2, dtype=np.uint8)
    i1 = np.ascontiguousarray(idx1, dtype=np.int64)
    i2 = np.ascontiguousarray(idx2, dtype=np.int64)
    st = L.oracle_seed_merge(t1.ctypes.data_as(C.c_void_p), i1.ctypes.data_as(C.c_void_p), C.c_int(pbyte1),
                             t2.ctypes.data_as(C.c_void_p), i2.ctypes.data_as(C.c_void_p), C.c_int(pbyte2),
                             C.c_int(freq), C.c_int(int(soft_mask)), C.c_int(int(flip)),
                             C.c_int64(pfirst), C.c_int64(plast), C.byref(o))
    if st != 0:
        raise MemoryError("oracle_seed_merge")
    res = (_take(o.nbuf, o.nlen), _take(o.cbuf, o.clen), o.nhits, o.tseed)
    L.oracle_free(C.byref(o))
    return res


def oracle_self_seed_merge(tab, idx, pbyte, freq=10, soft_mask=False, pfirst=0, plast=1 << 24):
    L = oracle_lib()
    o = _Seeds()
    t = np.ascontiguousarray(tab, dtype=np.uint8)
    i = np.ascontiguousarray(idx, dtype=np.int64)
    st = L.oracle_self_seed_merge(t.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p), C.c_int(pbyte),
                                  C.c_int(freq), C.c_int(int(soft_mask)), C.c_int64(pfirst), C.c_int64(plast),
                                  C.byref(o))
    if st != 0:
        raise MemoryError("oracle_self_seed_merge")
    res = (_take(o.nbuf, o.nlen), _take(o.cbuf, o.clen), o.nhits, o.tseed)
    L.oracle_free(C.byref(o))
    return res


def sorted_records(buf, width):
    """canonical multiset form of a packed record stream: rows sorted lexicographically."""
    a = np.frombuffer(buf, dtype=np.uint8)
    if a.size == 0:
        return a.reshape(0, width)
    a = a.reshape(-1, width)
    order = np.lexsort(a.T[::-1])
    return a[order]


# ------------------------------------------------------------------ Local_Alignment: oracle port and real reference

class _OSpec(C.Structure):
    _fields_ = [("ave_path", C.c_int), ("tspace", C.c_int), ("reach", C.c_int),
                ("score", C.c_int16 * 32768), ("table", C.c_int16 * 32768)]


class _OPath(C.Structure):
    _fields_ = [("abpos", C.c_int), ("bbpos", C.c_int), ("aepos", C.c_int), ("bepos", C.c_int),
                ("diffs", C.c_int), ("tlen", C.c_int)]


def oracle_spec(ave_corr=0.7, tspace=100, freq=(0.25, 0.25, 0.25, 0.25), reach=0):
    L = oracle_lib()
    sp = _OSpec()
    f = (C.c_float * 4)(*freq)
    L.oracle_align_spec(C.c_double(ave_corr), C.c_int(tspace), f, C.c_int(reach), C.byref(sp))
    return sp


def pad_seq(s):
    """numeric sequence with the sentinel 4 either side; returns (buffer, offset-1 view pointer helper)"""
    buf = np.empty(len(s) + 2, dtype=np.uint8)
    buf[0] = 4
    buf[1:-1] = s
    buf[-1] = 4
    return buf


def oracle_local_alignment(abuf, bbuf, spec, low, hgh, anti, lbord=-1, hbord=-1, acomp=False, selfie=False):
    """abuf/bbuf: padded buffers from pad_seq.  Returns (abpos,bbpos,aepos,bepos,diffs,trace uint16 array)."""
    L = oracle_lib()
    alen, blen = len(abuf) - 2, len(bbuf) - 2
    out = _OPath()
    trace = np.zeros(4 * (alen // spec.tspace + 2) + 8, dtype=np.uint16)
    st = L.oracle_local_alignment(C.c_void_p(abuf.ctypes.data + 1), C.c_int(alen),
                                  C.c_void_p(bbuf.ctypes.data + 1), C.c_int(blen),
                                  C.c_int(int(acomp)), C.c_int(int(selfie)), C.byref(spec),
                                  C.c_int(low), C.c_int(hgh), C.c_int(anti), C.c_int(lbord), C.c_int(hbord),
                                  C.byref(out), trace.ctypes.data_as(C.c_void_p))
    if st != 0:
        raise RuntimeError(f"oracle_local_alignment status {st}")
    return (out.abpos, out.bbpos, out.aepos, out.bepos, out.diffs, trace[:out.tlen].copy())


class _RPath(C.Structure):
    _fields_ = [("trace", C.c_void_p), ("tlen", C.c_int), ("diffs", C.c_int),
                ("abpos", C.c_int), ("bbpos", C.c_int), ("aepos", C.c_int), ("bepos", C.c_int)]


class _RAlign(C.Structure):
    _fields_ = [("path", C.POINTER(_RPath)), ("flags", C.c_uint32), ("aseq", C.c_void_p), ("bseq", C.c_void_p),
                ("alen", C.c_int), ("blen", C.c_int)]


_REFALIGN = None


def ref_align_lib():
    global _REFALIGN
    if _REFALIGN is None:
        L = C.CDLL(os.path.join(REF, "libalign_ref.so"))
        L.New_Work_Data.restype = C.c_void_p
        L.New_Align_Spec.restype = C.c_void_p
        L.New_Align_Spec.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_float), C.c_int]
        L.Local_Alignment.argtypes = [C.POINTER(_RAlign), C.c_void_p, C.c_void_p] + [C.c_int] * 5
        L.Free_Work_Data.argtypes = [C.c_void_p]
        L.Free_Align_Spec.argtypes = [C.c_void_p]
        _REFALIGN = L
    return _REFALIGN


class RefAligner:
    """the real reference's Local_Alignment (align.c:1423) through ctypes."""

    def __init__(self, ave_corr=0.7, tspace=100, freq=(0.25, 0.25, 0.25, 0.25), reach=0):
        self.L = ref_align_lib()
        f = (C.c_float * 4)(*freq)
        self.spec = self.L.New_Align_Spec(ave_corr, tspace, f, reach)
        self.work = self.L.New_Work_Data()

    def align(self, abuf, bbuf, low, hgh, anti, lbord=-1, hbord=-1, acomp=False, selfie=False):
        path = _RPath()
        al = _RAlign()
        al.path = C.pointer(path)
        al.flags = 2 if acomp else 0
        al.aseq = abuf.ctypes.data + 1
        al.bseq = (abuf.ctypes.data + 1) if selfie else (bbuf.ctypes.data + 1)
        al.alen = len(abuf) - 2
        al.blen = len(bbuf) - 2
        st = self.L.Local_Alignment(C.byref(al), self.work, self.spec, low, hgh, anti, lbord, hbord)
        if st != 0:
            raise RuntimeError("reference Local_Alignment failed")
        n = path.tlen
        tr = np.ctypeslib.as_array(C.cast(path.trace, C.POINTER(C.c_uint16)), shape=(max(n, 1),))[:n].copy()
        return (path.abpos, path.bbpos, path.aepos, path.bepos, path.diffs, tr)

    def trace_pts(self, abuf, bbuf, path, tspace=100, selfie=False, improve=False):
        """the real reference's Compute_Trace_PTS (align.c:6171) in GREEDIEST mode on an alignment given as
        (abpos, bbpos, aepos, bepos, diffs, uint16 trace points); returns (diffs, int32 edit trace)"""
        abpos, bbpos, aepos, bepos, diffs, tr = path
        pts = np.ascontiguousarray(tr, dtype=np.uint16).copy()
        rp = _RPath()
        rp.trace = pts.ctypes.data
        rp.tlen, rp.diffs = len(pts), diffs
        rp.abpos, rp.bbpos, rp.aepos, rp.bepos = abpos, bbpos, aepos, bepos
        al = _RAlign()
        al.path = C.pointer(rp)
        al.flags = 0
        al.aseq = abuf.ctypes.data + 1
        al.bseq = (abuf.ctypes.data + 1) if selfie else (bbuf.ctypes.data + 1)
        al.alen = len(abuf) - 2
        al.blen = (len(abuf) if selfie else len(bbuf)) - 2
        self.L.Compute_Trace_PTS.argtypes = [C.POINTER(_RAlign), C.c_void_p] + [C.c_int] * 4
        st = self.L.Compute_Trace_PTS(C.byref(al), self.work, tspace, 0, 1, -1)
        if st != 0:
            raise RuntimeError("reference Compute_Trace_PTS failed")
        if improve:                                   # Gap_Improver (align.c:6714), as every reader calls it next
            self.L.Gap_Improver.argtypes = [C.POINTER(_RAlign), C.c_void_p]
            if self.L.Gap_Improver(C.byref(al), self.work) != 0:
                raise RuntimeError("reference Gap_Improver failed")
        n = rp.tlen
        out = np.ctypeslib.as_array(C.cast(rp.trace, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
        return rp.diffs, out

    def close(self):
        self.L.Free_Work_Data(self.work)
        self.L.Free_Align_Spec(self.spec)


def oracle_trace_pts(abuf, bbuf, path, tspace=100, selfie=False):
    """oracle/trace_oracle.c: (diffs, int32 edit trace) of an alignment given by its trace points"""
    L = oracle_lib()
    abpos, bbpos, aepos, bepos, diffs, tr = path
    pts = np.ascontiguousarray(tr, dtype=np.uint16)
    cap = int(pts[0::2].sum()) + 16
    out = np.zeros(cap, dtype=np.int32)
    n, d = C.c_int(), C.c_int()
    b = abuf if selfie else bbuf
    st = L.oracle_trace_pts(C.c_void_p(abuf.ctypes.data + 1), C.c_int(len(abuf) - 2),
                            C.c_void_p(b.ctypes.data + 1), C.c_int(len(b) - 2), C.c_int(int(selfie)),
                            C.c_int(abpos), C.c_int(bbpos), C.c_int(aepos), C.c_int(bepos),
                            C.c_void_p(pts.ctypes.data), C.c_int(len(pts)), C.c_int(tspace),
                            C.c_void_p(out.ctypes.data), C.byref(n), C.byref(d))
    if st != 0:
        raise RuntimeError("oracle_trace_pts failed")
    return d.value, out[:n.value].copy()


def oracle_gap_improver(abuf, bbuf, path, trace, diffs, selfie=False):
    """oracle/gap_oracle.c on the int trace of Compute_Trace_PTS: (diffs, rewritten trace)"""
    L = oracle_lib()
    t = np.ascontiguousarray(trace, dtype=np.int32).copy()
    b = abuf if selfie else bbuf
    d = C.c_int(diffs)
    st = L.oracle_gap_improver(C.c_void_p(abuf.ctypes.data + 1), C.c_int(len(abuf) - 2),
                               C.c_void_p(b.ctypes.data + 1), C.c_int(len(b) - 2),
                               C.c_int(path[0]), C.c_int(path[1]), C.c_void_p(t.ctypes.data), C.c_int(len(t)),
                               C.byref(d))
    if st != 0:
        raise RuntimeError("oracle_gap_improver failed")
    return d.value, t


# ------------------------------------------------------------------ chain scan: oracle port and the reference's hit boxes

def ref_hit_boxes(a, b, workdir, threads=4, flags=()):
    """Run the DEBUG_HIT build of the reference (oracle/_ref/FastGA_hits, oracle/Makefile) and parse what it prints per
    chain that passes the coverage test (FastGA.c:3165-3225): rows (ctg1, ctg2, bucket, aux, cov, dgmin, dgmax, alow,
    ahgh) with ctg1/ctg2 the ORIGINAL contig indices, in print order (N pass before C pass inside every A part)."""
    import re
    exe = ref_bin("FastGA_hits")
    cmd = [exe, "-k", f"-T{threads}", f"-P{workdir}", "-1:" + os.path.join(workdir, "_hits_out"), *flags, a]
    if b is not None:
        cmd.append(b)
    r = run(cmd, cwd=workdir)
    rows, c1, c2, cur = [], -1, -1, None
    pc = re.compile(r"^\s+Contig (\d+) vs Contig (\d+)")
    ph = re.compile(r"^Hit on bucket (-?\d+)(\+1)? Coverage = (-?\d+)")
    pb = re.compile(r"^\s+Box:\s+Diag = (-?\d+):(-?\d+)\s+Anti = (-?\d+):(-?\d+):(-?\d+)")
    for ln in r.stdout.splitlines():
        m = pc.match(ln)
        if m:
            c1, c2 = int(m.group(1)), int(m.group(2))
            continue
        m = ph.match(ln)
        if m:
            cur = (int(m.group(1)), 1 if m.group(2) else 0, int(m.group(3)))
            continue
        m = pb.match(ln)
        if m and cur is not None:
            rows.append((c1, c2, cur[0], cur[1], cur[2], int(m.group(1)), int(m.group(2)), int(m.group(3)),
                         int(m.group(5))))
            cur = None
    return rows


def records_from_seed_bytes(nbytes, cbytes, ipost, icont, jpost, jcont, amxpos, bmxpos):
    """The reference's seed temp records (FastGA.c:961-966) -> the fields of its sort record (reimport_thread,
    FastGA.c:2703-2721) in its sort order (rmsd_sort: jcont, diag>>6, anti, diag&63, lcp inside an A contig; A contigs
    and the two strands are separate panels): dict of int64 arrays strand, actg, bctg, bucket, anti, drem, lcp."""
    w = 1 + ipost + icont + jpost + jcont
    cols = {k: [] for k in ("strand", "actg", "bctg", "bucket", "anti", "drem", "lcp")}
    for comp, buf in ((0, nbytes), (1, cbytes)):
        a = np.frombuffer(buf, dtype=np.uint8).reshape(-1, w).astype(np.int64)

        def le(c0, nb):
            v = np.zeros(len(a), dtype=np.int64)
            for k in range(nb):
                v |= a[:, c0 + k] << (8 * k)
            return v
        lcp = a[:, 0]
        i = le(1, ipost)
        actg = le(1 + ipost, icont)
        j = le(1 + ipost + icont, jpost)
        bc = le(1 + ipost + icont + jpost, jcont)
        bctg = bc & ((1 << (8 * jcont - 1)) - 1)              # bit 7 of the last byte is the B entry's own sign
        if comp:
            diag = (amxpos + bmxpos) - (i + j)
            anti = amxpos - (i - j)
        else:
            diag = bmxpos + (i - j)
            anti = i + j
        for k, v in (("strand", np.full(len(a), comp, dtype=np.int64)), ("actg", actg), ("bctg", bctg),
                     ("bucket", diag >> 6), ("anti", anti), ("drem", diag & 63), ("lcp", lcp)):
            cols[k].append(v)
    f = {k: np.concatenate(v) for k, v in cols.items()}
    order = np.lexsort((f["lcp"], f["drem"], f["anti"], f["bucket"], f["bctg"], f["actg"], f["strand"]))
    return {k: np.ascontiguousarray(v[order]) for k, v in f.items()}


def oracle_chain_scan(f, chain_break, chain_min, amxpos, bmxpos, alen_sorted):
    """oracle/chain_oracle.c over sorted record fields (records_from_seed_bytes / decoded device keys).  Returns an
    int64 array of rows (strand, actg, bctg, bucket, aux, cov, dgmin, dgmax, alow, ahgh), contig indices SORTED."""
    L = oracle_lib()
    n = len(f["anti"])
    arrs = [np.ascontiguousarray(f[k], dtype=np.int64) for k in ("strand", "actg", "bctg", "bucket", "anti", "drem",
                                                                  "lcp")]
    al = np.ascontiguousarray(alen_sorted, dtype=np.int64)
    out = C.POINTER(C.c_int64)()
    L.oracle_chain_scan.restype = C.c_int64
    nh = L.oracle_chain_scan(C.c_int64(n), *[x.ctypes.data_as(C.c_void_p) for x in arrs], C.c_int64(chain_break),
                             C.c_int64(chain_min), C.c_int64(amxpos), C.c_int64(bmxpos),
                             al.ctypes.data_as(C.c_void_p), C.byref(out))
    if nh < 0:
        raise MemoryError("oracle_chain_scan")
    rows = np.ctypeslib.as_array(out, shape=(max(nh, 1), 10))[:nh].copy() if nh else np.zeros((0, 10), np.int64)
    L.oracle_chain_free(out)
    return rows


def pack_keys(f, wa, wb, wd, wt):
    """sorted record fields -> the 128-bit keys of fga_seed_sort (lo64, hi64), LSB first: lcp(6) drem(6) anti(wt)
    bucket(wd) bctg(wb) actg(wa) strand(1)"""
    n = len(f["anti"])
    lo = np.zeros(n, dtype=np.uint64)
    hi = np.zeros(n, dtype=np.uint64)
    sh = 0
    for name, w in (("lcp", 6), ("drem", 6), ("anti", wt), ("bucket", wd), ("bctg", wb), ("actg", wa), ("strand", 1)):
        v = f[name].astype(np.uint64)
        if sh < 64:
            lo |= (v << np.uint64(sh)) if sh else v
            if sh + w > 64 and sh > 0:
                hi |= v >> np.uint64(64 - sh)
        else:
            hi |= v << np.uint64(sh - 64)
        sh += w
    out = np.empty(n, dtype=np.dtype([("lo", "<u8"), ("hi", "<u8")]))
    out["lo"], out["hi"] = lo, hi
    return out


def unpack_keys(k, wa, wb, wd, wt):
    """inverse of pack_keys (vectorised; Keys.fields of fastga_amd.device does the same with Python integers)"""
    lo, hi = k["lo"].astype(np.uint64), k["hi"].astype(np.uint64)
    out, sh = {}, 0
    for name, w in (("lcp", 6), ("drem", 6), ("anti", wt), ("bucket", wd), ("bctg", wb), ("actg", wa), ("strand", 1)):
        if sh >= 64:
            v = hi >> np.uint64(sh - 64)
        else:
            v = lo >> np.uint64(sh)
            if sh + w > 64 and sh > 0:
                v = v | (hi << np.uint64(64 - sh))
        out[name] = (v & np.uint64((1 << w) - 1)).astype(np.int64)
        sh += w
    return out
