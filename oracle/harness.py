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
