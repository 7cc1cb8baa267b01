"""numpy views of an opened GIX / GDB (host side), used by tests, bench and tools."""
import ctypes as C

import numpy as np

from .lib import load_library, check

NPREFIX = 1 << 24


class Gdb:
    def __init__(self, path):
        self.L = load_library()
        self.h = C.c_void_p()
        check(self.L.fga_gdb_open(path.encode(), C.byref(self.h)), f"open GDB {path}")
        self.path = path
        self.ncontig = self.L.fga_gdb_ncontig(self.h)
        self.seqtot = self.L.fga_gdb_seqtot(self.h)
        self.maxctg = self.L.fga_gdb_maxctg(self.h)
        self.clen = np.array([self.L.fga_gdb_contig_len(self.h, c) for c in range(self.ncontig)], dtype=np.int64)

    def contig(self, c):
        """numeric contig (0..3) with the sentinel 4 either side stripped."""
        n = int(self.clen[c])
        buf = np.empty(n + 2, dtype=np.uint8)
        self.L.fga_gdb_get_contig(self.h, c, buf.ctypes.data_as(C.c_void_p))
        return buf[1:n + 1]

    def close(self):
        if self.h:
            self.L.fga_gdb_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Gix:
    def __init__(self, path, handle=None):
        """open <root>.gix (+ .ktab parts), or wrap a handle produced by fga_dgix_build (path is then only a label;
        index/table are present only if the build kept a host copy)"""
        self.L = load_library()
        self.h = C.c_void_p()
        if handle is not None:
            self.h = handle
        else:
            check(self.L.fga_gix_open(path.encode(), C.byref(self.h)), f"open GIX {path}")
        L, h = self.L, self.h
        self.path = path
        self.nents = L.fga_gix_nents(h)
        self.ebytes = L.fga_gix_ebytes(h)
        self.postbytes = L.fga_gix_postbytes(h)
        self.contbytes = L.fga_gix_contbytes(h)
        self.pbyte = self.postbytes + self.contbytes
        self.nctg = L.fga_gix_nctg(h)
        self.nparts = L.fga_gix_nparts(h)
        self.partbeg = np.array([L.fga_gix_part_begin(h, p) for p in range(self.nparts + 1)], dtype=np.int64)
        self.maxpre = L.fga_gix_maxpre(h)
        self.perm = np.ctypeslib.as_array(L.fga_gix_perm(h), shape=(self.nctg,)).copy()
        self.index = np.ctypeslib.as_array(L.fga_gix_index(h), shape=(NPREFIX,)) if L.fga_gix_index(h) else None
        self.table = (np.ctypeslib.as_array(L.fga_gix_table(h), shape=(self.nents * self.ebytes,))
                      if L.fga_gix_table(h) else None)

    def entries(self):
        return self.table.reshape(self.nents, self.ebytes)

    def close(self):
        if self.h:
            self.index = self.table = None
            self.L.fga_gix_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fasta_to_gdb(fasta, target, ncut=0):
    L = load_library()
    check(L.fga_fasta_to_gdb(fasta.encode(), target.encode(), ncut), f"FASTA->GDB {fasta}")


def build_gix(gdb, target, nthreads=8, use_mask=False):
    check(gdb.L.fga_gix_build_masked(gdb.h, target.encode(), nthreads, int(use_mask)), f"GIX build {target}")
