"""Test helper: write the PRE-v1.3 layout of a genome index (distinct k-mers with counts in .ktab.N, positions in
.post.N, a stub with a second, 2^16-entry index of positions) from an index in today's layout.  Nothing in the tree can
produce such files any more (GIXmake writes today's layout only) but the reference's FastGA still reads them
(Open_Post_List / old_merge_thread, FastGA.c:206-570, 1027-1540), and so does fga_gix_open.  Format facts restated from
those readers: the stub is  kmer, nparts, minval, ibyte (int32) | int64 index[2^24] of table entries | postbytes,
contbytes, nfiles (int32) | maxp (int64) | freq, nctg (int32) | perm[nctg] (int32) | int64 index[2^16] of positions;
a .ktab part is  kmer (int32), n (int64), n x 9 bytes [7 suffix bytes, count, lcp];  a .post part is  postbytes,
contbytes (int32), n (int64), n x (postbytes+contbytes) bytes."""
import os
import shutil
import struct

import numpy as np

NPRE = 1 << 24


def write_legacy_index(src_root, dst_root, freq=255):
    """src_root / dst_root: paths without extension; the .gdb/.bps files are copied along"""
    sd, sr = os.path.split(src_root)
    dd, dr = os.path.split(dst_root)
    os.makedirs(dd, exist_ok=True)
    for ext in (".gdb", ".1gdb", ".bps"):
        for hidden in ("", "."):
            p = os.path.join(sd, hidden + sr + ext)
            if os.path.exists(p):
                shutil.copy(p, os.path.join(dd, hidden + dr + ext))
    raw = open(src_root + ".gix", "rb").read()
    kmer, nparts, minval, ibyte = struct.unpack_from("<4i", raw, 0)
    off = 16
    index = np.frombuffer(raw, dtype="<i8", count=NPRE, offset=off).copy()
    off += 8 * NPRE
    postb, contb, nfile = struct.unpack_from("<3i", raw, off); off += 12
    maxp, = struct.unpack_from("<q", raw, off); off += 8
    fq, nctg = struct.unpack_from("<2i", raw, off); off += 8
    perm = raw[off:off + 4 * nctg]; off += 4 * nctg
    sentinel, = struct.unpack_from("<q", raw, off)
    assert sentinel < 0, "the source must be in today's layout"
    eb = 9 + postb + contb
    parts, pbeg = [], [0]
    for p in range(1, nparts + 1):
        b = open(os.path.join(sd, f".{sr}.ktab.{p}"), "rb").read()
        k, n = struct.unpack_from("<iq", b, 0)
        assert k == kmer
        parts.append(np.frombuffer(b, dtype=np.uint8, count=n * eb, offset=12).reshape(n, eb))
        pbeg.append(pbeg[-1] + n)
    tab = np.concatenate(parts) if parts else np.zeros((0, eb), np.uint8)
    n = len(tab)
    assert n == index[-1]
    # prefix of every entry, then groups of equal (prefix, 7 suffix bytes)
    pre = np.searchsorted(index, np.arange(n), side="right")
    suf = np.zeros(n, np.uint64)
    for j in range(7):
        suf = (suf << np.uint64(8)) | tab[:, j].astype(np.uint64)
    first = np.ones(n, bool)
    if n > 1:
        first[1:] = (pre[1:] != pre[:-1]) | (suf[1:] != suf[:-1])
    starts = np.flatnonzero(first)
    counts = np.diff(np.append(starts, n))
    assert counts.max(initial=0) <= 255, "a k-mer with more than 255 positions does not fit the old count byte"
    old = np.zeros((len(starts), 9), np.uint8)
    old[:, :7] = tab[starts, :7]
    old[:, 7] = counts
    old[:, 8] = tab[starts, 8]
    posts = np.ascontiguousarray(tab[:, 9:])
    kindex = np.searchsorted(pre[starts], np.arange(NPRE), side="right").astype("<i8")       # inclusive cumulative
    pindex = index[(np.arange(1 << 16) << 8) | 0xff].astype("<i8")
    with open(dst_root + ".gix", "wb") as f:
        f.write(struct.pack("<4i", kmer, nparts, minval, ibyte))
        f.write(kindex.tobytes())
        f.write(struct.pack("<3i", postb, contb, nparts))
        f.write(struct.pack("<q", maxp))
        f.write(struct.pack("<2i", freq, nctg))
        f.write(perm)
        f.write(pindex.tobytes())
    gstart = np.searchsorted(starts, np.array(pbeg))          # group index at every part boundary
    for p in range(nparts):
        g0, g1 = int(gstart[p]), int(gstart[p + 1])
        with open(os.path.join(dd, f".{dr}.ktab.{p + 1}"), "wb") as f:
            f.write(struct.pack("<iq", kmer, g1 - g0))
            f.write(old[g0:g1].tobytes())
        with open(os.path.join(dd, f".{dr}.post.{p + 1}"), "wb") as f:
            f.write(struct.pack("<2iq", postb, contb, pbeg[p + 1] - pbeg[p]))
            f.write(posts[pbeg[p]:pbeg[p + 1]].tobytes())
    return dst_root
