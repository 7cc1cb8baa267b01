"""Python view of the device-side C-ABI (include/fastga_amd.h): Device, DeviceGix, seed merge."""
import ctypes as C

import numpy as np

from .lib import load_library, check, MergeParams, FgaError, STAGE_MERGE, STAGE_MERGE_PARTITION  # noqa: F401

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
