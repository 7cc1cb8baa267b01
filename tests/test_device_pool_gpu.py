"""GPU: the device-memory pool behind every allocation of the library (fga_device.hip): a release leaves a free piece, the
next request of that size or less is cut from it, neighbours merge, unused regions go back on fga_dev_trim."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

MiB = 1 << 20


def _malloc(L, dev, n):
    p = C.c_void_p()
    assert L.fga_dev_malloc(dev, n, C.byref(p)) == 0
    return p.value


def test_pieces_are_split_merged_and_reused():
    from fastga_amd.lib import load_library
    L = load_library()
    dev = C.c_void_p()
    assert L.fga_dev_open(0, C.byref(dev)) == 0
    L.fga_dev_trim(dev)
    base_avail = L.fga_dev_available(dev)
    r = _malloc(L, dev, 256 * MiB)                       # one region
    L.fga_dev_free(dev, C.c_void_p(r))
    assert abs(L.fga_dev_available(dev) - base_avail) <= 64 * MiB        # held by the pool, counted as available
    a = _malloc(L, dev, 64 * MiB)                        # cut from its front
    b = _malloc(L, dev, 64 * MiB)
    c = _malloc(L, dev, 100 * MiB)
    assert a == r and b == r + 64 * MiB and c == r + 128 * MiB
    L.fga_dev_free(dev, C.c_void_p(b))
    d = _malloc(L, dev, 30 * MiB)                        # the smallest free piece that fits: the 28 MiB tail does not, b's hole does
    assert d == b
    e = _malloc(L, dev, 20 * MiB)                        # tail of the region: 256 - 228 = 28 MiB
    assert e == r + 228 * MiB
    for p in (a, c, d, e):
        L.fga_dev_free(dev, C.c_void_p(p))
    f = _malloc(L, dev, 256 * MiB)                       # everything merged again
    assert f == r
    small = _malloc(L, dev, 4096)                        # below a MiB: not a piece of the pool
    assert not (r <= small < r + 256 * MiB)
    L.fga_dev_free(dev, C.c_void_p(small))
    L.fga_dev_free(dev, C.c_void_p(f))
    # data written through one piece is what a copy back reads (the piece is real memory at that address)
    import numpy as np
    g = _malloc(L, dev, 8 * MiB)
    src = np.arange(2 * MiB, dtype=np.uint32)
    out = np.zeros_like(src)
    assert L.fga_dev_upload(dev, C.c_void_p(g), src.ctypes.data_as(C.c_void_p), src.nbytes) == 0
    assert L.fga_dev_download(dev, out.ctypes.data_as(C.c_void_p), C.c_void_p(g), out.nbytes) == 0
    assert np.array_equal(src, out)
    L.fga_dev_free(dev, C.c_void_p(g))
    L.fga_dev_trim(dev)                                  # the region goes back to the driver
    h = _malloc(L, dev, 512 * MiB)
    L.fga_dev_free(dev, C.c_void_p(h))
    L.fga_dev_close(dev)


def test_a_session_s_second_run_allocates_nothing_new(toy_pair, tmp_path):
    """the buffers of a run are pieces of regions the pool keeps: the device footprint after a second, identical run is
    the footprint after the first"""
    from fastga_amd import device as D
    d, ra, rb = toy_pair
    ses = D.Session(ra, rb)
    ses.run(out_path=str(tmp_path / "a.1aln"))
    L = ses.L
    dev = L.fga_session_device(ses.h)
    p1 = L.fga_dev_peak_bytes(dev)
    a1 = L.fga_dev_available(dev)
    ses.run(out_path=str(tmp_path / "b.1aln"))
    assert abs(L.fga_dev_peak_bytes(dev) - p1) <= 64 * MiB and abs(L.fga_dev_available(dev) - a1) <= 64 * MiB
    assert open(tmp_path / "a.1aln", "rb").read() == open(tmp_path / "b.1aln", "rb").read()
    ses.close()
