"""fastga_amd -- MI355X-native seed-and-extend hot path of FastGA (host side mirrors the reference's C seams).

The product is `libfastga_amd.so` (C-ABI in include/fastga_amd.h); this package is the thin ctypes binding used
by tests, bench.py and the tools.  There is no CPU fallback: device entry points fail loudly without a GPU.
"""
from .lib import load_library, FgaError  # noqa: F401
