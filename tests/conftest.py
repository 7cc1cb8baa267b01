import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """The in-tree libfastga_amd.so; built on demand (hipcc cross-compiles without a GPU)."""
    from fastga_amd.lib import lib_path, load_library
    if not os.path.exists(lib_path()):
        import __graft_entry__ as g
        g.build()
    return load_library()


@pytest.fixture(scope="session")
def toy_pair(tmp_path_factory, built_library):
    """~0.6 Mbp pair, 12 contigs, 3 % divergence with repeats and rearrangements, built by our own tools."""
    from fastga_amd import workload
    d = str(tmp_path_factory.mktemp("toy"))
    ra, rb = workload.build_pair(d, seed=11, ncontig=12, total=600_000, divergence=0.03,
                                 repeat_frac=0.05, inv_frac=0.05, swap_frac=0.05)
    return d, ra, rb
