import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """libgndt.so, built in-tree if absent (hipcc cross-compiles without a GPU)."""
    import grid_ndt_amd as g
    g.build_native()
    from grid_ndt_amd import _lib
    return _lib.lib()
