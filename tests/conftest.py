import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build products once per session: hipcc cross-compiles without a GPU
    import __graft_entry__ as ge
    ge.build_lib()
    ge.build_oracle()


@pytest.fixture(scope="session")
def gx():
    import gnxraytracer_amd as gx
    gx.lib()
    return gx


@pytest.fixture(scope="session")
def gpu(gx):
    """Initialises device 0; the GPU tests fail loudly (not skip) when the HIP path is unavailable."""
    gx.init(0)
    return gx


def golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name))
