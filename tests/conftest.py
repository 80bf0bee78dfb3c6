import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so) -- the checker, never the thing under test."""
    from oracle import oracle as O
    return O


@pytest.fixture(scope="session")
def lib():
    """The product: the HIP library behind include/glfer_hip.h, built in-tree."""
    import glfer_amd
    glfer_amd.api.lib()
    return glfer_amd
