import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hp():
    """libhp_hip.so initialised on device 0; GPU tests FAIL (not skip) when it cannot be loaded."""
    from hyperpose_amd import _lib
    _lib.init(0)
    return _lib
