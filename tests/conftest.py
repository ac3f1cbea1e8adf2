import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build the library once where a compiler is available, so
    # that the CPU suite (C ABI exports, ONNX import, mirror headers) does not depend on a previous `__graft_entry__.build()`
    lib = os.path.join(ROOT, "hyperpose_amd", "libhp_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def hp():
    """libhp_hip.so initialised on device 0; GPU tests FAIL (not skip) when it cannot be loaded."""
    from hyperpose_amd import _lib
    _lib.init(0)
    return _lib
