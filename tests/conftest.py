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
    """The CPU restatement (test infrastructure)."""
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def ctx():
    """A libstereo_hip context on GPU 0.  Fails loudly if the library is missing."""
    # (either import order works: stereo_amd._lib binds to the HIP runtime torch ships, see
    # _bind_to_torchs_hip_runtime and tests/test_gpu_configs.py::test_library_before_torch)
    import stereo_amd
    c = stereo_amd.Context(0)
    yield c
    c.close()
