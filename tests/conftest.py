import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def co():
    """C++ CPU oracle (oracle/_build/liboracle.so)."""
    from oracle import coracle
    coracle.build()
    coracle.lib()
    return coracle


@pytest.fixture(scope="session")
def pr():
    from oracle import pyref
    return pyref


@pytest.fixture(scope="session")
def bzk():
    """libbzk context on cuda:0.  torch's default stream has the null handle, for which bzk_ctx_create makes its own
    non-blocking stream: tests that build inputs with torch kernels synchronise before handing them over.  GPU tests
    only - no fallback."""
    import torch
    from bazuka_amd import Bzk
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    torch.cuda.set_device(0)
    ctx = Bzk(0, torch.cuda.current_stream().cuda_stream)
    yield ctx
    ctx.close()
