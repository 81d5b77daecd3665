import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def libfdg():
    """The HIP extension must exist; build it (cross-compile) when it does not."""
    from feynmandiagram_jl_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return capi.lib()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the product has no CPU fallback)")
    return torch.device("cuda:0")


@pytest.fixture
def no_shipped_cache(libfdg, monkeypatch):
    """Tests that look at what a specialisation writes into their own cache directory: without the read-only lookup of the
    code objects shipped in feynmandiagram.jl_amd/kernel_cache (a hit there writes nothing)."""
    monkeypatch.setenv("FDG_CACHE_RO_DIR", "")
