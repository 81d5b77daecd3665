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


class _FdgOptions:
    """The library's process-default options for the duration of one test (fdg_set_default_option): what handles created from now on start
    with.  Replaces monkeypatch.setenv("FDG_...") -- the library reads the environment once per process.  A handle
    that already exists is changed with ``handle.set_option``."""

    def __init__(self):
        self._saved = {}

    def set(self, name, value):
        from feynmandiagram_jl_amd import capi
        if name not in self._saved:
            self._saved[name] = capi.get_default_option(name)
        capi.set_default_option(name, value)

    def unset(self, name):
        self.set(name, None)

    def restore(self):
        from feynmandiagram_jl_amd import capi
        for name, v in self._saved.items():
            capi.set_default_option(name, v)
        self._saved.clear()


@pytest.fixture
def fdgopt(libfdg):
    o = _FdgOptions()
    yield o
    o.restore()


@pytest.fixture
def no_shipped_cache(libfdg, fdgopt):
    """Tests that look at what a specialisation writes into their own cache directory: without the read-only lookup of the
    code objects shipped in feynmandiagram.jl_amd/kernel_cache (a hit there writes nothing)."""
    fdgopt.set("FDG_CACHE_RO_DIR", "")
