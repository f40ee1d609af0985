import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's many small CPU ops (vmap(grad), eigh of a few hundred rows) are far slower on a pool as wide as the host's
    # 256 hardware threads than on a handful of cores
    import torch
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    try:                       # numpy's OpenBLAS pool is separate (the oracle's host eigh)
        from threadpoolctl import threadpool_limits
        config._blas_limit = threadpool_limits(limits=min(8, os.cpu_count() or 1))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def lib_built():
    """Build libvmcpde.so once per session (hipcc cross-compiles for gfx950 without a GPU)."""
    from vmc_pde_amd import build
    return build.build()
