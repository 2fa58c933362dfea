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
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def built_libs():
    """Make sure the in-tree shared libraries exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    from stan_amd import hip, host
    if not (os.path.exists(hip.LIB_PATH) and os.path.exists(host.LIB_PATH)):
        g.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built_libs):
    import torch  # noqa: F401  (first, so one HIP runtime is shared)
    from stan_amd import hip
    ctx = hip.Context(0)
    yield ctx
    ctx.close()
