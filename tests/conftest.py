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
    """Bring the in-tree shared libraries and the console driver up to date (make is a no-op
    when they are; hipcc cross-compiles without a GPU) -- a stale binary against a changed
    header is worse than a missing one."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def gpu_ctx(built_libs):
    import torch  # noqa: F401  (first, so one HIP runtime is shared)
    from stan_amd import hip
    ctx = hip.Context(0)
    yield ctx
    ctx.close()
