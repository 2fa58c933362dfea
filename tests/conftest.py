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


def fake_rccl_env(mode, nranks_in_one_process=0):
    """Environment of a child that runs over tests/fake_rccl in `mode`:
      "sync"   every call drains its stream and moves the data on the host (the stand-in of rounds 1-5);
      "async"  FAKE_RCCL_ASYNC=1: send / recv / all-reduce are enqueued on the caller's stream like RCCL's.
    Ranks that share one PROCESS (stan_hip_init_multi) and one device need a hardware queue per stream in
    asynchronous mode, exactly as the product's peer-to-peer path does (a polling kernel blocks its queue)."""
    env = {"FAKE_RCCL_ASYNC": "1" if mode == "async" else "0"}
    if mode == "async" and nranks_in_one_process:
        env["GPU_MAX_HW_QUEUES"] = str(2 * nranks_in_one_process + 4)
    return env


@pytest.fixture(params=["sync", "async"])
def fake_mode(request):
    return request.param
