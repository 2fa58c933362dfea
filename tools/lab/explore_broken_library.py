import sys, os, pathlib, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.chdir(sys.path[0])
from tests import test_gpu_sharded as T
from tests.conftest import fake_rccl_env
for n in (100, 148):
    for rep in range(2):
        for tag, env in (("sync", fake_rccl_env("sync")), ("syncdev", dict(fake_rccl_env("sync"), FAKE_RCCL_SYNC_DEVICE="1")), ("async", fake_rccl_env("async"))):
            tmp = pathlib.Path(tempfile.mkdtemp())
            T._broken_library_run(tmp, "%s_%d_%d" % (tag, n, rep), n, env)
