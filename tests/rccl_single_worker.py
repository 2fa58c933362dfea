"""One process, one GPU, a REAL 1-rank RCCL communicator (tests/test_gpu_parity.py runs this as a
child: a process that has held an RCCL communicator slows every later multi-process GPU test of the
same pytest session to a crawl on a shared device, so the pytest process itself never creates one).
Prints 'same_bits <0|1> its <a> <b>' and 'rccl <file> reused <0|1> mappings <n>': which librccl the
library resolved, whether it was the one the host (torch) had already mapped, and how many distinct
librccl files /proc/self/maps shows afterwards (never two RCCL builds in one process)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from stan_amd import hip, problem  # noqa: E402

job = problem.cube_job(8, jitter=0.1)
args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
res = []
for use_comm in (False, True):
    ctx = hip.Context(0)
    if use_comm:
        import torch.distributed  # noqa: F401  (libtorch_hip.so brings torch's bundled librccl.so into the process)
        ctx.comm_init(0, 1, ctx.unique_id())
        info = ctx.comm_info()
        mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
        print("rccl %s reused %d mappings %d version %d" % (info["library"], int(info["library_reused"]), len(mapped),
                                                             info["rccl_version"]))
        print("mapped " + " ".join(mapped))
    K = ctx.assemble_hex8(*args)
    res.append(K.cg_solve(job.F, 1e-10))
    K.free()
    ctx.close()
same = res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])
print("same_bits %d its %d %d" % (int(same), res[0][1]["iterations"], res[1][1]["iterations"]))
