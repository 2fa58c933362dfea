"""SpMV time against the offset of the value stream inside its allocation (STAN_LAB_VALS_OFFSET)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0)
offs = list(range(0, 4096 + 1, 256)) + [8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304, 0, 512]
for off in offs:
    os.environ["STAN_LAB_VALS_OFFSET"] = str(off)
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    t = [K.spmv_bench(20) for _ in range(3)]
    print("offset %8d: spmv_bench %.4f %.4f %.4f ms  plan %s" % (off, *t, hex(K.info()["n_slots"])), flush=True)
    K.free()
