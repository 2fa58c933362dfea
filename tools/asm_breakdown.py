"""Assembly wall/event times over repeated calls at size n (allocation effects vs kernel time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
job = problem.cube_job(n)
ctx = hip.Context(0); ctx.set_profiling(True)
for i in range(5):
    t0 = time.perf_counter()
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    t1 = time.perf_counter()
    p = ctx.profile()
    K.free()
    t2 = time.perf_counter()
    print("call %d: wall %.1f ms (host-pointer entry incl. uploads), events: symbolic %.1f + numeric %.1f ms; free %.1f ms"
          % (i, (t1 - t0) * 1e3, p["symbolic_ms"], p["numeric_ms"], (t2 - t1) * 1e3), flush=True)
