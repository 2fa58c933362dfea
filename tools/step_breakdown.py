"""assemble + CG + free, repeated: per-call wall and event times (what bench.py's step() does)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = len(sys.argv) > 2 and sys.argv[2] == "dev"
job = problem.cube_job(n)
ctx = hip.Context(0); ctx.set_profiling(True); ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
if dev:
    d = {k: torch.from_numpy(getattr(job, k)).cuda() for k in ("xyz", "node_dof", "conn", "elem_mat", "elem_type", "red", "F")}
    dU = torch.zeros(job.n_red, dtype=torch.float64, device="cuda")
for i in range(4):
    t0 = time.perf_counter()
    if dev:
        K = ctx.assemble_hex8_dev(job.xyz.shape[0], d["xyz"].data_ptr(), d["node_dof"].data_ptr(), job.conn.shape[0],
                                  d["conn"].data_ptr(), d["elem_mat"].data_ptr(), d["elem_type"].data_ptr(),
                                  job.mat_E_nu, job.n_dof, d["red"].data_ptr())
    else:
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    t1 = time.perf_counter()
    pa = ctx.profile()
    if dev:
        rep = K.cg_solve_dev(d["F"].data_ptr(), dU.data_ptr(), 1e-8, 400)
    else:
        U, rep = K.cg_solve(job.F, 1e-8, 400)
    t2 = time.perf_counter()
    pc = ctx.profile()
    K.free()
    t3 = time.perf_counter()
    print("call %d: assemble wall %.1f ms (events %.1f + %.1f); cg(400 its) wall %.1f ms (events %.1f); free %.1f ms"
          % (i, (t1 - t0) * 1e3, pa["symbolic_ms"], pa["numeric_ms"], (t2 - t1) * 1e3, pc["cg_ms"], (t3 - t2) * 1e3), flush=True)
