"""The CPU port of the reference algorithm (oracle) at the sizes SURVEY.md section 8d lists, on this
host's cores, with the GPU path on the same jobs: python tools/cpu_sizes.py [n ...]  (default 24 100)
n = 24 is the 43 650-DOF class of the reference's screenshot run (9.03 s on the author's PC)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import bench_legs as bench
from stan_amd import hip, problem
sizes = [int(a) for a in sys.argv[1:]] or [24, 100]
ctx = hip.Context(0)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
for n in sizes:
    base, base_all, Uo, repo = bench.cpu_baseline(n, 1e-8, return_u=True)
    job = problem.cube_job(n)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        U, rep = K.cg_solve(job.F, 1e-8)
        dt = time.perf_counter() - t0
        K.free()
        best = dt if best is None else min(best, dt)
    print(json.dumps({"n": n, "n_dof": job.n_dof, "cpu_port": base, "cpu_port_all_cores": base_all,
                      "gpu_s_host_pointers": best, "gpu_DOF_per_s": job.n_dof / best,
                      "gpu_iterations": rep["iterations"], "oracle_iterations": repo["iterations"],
                      "gpu_termination_type": rep["terminationtype"], "oracle_termination_type": repo["terminationtype"],
                      # parity at this size: same seeded job, same mode (merit stop off, eps 1e-8)
                      "max_abs_U_diff_over_max_abs_U": float(np.abs(U - Uo).max() / np.abs(Uo).max()),
                      "kappa_eps_bound": 12.7 * n * n * 1e-8,
                      "speedup_vs_reference_like_port": job.n_dof / best / base["value"]}), flush=True)
