"""An irregular mesh at real size under the oracle: python tools/perforated_parity.py [n] [fraction]
The n^3 box with `fraction` of its elements removed (stan_amd.cube.perforated_mesh), bench mode (merit stop off,
eps 1e-8): the oracle's U (assembly on 8 threads, CG with the product on all cores) against the GPU's with the
padded streams and with folded rows (STAN_OPT_ROW_FOLDING 0 / default)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import bench_legs as bench
from stan_amd import hip, problem
from oracle import pyoracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
job = problem.perforated_job(n, frac)
args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
ctx.set_profiling(True)
gpu = {}
for fold in (0, -1):
    ctx.set_option(hip.OPT_ROW_FOLDING, fold)
    t0 = time.perf_counter()
    K = ctx.assemble_hex8(*args)
    U, rep = K.cg_solve(job.F, 1e-8)
    dt = time.perf_counter() - t0
    gpu[fold] = (U, rep, dt, ctx.profile()["repacked_streams"], K.info()["folded_slots_permille"])
    K.free()
t0 = time.perf_counter()
rc, A = O.assemble(*args, n_threads=min(8, bench.effective_cores()))
t1 = time.perf_counter()
O.set_mv_threads(bench.effective_cores())
Uo, repo = O.cg(A, job.F, 1e-8, merit_stop=False)
t2 = time.perf_counter()
um = float(np.abs(Uo).max())
print(json.dumps({"mesh": "%d^3 box, %.0f %% of the elements removed" % (n, 100 * frac), "n_dof": job.n_dof, "elements": int(job.conn.shape[0]),
                  "oracle_s": {"assembly": t1 - t0, "cg_all_cores": t2 - t1}, "oracle_iterations": repo["iterations"],
                  "oracle_termination_type": repo["terminationtype"],
                  "gpu_padded": {"iterations": gpu[0][1]["iterations"], "termination_type": gpu[0][1]["terminationtype"],
                                 "s_host_pointers": gpu[0][2], "max_abs_dU_over_max_abs_U": float(np.abs(gpu[0][0] - Uo).max() / um)},
                  "gpu_folded": {"iterations": gpu[-1][1]["iterations"], "termination_type": gpu[-1][1]["terminationtype"],
                                 "s_host_pointers": gpu[-1][2], "streams": gpu[-1][3], "folded_slots_permille": gpu[-1][4],
                                 "max_abs_dU_over_max_abs_U": float(np.abs(gpu[-1][0] - Uo).max() / um)},
                  "folded_vs_padded_max_abs_dU_over_max_abs_U": float(np.abs(gpu[-1][0] - gpu[0][0]).max() / um)}), flush=True)
