"""One PROCESS, several ranks on GPU 0 (stan_hip_init_multi), the CG's exchanges once over the RCCL
stand-in (tests/fake_rccl) and once PEER TO PEER (STAN_OPT_COMM_P2P): tests/test_gpu_transports.py compares
the bits.  Needs GPU_MAX_HW_QUEUES >= 2 * nranks + 2 in the environment (ranks share the device).
usage: p2p_worker.py <n | perf:n:frac> <nranks> <out.npz>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from stan_amd import hip, problem  # noqa: E402

spec, nranks, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if spec.startswith("perf:"):
    from stan_amd.problem import perforated_job
    _, n, frac = spec.split(":")
    job = perforated_job(int(n), float(frac))
else:
    job = problem.cube_job(int(spec), jitter=0.05)
ctx = hip.Context(devices=[0] * nranks)
ctx.set_profiling(True)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
res = {}
for tag, p2p in (("rccl", 0), ("p2p", 1), ("rccl2", 0), ("p2p2", 1)):
    ctx.set_option(hip.OPT_COMM_P2P, p2p)
    info = ctx.comm_info()
    assert info["p2p"] == bool(p2p), info
    for loop, sr in (("classic", 0), ("sr", 1)):
        ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, sr)
        for pname, prec, eps, maxit in (("f64", hip.PREC_FP64, 1e-9, 0), ("fx48", hip.PREC_FIXED48, 1e-9, 0),
                                        ("cap", hip.PREC_FP64, 1e-30, 37)):   # MaxIts inside a refresh cycle: type 5
            U, rep = K.cg_solve(job.F, eps, maxit, precision_mode=prec)
            prof = ctx.profile()
            key = "%s_%s_%s" % (tag, loop, pname)
            res["U_" + key] = U
            res["rep_" + key] = np.array([rep["terminationtype"], rep["iterations"]])
            res["res_" + key] = rep["rel_residual"]
            res["coll_" + key] = np.array([prof["loop_collectives"], prof["loop_stream_waits"],
                                           prof["loop_kernel_launches"], prof["loop_iterations_enqueued"],
                                           prof["comm_reduce_calls"], prof["comm_halo_calls"]])
            res["ms_" + key] = np.array([prof["comm_reduce_ms_total"], prof["comm_halo_ms_total"], prof["cg_ms"]])
ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 0)
# merit stop off + fold off: the unfolded reduction launches publish peer to peer too
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
for tag, p2p in (("rccl", 0), ("p2p", 1)):
    ctx.set_option(hip.OPT_COMM_P2P, p2p)
    for fold in (1, 0):
        ctx.set_option(hip.OPT_CG_FOLD_REDUCE, fold)
        U, rep = K.cg_solve(job.F, 1e-9)
        res["U_%s_nomerit_fold%d" % (tag, fold)] = U
        res["rep_%s_nomerit_fold%d" % (tag, fold)] = np.array([rep["terminationtype"], rep["iterations"]])
res["n_halo"] = K.info()["n_halo"]
np.savez(out, **res)
K.free()
ctx.close()
print("P2P_WORKER_OK")
