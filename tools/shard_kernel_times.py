"""What one rank of an N-rank solve costs, as far as ONE GPU can measure it:  python tools/shard_kernel_times.py [n] [its]

  (a) the shard's own product: a DETACHED middle rank of N = 2 / 4 / 8 (no communicator: partition, assembly and the local
      product work) assembles its rows of the n^3 cube and times the CG's SpMV kernel on them (stan_hip_spmv_bench);
  (b) what a communicator adds to an iteration when nothing has to travel: the same capped solve on one rank without a
      communicator and with a REAL 1-rank RCCL communicator (two all-reduce launches per classic iteration, one per
      single-reduction iteration; no halo), per iteration.

Neither is a multi-GPU measurement: (a) is the compute share of a rank, (b) the launch floor of the collectives.  The
link latency and the wait for the slowest rank are not in it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from stan_amd import hip, problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
its = int(sys.argv[2]) if len(sys.argv) > 2 else 400
job = problem.cube_job(n)
args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
print("cube %d^3: %d DOF" % (n, job.F.shape[0]), flush=True)

print("(a) SpMV of one rank's rows (fp64, packed columns as the solve selects them)")
for nranks in (1, 2, 4, 8):
    rank = nranks // 2
    ctx = hip.Context(0)
    if nranks > 1:
        ctx.comm_init(rank, nranks, None)
    K = ctx.assemble_hex8(*args)
    info = K.info()
    K.spmv_bench(50)
    ms = min(K.spmv_bench(200) for _ in range(3))
    print("  N = %d rank %d: %9d owned block rows, %7d halo, %d slots  ->  %.4f ms  (x N = %.3f)"
          % (nranks, rank, info["row_end"] - info["row_begin"], info["n_halo"], info["n_slots"], ms, ms * nranks), flush=True)
    K.free()
    ctx.close()

print("(b) capped solve of %d iterations, one rank, merit stop off" % its)
for single in (0, 1):
    row = []
    for use_comm in (False, True):
        ctx = hip.Context(0)
        ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
        ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, single)
        if use_comm:
            ctx.comm_init(0, 1, ctx.unique_id())
        K = ctx.assemble_hex8(*args)
        F = torch.from_numpy(job.F).cuda()
        U = torch.zeros_like(F)
        K.cg_solve_dev(F.data_ptr(), U.data_ptr(), 1e-30, 50)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r = K.cg_solve_dev(F.data_ptr(), U.data_ptr(), 1e-30, its)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3 / max(r["iterations"], 1))
        row.append(best)
        K.free()
        ctx.close()
    print("  %s loop: %.4f ms / iteration alone, %.4f with a 1-rank RCCL communicator  (+%.1f us)"
          % ("single-reduction" if single else "classic", row[0], row[1], (row[1] - row[0]) * 1e3), flush=True)
