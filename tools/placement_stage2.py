"""The two stages of the placement search at one size, probe by probe (STAN_PLACEMENT_TRACE):  python tools/placement_stage2.py [n] [tries] [max_gb]
Stage 1 times candidate blocks for the value stream; stage 2 (round 4) moves only the vectors the products write, behind
spacer blocks (placement.hip).  Prints the search's trace, the SpMV time of the matrix as placed and the profile."""
import os, sys
os.environ.setdefault("STAN_PLACEMENT_TRACE", "1")   # "sweep": walk all spacers of stage 2, keep nothing
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 32
max_gb = float(sys.argv[3]) if len(sys.argv) > 3 else 0
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_PLACEMENT_TRIES, tries)
if max_gb > 0:
    ctx.set_option(hip.OPT_PLACEMENT_MAX_BYTES, int(max_gb * 1e9))
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
K.spmv_bench(50)                                   # (the first launches after the host's work run ~2 % slow)
ms = min(K.spmv_bench(20) for _ in range(3))
ctx.set_option(hip.OPT_PACKED_COLUMNS, 0)
ms_i32 = min(K.spmv_bench(20) for _ in range(3))
ctx.set_option(hip.OPT_PACKED_COLUMNS, 1)
p = ctx.profile()
print("n = %d: SpMV as placed %.4f ms (with the int32 columns the probes read: %.4f); search: %d candidates, kept %.4f ms, slowest %.4f ms, moved = %d" %
      (n, ms, ms_i32, p["placement_candidates"], p["placement_ms_best"], p["placement_ms_worst"], p["placement_moved_vectors"]))
