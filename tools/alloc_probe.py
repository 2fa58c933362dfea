"""Is the big hipMalloc inside the assembly slow once torch's CUDA context exists?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
job = problem.cube_job(n)
ctx = hip.Context(0); ctx.set_profiling(True)
def run(tag):
    for i in range(2):
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        p = ctx.profile(); K.free()
        print("%s call %d: events symbolic %.1f + numeric %.1f ms" % (tag, i, p["symbolic_ms"], p["numeric_ms"]), flush=True)
run("before torch.cuda init")
t = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
run("after a 1-element cuda tensor")
big = torch.empty(1 << 28, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()   # 2 GiB
run("with a 2 GiB torch tensor alive")
del big; torch.cuda.empty_cache()
run("after empty_cache")
print(torch.cuda.memory_summary(abbreviated=True)[:600])
