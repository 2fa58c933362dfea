import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as O
from stan_amd import problem
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup cpu.max", e)
j = problem.cube_job(40)
rc, A = O.assemble(j.xyz, j.node_dof, j.conn, j.elem_mat, j.elem_type, j.mat_E_nu, j.red, n_threads=8)
for t in (1, 2, 4, 8, 16, 32):
    O.set_mv_threads(t)
    t0 = time.time(); U, r = O.cg(A, j.F, 1e-8, merit_stop=False); print(t, "threads: CG %.2f s" % (time.time() - t0), r["iterations"])
