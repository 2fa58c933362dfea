"""One PROCESS driving several ranks through stan_hip_init_multi (tests/test_gpu_multi.py): the
ranks are worker threads of the library; on the one-GPU test box they all drive GPU 0 and RCCL
is replaced by tests/fake_rccl (STAN_RCCL_LIB), because real RCCL refuses two ranks on one device.
usage: multi_worker.py <n> <nranks> <out.npz>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from stan_amd import hip, problem  # noqa: E402

n, nranks, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
job = problem.cube_job(n, jitter=0.05)
ctx = hip.Context(devices=[0] * nranks)
ctx.set_profiling(True)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
info = K.info()
U, rep = K.cg_solve(job.F, 1e-6)
prof = ctx.profile()
Ux, repx = K.cg_solve(job.F, 1e-6, precision_mode=hip.PREC_FIXED48)
ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 1)
Us, reps = K.cg_solve(job.F, 1e-6)
profs = ctx.profile()
ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 0)
ctx.set_option(hip.OPT_PACKED_COLUMNS, 0)      # int32 column stream: the same bits on every shard
Up, repp = K.cg_solve(job.F, 1e-6)
ctx.set_option(hip.OPT_PACKED_COLUMNS, 1)
assert repp == rep and np.array_equal(Up, U), "packed column stream changed the sharded solve"
# per-element stress recovery split over the devices
disp = np.zeros(job.n_dof); disp[job.red != -1] = U
strain, stress = ctx.recover_hex8(job.xyz, disp[job.node_dof], job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
# ... and the same with the results kept on the devices, mapped across the chunk boundaries
res = ctx.recover_hex8_keep(job.xyz, disp[job.node_dof], job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
ne = job.conn.shape[0]
keep_equal = True
for a, b in ((0, ne), (3, ne - 2), (ne // nranks - 1, ne // nranks + 2), (ne - 1, ne)):
    e2, s2 = res.map(a, b)
    keep_equal = keep_equal and np.array_equal(e2, strain[a:b]) and np.array_equal(s2, stress[a:b])
res.free()
unsupported = 0
try:
    K.to_csr()
except hip.StanHipError as e:
    unsupported = e.code
np.savez(out, U=U, Ux=Ux, Us=Us, its=rep["iterations"], term=rep["terminationtype"], its_s=reps["iterations"],
         term_s=reps["terminationtype"], its_x=repx["iterations"], strain=strain, stress=stress,
         n_blocks=info["n_blocks"], n_halo=info["n_halo"], n_elem_dev=info["n_elements_on_device"], unsupported=unsupported, keep_equal=keep_equal,
         coll_per_it=prof["loop_collectives"] / max(prof["loop_iterations_enqueued"], 1),
         coll_per_it_s=profs["loop_collectives"] / max(profs["loop_iterations_enqueued"], 1))
K.free()
ctx.close()
