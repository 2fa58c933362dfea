"""Lab: is the 'placement lottery' a property of the block or of the moment?  Candidates are all
allocated first, then timed round-robin.  Needs STAN_HIP_LIB=.../build_lab/libstan_hip_lab.so.
usage: placement_rounds.py [n=148] [ntries=6] [nrounds=40] [reps=5] [pause_ms=0]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
a = [int(v) for v in sys.argv[1:]] + [None] * 5
n, ntries, nrounds, reps, pause = a[0] or 148, a[1] or 6, a[2] or 40, a[3] or 5, a[4] or 0
job = problem.cube_job(n)
ctx = hip.Context(0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
ms = np.zeros(nrounds * ntries); ts = np.zeros(nrounds * ntries)
ctx._chk(ctx.lib.stan_hip_lab_placement_rounds(ctx.h, K.k, C.c_int32(ntries), C.c_int32(nrounds), C.c_int32(reps), C.c_int32(pause),
                                               ms.ctypes.data_as(C.POINTER(C.c_double)), ts.ctypes.data_as(C.POINTER(C.c_double))))
ms = ms.reshape(nrounds, ntries); ts = ts.reshape(nrounds, ntries)
np.set_printoptions(linewidth=220, precision=3, suppress=True)
print("whole-SpMV ms; rows = rounds (time ->), columns = candidate blocks; last column = host seconds at the end of the round")
print(np.column_stack([ms, ts[:, -1]]))
print("per candidate over the rounds: mean %s\n                               std  %s" % (ms.mean(axis=0), ms.std(axis=0)))
print("per round over the candidates: std  %s" % ms.std(axis=1))
tot = ms.var()
between_cand = ms.mean(axis=0).var()
between_round = ms.mean(axis=1).var()
print("variance decomposition: total %.5f; between candidates %.5f (%.0f %%); between rounds %.5f (%.0f %%)" %
      (tot, between_cand, 100 * between_cand / tot, between_round, 100 * between_round / tot))
K.free(); ctx.close()
