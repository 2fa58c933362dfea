"""Counter run for the placement question (lab library): SpMV launches cross-paired / self-paired / cross-paired.
Run directly under rocprofv3 (tools/placement_pmc.sh):  rocprofv3 --pmc <counter> ... -- python3 tools/placement_pmc.py [n] [reps]
Needs STAN_HIP_LIB=stan_amd/csrc/build_lab/libstan_hip_lab.so."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_PLACEMENT_TRIES, 1)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
out = np.zeros(3)
ctx._chk(ctx.lib.stan_hip_lab_pairing_pmc(ctx.h, K.k, C.c_int32(reps), out.ctypes.data_as(C.POINTER(C.c_double))))
print("PAIRING_MS cross %.4f self %.4f cross %.4f  (reps %d, n %d)" % (out[0], out[1], out[2], reps, n), flush=True)
K.free(); ctx.close()
