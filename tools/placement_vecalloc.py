"""Lab: vectors from different allocators against values from hipMalloc (plain allocation, no search).
Needs STAN_HIP_LIB=.../build_lab/libstan_hip_lab.so.  usage: placement_vecalloc.py [n=148]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_PLACEMENT_TRIES, 1)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
out = np.zeros(5)
ctx._chk(ctx.lib.stan_hip_lab_placement_vecalloc(ctx.h, K.k, out.ctypes.data_as(C.POINTER(C.c_double))))
print("SpMV ms, values in K's hipMalloc block; vectors: inside that block (same-group reference) %.4f | fresh hipMalloc %.4f | "
      "hipMallocAsync %.4f | virtual-memory API %.4f | the context's vectors %.4f" % tuple(out))
o8 = np.zeros(8)
ctx._chk(ctx.lib.stan_hip_lab_placement_vecshape(ctx.h, K.k, o8.ctypes.data_as(C.POINTER(C.c_double))))
print("vectors in: two fresh vector-sized blocks %.4f | one block of two vectors %.4f | one 1 GiB block %.4f | blocks 3 and 5 of eight vector-sized ones %.4f || "
      "allocated in the opposite order: %.4f | %.4f | %.4f | %.4f" % tuple(o8))
K.free(); ctx.close()
