"""Lab: is a slow value-stream block slow for EVERY access order, or only for the lockstep
front-to-back walk of the default kernel?  (needs the lab build:
STAN_HIP_LIB=stan_amd/csrc/build_lab/libstan_hip_lab.so)
usage: placement_variants.py [n=148] [ntries=8]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
ntries = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = [9, 0, 13, 14, 15, 16, 12]
job = problem.cube_job(n)
ctx = hip.Context(0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
info = K.info()
bytes_alg = info["n_blocks"] * 76 + info["n_block_rows"] * 52
ms = np.zeros(ntries * len(variants))
va = np.array(variants, dtype=np.int32)
ctx._chk(ctx.lib.stan_hip_lab_placement_variants(ctx.h, K.k, C.c_int32(ntries), C.c_int32(len(variants)),
                                                 va.ctypes.data_as(C.POINTER(C.c_int32)), C.c_int32(10),
                                                 ms.ctypes.data_as(C.POINTER(C.c_double))))
ms = ms.reshape(ntries, len(variants))
np.set_printoptions(linewidth=200, precision=4, suppress=True)
print("whole-SpMV ms, rows = candidate blocks, columns = variants", variants)
print("(9 default: nt + XCD-chunked; 0 plain; 13 = 9 without nt; 14 odd slices backwards; 15 rotated start; 16 hashed start; 12 = 9 unroll 4)")
print(ms)
print("per variant: min %s" % ms.min(axis=0), "\n             max %s" % ms.max(axis=0))
print("spread (max/min - 1) per variant in %%: %s" % (100 * (ms.max(axis=0) / ms.min(axis=0) - 1)))
print("fraction of 8 TB/s at the per-variant median: %s" % (bytes_alg / np.median(ms, axis=0) / 1e6 / 8000))
K.free(); ctx.close()
