"""Lab: what distinguishes a slow value-stream block -- translation cost (page-touch probes) and
allocation strategy.  Needs STAN_HIP_LIB=stan_amd/csrc/build_lab/libstan_hip_lab.so.
usage: placement_alloc.py [n=148]"""
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
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
strategy = np.array([4, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 0, 0, 1, 2], dtype=np.int32)
names = {0: "hipMalloc(bytes)", 1: "hipMalloc(pow2)", 2: "hipMalloc(GiB-rounded)", 3: "hipExtMalloc(uncached)", 4: "K's own (pool)"}
out = np.zeros(len(strategy) * 5)
addr = np.zeros(len(strategy), dtype=np.uint64)
ctx._chk(ctx.lib.stan_hip_lab_placement_alloc(ctx.h, K.k, C.c_int32(len(strategy)), strategy.ctypes.data_as(C.POINTER(C.c_int32)),
                                              out.ctypes.data_as(C.POINTER(C.c_double)), addr.ctypes.data_as(C.POINTER(C.c_uint64))))
out = out.reshape(-1, 5)
print("%-24s %-16s %9s %10s %10s %10s %9s" % ("strategy", "address", "SpMV ms", "touch4K ms", "touch64K ms", "touch2M ms", "alloc ms"))
for i, s in enumerate(strategy):
    print("%-24s 0x%012x %9.4f %10.4f %10.4f %10.5f %9.1f" % (names[int(s)], addr[i], *out[i]))
ok = out[:, 0] > 0
c = np.corrcoef(out[ok, 0], out[ok, 1])[0, 1], np.corrcoef(out[ok, 0], out[ok, 2])[0, 1], np.corrcoef(out[ok, 0], out[ok, 3])[0, 1]
print("correlation of the SpMV time with the touch times (4K, 64K, 2M): %.2f %.2f %.2f" % c)
K.free(); ctx.close()
