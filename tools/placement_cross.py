"""Lab: does the placement of the gather vector / product matter as well as that of the values?
Needs STAN_HIP_LIB=.../build_lab/libstan_hip_lab.so.   usage: placement_cross.py [n=148] [ntries=16]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
ntries = int(sys.argv[2]) if len(sys.argv) > 2 else 16
job = problem.cube_job(n)
ctx = hip.Context(0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
out = np.zeros(ntries); cf = np.zeros(ntries); cs = np.zeros(ntries)
bf, bs = C.c_int32(0), C.c_int32(0)
P = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
ctx._chk(ctx.lib.stan_hip_lab_placement_cross(ctx.h, K.k, C.c_int32(ntries), P(out), P(cf), P(cs), C.byref(bf), C.byref(bs)))
np.set_printoptions(linewidth=220, precision=4, suppress=True)
print("class probe (values in block t, vectors from the pool), ms:\n", out)
print("values in the FASTEST block (%d), x and y inside block t (t = %d: inside the value block itself), ms:\n" % (bf.value, bf.value), cf)
print("values in the SLOWEST block (%d), x and y inside block t, ms:\n" % bs.value, cs)
ok = (cf > 0) & (cs > 0)
if ok.sum() > 2:
    print("correlation of block t's own class probe with the SpMV time when only the VECTORS live in it: "
          "values fast %.2f, values slow %.2f" % (np.corrcoef(out[ok], cf[ok])[0, 1], np.corrcoef(out[ok], cs[ok])[0, 1]))
K.free(); ctx.close()
