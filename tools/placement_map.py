"""Lab: map of where a slow value-stream block loses its time.
Needs the lab build: STAN_HIP_LIB=stan_amd/csrc/build_lab/libstan_hip_lab.so.
usage: placement_map.py [n=148] [ntries=8] [nseg=16] [cg=1]
Prints per candidate block: address, whole-SpMV ms, per-segment SpMV ms, per-segment plain-read
GB/s; then (cg=1) the in-CG SpMV time with K living in the fastest and in the slowest candidate."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
ntries = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nseg = int(sys.argv[3]) if len(sys.argv) > 3 else 16
do_cg = int(sys.argv[4]) if len(sys.argv) > 4 else 1
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
ctx.set_profiling(True)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
info = K.info()
bytes_alg = info["n_blocks"] * 76 + info["n_block_rows"] * 52
seg_bytes = info["n_slots"] * 64 * 72 / nseg
per = 1 + 2 * nseg


def scan(keep):
    ms = np.zeros(ntries * per)
    addr = np.zeros(ntries, dtype=np.uint64)
    ctx._chk(ctx.lib.stan_hip_lab_placement_map(ctx.h, K.k, C.c_int32(ntries), C.c_int32(nseg), C.c_int32(keep),
                                                ms.ctypes.data_as(C.POINTER(C.c_double)),
                                                addr.ctypes.data_as(C.POINTER(C.c_uint64))))
    return ms.reshape(ntries, per), addr


ms, addr = scan(0)
print("candidate  address            whole-SpMV ms (GB/s)   sum of segments ms")
for t in range(ntries):
    if ms[t, 0] < 0:
        continue
    print("%2d  0x%012x  %.4f (%5.0f)  %.4f" % (t, addr[t], ms[t, 0], bytes_alg / ms[t, 0] / 1e6, ms[t, 1:1 + nseg].sum()))
np.set_printoptions(linewidth=250, precision=1, suppress=True)
print("per-segment SpMV time in us (rows = candidates, %d segments front to back):" % nseg)
print(ms[:, 1:1 + nseg] * 1e3)
print("per-segment plain front-to-back read in GB/s:")
print(seg_bytes / (ms[:, 1 + nseg:] * 1e-3) / 1e9)
valid = ms[:, 0] > 0
fast, slow = ms[valid, 0].min(), ms[valid, 0].max()
print("spread: fastest %.4f ms, slowest %.4f ms (%.1f %%)" % (fast, slow, 100 * (slow / fast - 1)))
if do_cg:
    for keep, tag in ((1, "fastest"), (2, "slowest")):
        ms2, addr2 = scan(keep)
        U, rep = K.cg_solve(job.F, 1e-8)
        pr = ctx.profile()
        print("CG with K in the %s candidate of a fresh scan (whole-SpMV %.4f ms back to back): %d its, cg %.1f ms, in-CG SpMV %.4f ms" %
              (tag, ms2[valid, 0].min() if keep == 1 else ms2[valid, 0].max(), rep["iterations"], pr["cg_ms"],
               pr["spmv_ms_total"] / max(pr["spmv_launches"], 1)))
K.free()
ctx.close()
