#!/usr/bin/env python3
"""k_recover (stress / strain recovery, Element.cs:211-267) timed at a cube size, inputs resident in HBM
(stan_hip_recover_hex8_dev): HIP events over `reps` launches, algorithmic bytes per element = 768 B written
(strain + stress, 2 x 48 doubles) + 224 B read (8 node indices + its share of coordinates and displacements:
SURVEY.md section 8f).  Run under rocprofv3 --kernel-trace / --pmc for the kernel's own time and HBM traffic.
usage: recover_time.py [n=148] [reps=10]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from stan_amd import hip, problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
job = problem.cube_job(n)
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
rng = np.random.default_rng(7)
disp = rng.standard_normal((job.xyz.shape[0], 3)) * 1e-3
d_xyz = torch.from_numpy(job.xyz).to(dev)
d_disp = torch.from_numpy(disp).to(dev)
d_conn = torch.from_numpy(job.conn).to(dev)
d_mat = torch.from_numpy(job.elem_mat).to(dev)
d_typ = torch.from_numpy(job.elem_type).to(dev)
ne = job.conn.shape[0]
d_e = torch.empty(ne * 48, dtype=torch.float64, device=dev)
d_s = torch.empty(ne * 48, dtype=torch.float64, device=dev)
E = np.ascontiguousarray(job.mat_E_nu, dtype=np.float64).reshape(-1, 2)


def run():
    ctx._chk(ctx.lib.stan_hip_recover_hex8_dev(
        ctx.h, C.c_int64(job.xyz.shape[0]), hip._dev(d_xyz.data_ptr(), C.c_double), hip._dev(d_disp.data_ptr(), C.c_double),
        C.c_int64(ne), hip._dev(d_conn.data_ptr(), C.c_int32), hip._dev(d_mat.data_ptr(), C.c_int32),
        hip._dev(d_typ.data_ptr(), C.c_uint8), C.c_int32(E.shape[0]), hip._ptr(E, C.c_double),
        hip._dev(d_e.data_ptr(), C.c_double), hip._dev(d_s.data_ptr(), C.c_double)))


run()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(reps):
    run()          # (the call synchronises its stream: wall time per call = kernel + ~20 us)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
alg = ne * (768 + 224)
# yardstick (round 6): a kernel that ONLY writes the two result arrays (torch's fill, HIP events): three quarters of k_recover's
# bytes are write-once stores, and a write stream does not reach the read rate of this memory system
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
d_e.zero_(); d_s.zero_(); torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    d_e.zero_(); d_s.zero_()
e1.record(); torch.cuda.synchronize()
fill_ms = e0.elapsed_time(e1) / reps
print(json.dumps({"kernel": "k_recover<false>", "n": n, "elements": ne, "reps": reps, "ms_per_call_wall": ms,
                  "algorithmic_bytes": alg, "GBs_wall": alg / (ms * 1e-3) / 1e9, "frac_of_8TBs_wall": alg / (ms * 1e-3) / 8e12,
                  "write_only_fill_of_the_results_ms": fill_ms, "write_only_fill_GBs": ne * 768 / (fill_ms * 1e-3) / 1e9,
                  "wall_over_write_only_fill": ms / fill_ms}))
ctx.close()
