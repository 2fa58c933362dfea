#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_ac
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
for fm in 2 5 9; do
STAN_LAB_PLACEMENT_FORCE_MISSES=$fm python3 - > $OUT/force_$fm.txt 2>&1 <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch, time
from stan_amd import hip, problem
job = problem.cube_job(148)
free0 = torch.cuda.mem_get_info()[0]
ctx = hip.Context(0); ctx.set_option(hip.OPT_CG_MERIT_STOP, 0); ctx.set_option(hip.OPT_PLACEMENT_TRIES, 32); ctx.set_profiling(True)
t0 = time.time()
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
t1 = time.time()
p = ctx.profile()
free1 = torch.cuda.mem_get_info()[0]
U, rep = K.cg_solve(job.F, 1e-8)
p2 = ctx.profile()
print("forced misses $fm: candidates %d, assemble wall %.3f s, probe kept %.4f slowest %.4f, device memory in use after the search %.2f GB, its %d, in-CG SpMV %.4f ms" %
      (p["placement_candidates"], t1 - t0, p["placement_ms_best"], p["placement_ms_worst"], (free0 - free1) / 1e9, rep["iterations"], p2["spmv_ms_total"] / p2["spmv_launches"]))
K.free(); ctx.close()
print("after close: %.2f GB in use" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e9))
PY
cat $OUT/force_$fm.txt | tail -2
done
