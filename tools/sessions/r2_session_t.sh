#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_t
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all_$i.txt 2>&1
  grep -E "passed|failed|Error" $OUT/pytest_all_$i.txt | tail -3
done
# the fold under load: 148^3 solves alternating fold on/off, 6 times, bits must agree
python3 - > $OUT/fold_stress.txt 2>&1 <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, torch
from stan_amd import hip, problem
job = problem.cube_job(148)
ctx = hip.Context(0); ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
ref = None
for i in range(6):
    ctx.set_option(hip.OPT_CG_FOLD_REDUCE, i & 1)
    U, rep = K.cg_solve(job.F, 1e-8)
    if ref is None: ref = (U.copy(), rep)
    print(i, rep, "bit-identical:", bool(np.array_equal(U, ref[0]) and rep == ref[1]), flush=True)
PY
cat $OUT/fold_stress.txt
