#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_n
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -5
python3 - > $OUT/packed_ab.txt 2>&1 <<PY
import sys, os
sys.path.insert(0, "$R")
import numpy as np, torch
from stan_amd import hip, problem
for n in (148, 200):
    job = problem.cube_job(n)
    ctx = hip.Context(0); ctx.set_option(hip.OPT_CG_MERIT_STOP, 0); ctx.set_option(hip.OPT_PLACEMENT_TRIES, 24); ctx.set_profiling(True)
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    for prec, tag in ((hip.PREC_FP64, "fp64"), (hip.PREC_FIXED48, "fixed48"), (hip.PREC_MIXED, "mixed")):
        for rnd in range(2):
            for packed in (0, 1):
                ctx.set_option(hip.OPT_PACKED_COLUMNS, packed)
                U, rep = K.cg_solve(job.F, 1e-8, precision_mode=prec)
                p = ctx.profile()
                ms = p["spmv_ms_total"] / max(p["spmv_launches"], 1)
                print("n %d %-8s round %d packed %d: cg %.1f ms  in-CG SpMV %.4f ms  bytes %.4e -> %.0f GB/s  its %d  slots packed %d of %d" %
                      (n, tag, rnd, packed, p["cg_ms"], ms, p["spmv_bytes"], p["spmv_bytes"] / ms / 1e6, rep["iterations"], p["col_slots_packed"], K.info()["n_slots"]), flush=True)
    K.free(); ctx.close()
PY
cat $OUT/packed_ab.txt
