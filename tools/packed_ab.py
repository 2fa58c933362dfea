"""Packed column stream on/off in ONE process per size and value stream (in-CG SpMV, whole solve).
usage: packed_ab.py [sizes=148,200] [streams=fp64,fixed48,mixed] [rounds=2]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "148,200").split(",")]
streams = (sys.argv[2] if len(sys.argv) > 2 else "fp64,fixed48,mixed").split(",")
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
PM = {"fp64": hip.PREC_FP64, "fixed48": hip.PREC_FIXED48, "mixed": hip.PREC_MIXED}
for n in sizes:
    job = problem.cube_job(n)
    ctx = hip.Context(0); ctx.set_option(hip.OPT_CG_MERIT_STOP, 0); ctx.set_option(hip.OPT_PLACEMENT_TRIES, 24); ctx.set_profiling(True)
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    for tag in streams:
        for rnd in range(rounds):
            for packed in (0, 1):
                ctx.set_option(hip.OPT_PACKED_COLUMNS, packed)
                U, rep = K.cg_solve(job.F, 1e-8, precision_mode=PM[tag])
                p = ctx.profile()
                ms = p["spmv_ms_total"] / max(p["spmv_launches"], 1)
                print("n %d %-8s round %d packed %d: cg %.1f ms  in-CG SpMV %.4f ms  bytes %.4e -> %.0f GB/s  its %d  slots packed %d of %d" %
                      (n, tag, rnd, packed, p["cg_ms"], ms, p["spmv_bytes"], p["spmv_bytes"] / ms / 1e6, rep["iterations"],
                       p["col_slots_packed"], K.info()["n_slots"]), flush=True)
    K.free(); ctx.close()
