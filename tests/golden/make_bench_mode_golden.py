"""Generates tests/golden/bench_mode_*.npz: the ORACLE's answer on the jobs BASELINE.json's configs name.

    python tests/golden/make_bench_mode_golden.py [--out DIR] [--gpu] job [job ...]
        job = 100 | 148 | 200 ...   the n^3 HEX8_G2 cube of bench.py (clamp x = 0, PointLoad (0,0,50) on x = n)
              p120:0.4              the 120^3 box with 40 % of its elements knocked out (cube.perforated_mesh)

Bench mode = what bench.py runs: merit-function stop off, eps 1e-8.  The oracle
(oracle/stan_oracle.c: ParallelAssembly_K + LinearSolver_CG restated, SolverFunctions.cs:117-180, 270-330) is run
HERE, once, by this script -- 100 s at 100^3, 8 min and ~60 GB at 148^3, 25 min and ~150 GB at 200^3 -- and the
fixture keeps what a test needs to hold the GPU path against it without repeating that run:
    iterations, terminationtype, rel_residual          the oracle's report
    idx [4096] int64, U [4096] float64                 the oracle's displacements at a fixed, seeded sample of
                                                       reduced DOF indices (the entry of largest |U| included)
    u_max, u_sum, u_l2                                 max |U|, sum U, ||U||_2 over ALL reduced DOFs
    n_dof, n_red, n_elem                               size check of the regenerated job
    oracle_s [2]                                       assembly / CG seconds of the oracle on this host (serial CG,
                                                       K_e on `cores` threads: the reference's parallelism)
With --gpu the same job is also solved through the C-ABI on cuda:0 and the comparison is printed (a check that
the fixture and the library agree on the day it is made; the tests repeat it on the driver's box).

The jobs are regenerated from seeds (stan_amd.problem); nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

SAMPLE = 4096
SEED = 20261002
EPS = 1e-8


def make_job(name):
    from stan_amd import problem
    if name.startswith("p"):
        n, frac = name[1:].split(":")
        return problem.perforated_job(int(n), float(frac)), "bench_mode_p%s_k%s" % (n, frac)
    return problem.cube_job(int(name)), "bench_mode_%s" % name


def sample_indices(n_red, U=None):
    """The fixed sample: SAMPLE distinct reduced-DOF indices from a seeded generator, ascending; when U is given
    (generation) the index of its largest entry replaces the last one so that max |U| itself is held."""
    rng = np.random.default_rng(SEED)
    idx = np.sort(rng.choice(n_red, size=min(SAMPLE, n_red), replace=False)).astype(np.int64)
    if U is not None:
        top = int(np.abs(U).argmax())
        if top not in idx:
            idx[-1] = top
            idx = np.sort(idx)
    return idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("jobs", nargs="+")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--mv-threads", type=int, default=1,
                    help="threads of the oracle's matrix-vector product (1 = serial, what alglib does)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import bench_legs as bench
    from oracle import pyoracle as O
    cores = min(8, bench.effective_cores())
    for name in args.jobs:
        job, stem = make_job(name)
        t0 = time.perf_counter()
        rc, A = O.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red,
                           n_threads=cores)
        assert rc == 0
        t1 = time.perf_counter()
        O.set_mv_threads(args.mv_threads)
        U, rep = O.cg(A, job.F, EPS, merit_stop=False)
        O.set_mv_threads(1)
        t2 = time.perf_counter()
        del A
        idx = sample_indices(job.n_red, U)
        path = os.path.join(args.out, stem + ".npz")
        np.savez_compressed(path, idx=idx, U=U[idx], iterations=np.int64(rep["iterations"]),
                            terminationtype=np.int64(rep["terminationtype"]), rel_residual=np.float64(rep["rel_residual"]),
                            u_max=np.float64(np.abs(U).max()), u_sum=np.float64(U.sum()),
                            u_l2=np.float64(np.sqrt(np.dot(U, U))), n_dof=np.int64(job.n_dof),
                            n_red=np.int64(job.n_red), n_elem=np.int64(job.conn.shape[0]),
                            oracle_s=np.array([t1 - t0, t2 - t1]), cores=np.int64(cores),
                            mv_threads=np.int64(args.mv_threads), eps=np.float64(EPS))
        line = {"job": name, "fixture": os.path.basename(path), "bytes": os.path.getsize(path), "n_dof": job.n_dof,
                "oracle_iterations": rep["iterations"], "oracle_termination_type": rep["terminationtype"],
                "oracle_rel_residual": rep["rel_residual"],
                "cpu_port": {"value": job.n_dof / (t2 - t0), "unit": "DOF/s", "cores": cores, "kind": "port",
                             "seconds": t2 - t0, "assembly_s": t1 - t0, "cg_s": t2 - t1,
                             "mv_threads": args.mv_threads}}
        if args.gpu:
            import torch  # noqa: F401
            from stan_amd import hip
            ctx = hip.Context(0)
            ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
            t3 = time.perf_counter()
            K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
            Ug, repg = K.cg_solve(job.F, EPS)
            t4 = time.perf_counter()
            K.free()
            ctx.close()
            line.update({"gpu_iterations": repg["iterations"], "gpu_termination_type": repg["terminationtype"],
                         "gpu_s_host_pointers_cold": t4 - t3,
                         "max_abs_dU_over_max_abs_U": float(np.abs(Ug - U).max() / np.abs(U).max()),
                         "sample_max_abs_dU_over_max_abs_U": float(np.abs(Ug[idx] - U[idx]).max() / np.abs(U).max())})
        print(json.dumps(line), flush=True)
        del U


if __name__ == "__main__":
    main()
