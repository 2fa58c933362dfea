"""Fuzz sweep of the hot path against the oracle (tests/fuzz.py): python tools/fuzz_parity.py [first] [count] [fold] [kind] [mode]
fold = 1: the products of every job read the FOLDED streams (STAN_OPT_ROW_FOLDING forced on, the small-system kernel
off: the jobs are tiny), so the sweep exercises fold.hip's plans on a few hundred ragged meshes.
kind: box (default) | collapsed (box jobs with 5 % of their elements collapsed into wedges) | revolved (solids of revolution
with collapsed hexes on the axis, 3 ... 160 sectors: the high-valence slow paths) | round5 (box jobs through the paths round 5
added, the large-system kernel forced: the first product scaling the matrix against the scaling pass, bit for bit; the fp32
copy refined against the fp64 answer, its reported residual against an independent one).  mode: STAN_OPT_ASSEMBLY_MODE (0 / 1)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from stan_amd import hip
from oracle import pyoracle as oracle
from tests import fuzz
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = hip.Context(0)
fold = len(sys.argv) > 3 and int(sys.argv[3]) != 0
if fold:
    ctx.set_option(hip.OPT_SPMV_SMALL, 0)
    ctx.set_option(hip.OPT_ROW_FOLDING, 1)
    ctx.set_profiling(True)
kind = sys.argv[4] if len(sys.argv) > 4 else "box"
ctx.set_option(hip.OPT_ASSEMBLY_MODE, int(sys.argv[5]) if len(sys.argv) > 5 else 0)
if kind == "round5":
    ctx.set_option(hip.OPT_SPMV_SMALL, 0)
    ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    ctx.set_profiling(True)
    n_ok = n_skip = n_refined = n_singular = 0
    worst_du = worst_res_gap = 0.0
    for seed in range(first, first + count):
        job = fuzz.random_job(seed)
        if job is None or job.has_g1 or job.n_red == 0 or np.linalg.norm(job.F) == 0:
            n_skip += 1
            continue
        args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        res = {}
        for lazy in (0, 1):
            ctx.set_option(hip.OPT_CG_LAZY_SCALING, lazy)
            K = ctx.assemble_hex8(*args)
            U, rep = K.cg_solve(job.F, 1e-9, 20000)
            res[lazy] = (U, rep, K)
        assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0]), (seed, res[0][1], res[1][1])
        K = res[1][2]
        res[0][2].free()
        if res[1][1]["terminationtype"] == 1:       # (a floating sub-structure: K singular, nothing to compare)
            Um, repm = K.cg_solve(job.F, 1e-9, 60000, precision_mode=hip.PREC_MIXED)
            pr = ctx.profile()
            indep = K.scaled_residual(job.F, Um)
            gap = abs(indep - repm["rel_residual"]) / max(repm["rel_residual"], 1e-300)
            if repm["rel_residual"] > 1e-12:
                worst_res_gap = max(worst_res_gap, gap)
                assert gap <= 0.2, (seed, repm, indep)
            if repm["terminationtype"] == 1:
                assert repm["rel_residual"] <= 1e-9, (seed, repm)
                du = float(np.abs(Um - res[1][0]).max() / np.abs(res[1][0]).max())
                if du > 1e-4:
                    # a mechanism the SPCs leave free (K singular, the load compatible): both answers solve the system and
                    # differ by a null vector -- the restarted passes take another path through it than the one fp64 run
                    null = float(np.linalg.norm(K.spmv(Um - res[1][0])) / np.linalg.norm(job.F))
                    assert null <= 1e-7, (seed, du, null, repm)
                    n_singular += 1
                else:
                    worst_du = max(worst_du, du)
                n_refined += pr["refine_passes"] > 1
        K.free()
        n_ok += 1
        if seed % 20 == 0:
            print("seed %d: its %d, mixed %s" % (seed, res[1][1]["iterations"], repm if res[1][1]["terminationtype"] == 1 else "-"), flush=True)
    print("round5 sweep: %d jobs (lazy scaling == scaling pass bit for bit), %d skipped; fp32 copy: %d needed refinement passes, "
          "worst |U - U64| / |U64| %.2e (%d singular systems: the two answers differ by a null vector, checked), worst reported-vs-independent "
          "residual gap %.1e" % (n_ok, n_skip, n_refined, worst_du, n_singular, worst_res_gap))
    sys.exit(0)
folded_jobs = 0
ok = skipped = 0
worst = {"k_err": 0.0, "u_err": 0.0, "res": 0.0, "res48": 0.0}
for seed in range(first, first + count):
    job = (fuzz.random_revolved_job(seed) if kind == "revolved" else
           fuzz.random_job(seed, collapse=0.05 if kind == "collapsed" else 0.0))
    if job is None:
        skipped += 1
        continue
    try:
        out = fuzz.check_job(ctx, oracle, job)
    except AssertionError as e:
        print("seed %d FAILED: %s" % (seed, e))
        raise
    ok += 1
    if fold and ctx.profile()["repacked_streams"]:
        folded_jobs += 1
    worst["k_err"] = max(worst["k_err"], out.get("k_err", 0.0))
    worst["u_err"] = max(worst["u_err"], out.get("u_err", 0.0))
    worst["res"] = max(worst["res"], out.get("res", (0.0, 0.0))[0])
    worst["res48"] = max(worst["res48"], out.get("res48", 0.0))
    if seed % 20 == 0:
        print("seed %d: %s" % (seed, out), flush=True)
print("fuzz: %d jobs checked, %d disconnected meshes skipped; worst %s" % (ok, skipped, worst))
if fold:
    print("folded streams read in the last solve of %d jobs" % folded_jobs)
