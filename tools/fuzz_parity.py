"""Fuzz sweep of the hot path against the oracle (tests/fuzz.py): python tools/fuzz_parity.py [first] [count] [fold] [kind] [mode]
fold = 1: the products of every job read the FOLDED streams (STAN_OPT_ROW_FOLDING forced on, the small-system kernel
off: the jobs are tiny), so the sweep exercises fold.hip's plans on a few hundred ragged meshes.
kind: box (default) | collapsed (box jobs with 5 % of their elements collapsed into wedges) | revolved (solids of revolution
with collapsed hexes on the axis, 3 ... 160 sectors: the high-valence slow paths).  mode: STAN_OPT_ASSEMBLY_MODE (0 / 1)."""
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
