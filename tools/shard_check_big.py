"""Detached-rank shard check at bench size: every rank's shard of the 148^3 cube (default 8 ranks) is
assembled on this GPU; device plan == host plan, shard x [owned | halo] == rows of the unsharded product."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
from tests import fuzz
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
job = problem.cube_job(n)
t0 = time.time()
fuzz.check_shards(lambda: hip.Context(0), job, nranks)
print("%d^3 on %d detached ranks: plans equal, shard products bit-equal to the unsharded one (%.1f s)" % (n, nranks, time.time() - t0))
