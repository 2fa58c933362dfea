#!/usr/bin/env python3
"""Round 6: the mid-size product -- what one of 8 ranks runs at 148^3 -- with one wavefront per slice
(the library's choice, variant 9) against two wavefronts per slice (k_spmv_pair: variant 20).
bench.py per size and variant in child processes; prints SpMV ms per launch, fraction of 8 TB/s, DOF/s, iterations.
usage: python tools/spmv_pair_sizes.py [sizes...]   (default 60 72 80 100)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(a) for a in sys.argv[1:]] or [60, 72, 80, 100]
print("%5s %8s %12s %8s %12s %6s %14s" % ("n", "variant", "spmv ms", "frac", "DOF/s", "its", "rel_residual"))
for n in sizes:
    base = None
    for v in (-1, 20):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--size", str(n), "--steps", "3", "--warmup", "1", "--no-cpu",
                              "--no-secondary", "--spmv-variant", str(v)], capture_output=True, text=True, cwd=ROOT, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("%5d %8d  failed: %s" % (n, v, out.stderr[-300:].replace("\n", " | ")))
            continue
        d = json.loads(line[-1])
        r, c = d["roofline"], d["config"]
        base = base or r["avg_launch_ms"]
        print("%5d %8d %12.4f %8.3f %12.4e %6d %14.3e   x%.3f" % (n, v, r["avg_launch_ms"], r["frac"], d["value"] or 0.0, c["cg_iterations"],
                                                                 c["rel_residual"], base / r["avg_launch_ms"]), flush=True)
