"""Erratum helper: bench JSONs written before the launch-accounting fix averaged the SpMV time over
ALL enqueued launches, including the few dozen early-exit launches queued behind a converged solve.
Recompute avg_launch_ms / frac over the launches that did work (its - its//10 single-product
launches per solve with the fused refresh, its + its//10 with the literal one).
usage: python tools/recount_launches.py profiles/r01/bench_*.json"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); r = d["roofline"]; c = d["config"]
    its, steps, L = c["cg_iterations"], d["steps"], r["launches"]
    if r.get("two_product_launches") is not None:
        work = (its - its // 10) * steps          # fused refresh: single-product launches only
    elif L / steps < its + its // 10 - 60:
        work = its * steps                         # transitional files: both launch kinds in one average
    else:
        work = (its + its // 10) * steps           # literal refresh: one extra product every 10 iterations
    if L <= work:
        print("%-46s already counts working launches only (%d)" % (f.split("/")[-1], L)); continue
    fac = L / work
    print("%-46s launches %5d of which working %5d: avg %.4f -> %.4f ms, frac %.3f -> %.3f" %
          (f.split("/")[-1], L, work, r["avg_launch_ms"], r["avg_launch_ms"] * fac, r["frac"], r["frac"] / fac))
