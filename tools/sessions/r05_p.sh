#!/bin/bash
# Round 5, session p: config 2 (100^3) -- one wavefront per slice (default above 150 000 block rows) against one workgroup
# per slice (k_spmv_small) at that size, and sizes in between.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for n in 60 80 100; do
  for lim in -1 4000000; do
    timeout 600 python3 bench.py --size $n --steps 3 --warmup 1 --no-cpu --no-secondary --spmv-small-rows $lim > $O/bench_n${n}_small${lim}.json 2>> $O/err.txt
  done
done
python3 - <<'PY' > $O/summary.txt
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r05p/bench_n*.json"))):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(os.path.basename(f), "value %.3e ms/step %.1f spmv %.4f ms frac %.3f its %d" % (d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["config"]["cg_iterations"]))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
cat $O/summary.txt
