#!/bin/bash
# A/B of the load policy of the vector kernels inside the CG (round 4): non-temporal (shipped until now) against plain loads
# for the read-only operands (v in k_step; r, x in k_update) and for the operands rewritten in place (r in k_step, p in k_update).
# usage (GPU box, repo root): bash tools/vec_load_ab.sh <outdir> [bench args]
OUT=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd $R/stan_amd/csrc
i=0
for FLAGS in "" "-DSTAN_VLD_RO_NT=0" "-DSTAN_VLD_RO_NT=0 -DSTAN_VLD_RMW_NT=0" "-DSTAN_VLD_RMW_NT=0"; do
  i=$((i+1))
  rm -rf build_lab
  make -s -j8 lab CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast $FLAGS" > /dev/null 2>&1 || exit 1
  cp build_lab/libstan_hip_lab.so /tmp/libstan_vld_$i.so
done
rm -rf build_lab
cd $R
for round in 1 2; do
  i=0
  for FLAGS in "nt_all" "ro_plain" "all_plain" "rmw_plain"; do
    i=$((i+1))
    STAN_HIP_LIB=/tmp/libstan_vld_$i.so timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/bench_${FLAGS}_$round.json 2> $OUT/bench_${FLAGS}_$round.err
    python3 - <<PY
import json
d = json.load(open("$OUT/bench_${FLAGS}_$round.json")); r = d["roofline"]; c = d["config"]
its = c["cg_iterations"]
spmv = r["avg_launch_ms"] * r["launches"] / d["steps"] + r.get("two_product_avg_ms", 0) * r.get("two_product_launches", 0) / d["steps"]
print("%-10s round $round: %.4f M DOF/s, %.1f ms/step, SpMV %.4f ms, spmv2 %.4f ms, non-product time per iteration %.4f ms, probe kept %.4f" %
      ("$FLAGS", d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], r.get("two_product_avg_ms", 0), (c["cg_ms"] - spmv) / its, c["placement_search"]["probe_ms_kept"]))
PY
  done
done
