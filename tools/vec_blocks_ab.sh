#!/bin/bash
# A/B of the grid of the CG's vector kernels inside the CG (round 4): STAN_VEC_BLOCKS = 1024 / 2048 (shipped) / 4096 / 16384.
# usage (GPU box, repo root): bash tools/vec_blocks_ab.sh <outdir>
OUT=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/$OUT
cd $R/stan_amd/csrc
for B in 1024 2048 4096 16384; do
  rm -rf build_lab
  make -s -j8 lab CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast -DSTAN_VEC_BLOCKS=$B" > /dev/null 2>&1 || exit 1
  cp build_lab/libstan_hip_lab.so /tmp/libstan_vb_$B.so
done
rm -rf build_lab
cd $R
for round in 1 2 3; do
  for B in 1024 2048 4096 16384; do
    STAN_HIP_LIB=/tmp/libstan_vb_$B.so timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > $OUT/bench_${B}_$round.json 2> $OUT/bench_${B}_$round.err
    python3 - <<PY
import json
d = json.load(open("$OUT/bench_${B}_$round.json")); r = d["roofline"]; c = d["config"]
its = c["cg_iterations"]
spmv = r["avg_launch_ms"] * r["launches"] / d["steps"] + r.get("two_product_avg_ms", 0) * r.get("two_product_launches", 0) / d["steps"]
print("blocks %5d round $round: %.4f M DOF/s, %.1f ms/step, SpMV %.4f ms, non-product time per iteration %.4f ms" %
      ($B, d["value"] / 1e6, d["ms_per_step"], r["avg_launch_ms"], (c["cg_ms"] - spmv) / its))
PY
  done
done
