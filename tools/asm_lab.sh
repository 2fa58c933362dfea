#!/bin/bash
# Timing-only builds of k_numeric on the GPU box (ablations / tuning macros), 148^3 assembly.
# usage: bash tools/asm_lab.sh "-DSTAN_ABL=1" "-DSTAN_ABL=2" ...
#   STAN_ABL: 1 no block arithmetic (phase B), 2 no ordered accumulation (phase D), 3 no write-out, 4 no slot search,
#             5 no Jacobians (phase A), 6 no incidence chunks at all (set-up, chains, write-out only)
cd $GRAFT_REPO_ROOT/stan_amd/csrc
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  rm -rf build_abl; mkdir -p build_abl
  for f in api assembly assembly_scatter placement cg fold scan comm p2p recovery multi; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-function -ffp-contract=fast $FLAGS -c $f.hip -o build_abl/$f.o 2>/dev/null &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libstan_abl_$i.so build_abl/*.o -ldl
  echo "== $FLAGS"
  STAN_HIP_LIB=/tmp/libstan_abl_$i.so python3 $GRAFT_REPO_ROOT/tools/asm_time.py
done
rm -rf build_abl
