#!/bin/bash
# Timing-only builds of k_numeric on the GPU box (ablations / tuning macros), 148^3 assembly.
# usage: bash tools/asm_lab.sh "-DSTAN_ABL=1" "-DSTAN_ABL=2" ...
cd $GRAFT_REPO_ROOT/stan_amd/csrc
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  rm -rf build_lab; mkdir -p build_lab
  for f in api assembly assembly_scatter placement cg scan comm recovery csr_lab; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast $FLAGS -c $f.hip -o build_lab/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libstan_lab_$i.so build_lab/*.o -ldl
  echo "== $FLAGS"
  STAN_HIP_LIB=/tmp/libstan_lab_$i.so python3 $GRAFT_REPO_ROOT/tools/asm_time.py
done
rm -rf build_lab
