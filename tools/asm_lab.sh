#!/bin/bash
# Builds k_numeric variants on the GPU box and times the assembly at the bench size.
cd $GRAFT_REPO_ROOT/stan_amd/csrc
for V in ${LAB_VARIANTS:-"2 2" "1 2" "1 3"}; do
  set -- $V
  rm -rf build_lab; mkdir -p build_lab
  for f in api assembly cg scan comm recovery; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast -DSTAN_GP_UNROLL=$1 -DSTAN_NUM_WAVES=$2 -c $f.hip -o build_lab/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libstan_lab_$1_$2.so build_lab/*.o -ldl
  echo "== GP_UNROLL=$1 NUM_WAVES=$2"
  STAN_HIP_LIB=/tmp/libstan_lab_$1_$2.so python3 $GRAFT_REPO_ROOT/tools/asm_time.py
done
