#!/bin/bash
# Build the library with extra compiler flags into /tmp and run a script against it (STAN_HIP_LIB).
# usage: bash tools/lib_lab.sh "<script and args>" "-DFLAG=1" "-DFLAG=0" ...
cd $GRAFT_REPO_ROOT/stan_amd/csrc
SCRIPT=$1; shift
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  rm -rf build_lab; mkdir -p build_lab
  for f in api assembly assembly_scatter placement cg scan comm recovery csr_lab; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast $FLAGS -c $f.hip -o build_lab/$f.o &
  done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libstan_lab_$i.so build_lab/*.o -ldl
  echo "== $FLAGS"
  STAN_HIP_LIB=/tmp/libstan_lab_$i.so python3 $GRAFT_REPO_ROOT/$SCRIPT
done
rm -rf build_lab
