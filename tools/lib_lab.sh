#!/bin/bash
# Build the LAB flavour of the library (-DSTAN_LAB: all A/B kernel variants + lab/csr_lab.hip) with
# extra compiler flags into /tmp and run a script against it (STAN_HIP_LIB).  The product library
# (stan_amd/lib/libstan_hip.so) carries none of this.
# usage: bash tools/lib_lab.sh "<script and args>" "-DFLAG=1" "-DFLAG=0" ...   ("" = no extra flag)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}/stan_amd/csrc
SCRIPT=$1; shift
[ $# -eq 0 ] && set -- ""
i=0
for FLAGS in "$@"; do
  i=$((i+1))
  rm -rf build_lab
  make -s -j8 lab CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -ffp-contract=fast $FLAGS" || exit 1
  cp build_lab/libstan_hip_lab.so /tmp/libstan_lab_$i.so
  echo "== $FLAGS"
  STAN_HIP_LIB=/tmp/libstan_lab_$i.so python3 ${GRAFT_REPO_ROOT:-../..}/$SCRIPT
done
rm -rf build_lab
