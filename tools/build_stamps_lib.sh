#!/bin/bash
# Diagnostic build of the library with the pair chase's per-phase stamps (-DPAIR_STAMPS) next to the product library:
#   bash tools/build_stamps_lib.sh   ->  springcraft_amd/libspringcraft_hip_stamps.so   (git-ignored; select it with
#   SPRINGCRAFT_HIP_LIB=...; tools/pair_stamps.py and tools/r04_final.sh b use it)
set -eu
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
python springcraft_amd/csrc/build.py > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -c springcraft_amd/csrc/twostage.hip -o /tmp/twostage_stamps.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall \
  -Wno-unused-function -I include -mllvm -pragma-unroll-threshold=1000000 -DPAIR_STAMPS
$HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_stamps.so /tmp/twostage_stamps.o \
  $(ls springcraft_amd/csrc/obj/*.o | grep -v twostage.o)
echo springcraft_amd/libspringcraft_hip_stamps.so
