#!/bin/bash
# Diagnostic build of the library with k_sytrd_resident's per-segment stamps (-DRES_STAMPS) next to the product library:
#   bash tools/build_res_stamps_lib.sh  ->  springcraft_amd/libspringcraft_hip_res_stamps.so  (git-ignored; select it with
#   SPRINGCRAFT_HIP_LIB=...; tools/resident_check.py --stamps uses it)
set -eu
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
python springcraft_amd/csrc/build.py > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -c springcraft_amd/csrc/tridiag.hip -o /tmp/tridiag_stamps.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall \
  -Wno-unused-function -I include -DRES_STAMPS
$HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_res_stamps.so /tmp/tridiag_stamps.o \
  $(ls springcraft_amd/csrc/obj/*.o | grep -v tridiag.o)
echo springcraft_amd/libspringcraft_hip_res_stamps.so
