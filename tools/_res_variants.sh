#!/bin/bash
# builds variants of the library with RES_* macros: bash tools/_res_variants.sh name "-DFLAG ..." ...
set -eu
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  $HIPCC -c springcraft_amd/csrc/tridiag.hip -o /tmp/tridiag_$name.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I include $flags
  $HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_var_$name.so /tmp/tridiag_$name.o $(ls springcraft_amd/csrc/obj/*.o | grep -v tridiag.o)
  echo built $name
done
