#!/bin/bash
# does the stamped build of k_bt2_role (its waits for the LDS counter at the stamps) run the bench step faster than the plain one?
set -eu
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -c springcraft_amd/csrc/twostage.hip -o /tmp/twostage_stamps.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall \
  -Wno-unused-function -I include -mllvm -pragma-unroll-threshold=1000000 -DBT2_STAMPS -DBT2_ROLE_STAMPS
$HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_stamps.so /tmp/twostage_stamps.o \
  $(ls springcraft_amd/csrc/obj/*.o | grep -v twostage.o)
export SPRINGCRAFT_BT2_ROLE=1
bash tools/r05_ab_lib.sh springcraft_amd/libspringcraft_hip.so springcraft_amd/libspringcraft_hip_stamps.so 2
