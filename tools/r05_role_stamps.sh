#!/bin/bash
# stamps of k_bt2_role's MFMA waves (diagnostic library)
set -eu
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out/r05_role
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -c springcraft_amd/csrc/twostage.hip -o /tmp/twostage_stamps.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall \
  -Wno-unused-function -I include -mllvm -pragma-unroll-threshold=1000000 -DBT2_STAMPS -DBT2_ROLE_STAMPS
$HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_stamps.so /tmp/twostage_stamps.o \
  $(ls springcraft_amd/csrc/obj/*.o | grep -v twostage.o)
SPRINGCRAFT_BT2_ROLE=1 SPRINGCRAFT_HIP_LIB=$PWD/springcraft_amd/libspringcraft_hip_stamps.so timeout -k 10 300 python tools/bt2_role_stamps.py 24 2000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_role/stamps.txt
