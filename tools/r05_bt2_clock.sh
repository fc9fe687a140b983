#!/bin/bash
# shader clock during k_bt2_apply and k_bt2_role (diagnostic library, -DBT2_CLOCK)
set -eu
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out/r05_role
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -c springcraft_amd/csrc/twostage.hip -o /tmp/twostage_clk.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall \
  -Wno-unused-function -I include -mllvm -pragma-unroll-threshold=1000000 -DBT2_CLOCK
$HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_stamps.so /tmp/twostage_clk.o \
  $(ls springcraft_amd/csrc/obj/*.o | grep -v twostage.o)
for r in 0 1; do
  SPRINGCRAFT_BT2_ROLE=$r SPRINGCRAFT_HIP_LIB=$PWD/springcraft_amd/libspringcraft_hip_stamps.so timeout -k 10 300 python tools/bt2_clock.py 24 2000 6 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_role/clock.txt
done
