#!/bin/bash
# per-phase cycles of k_bulge_chase at the shapes of C4, C2 and one N = 2000 structure (diagnostic library, -DCHASE_STAMPS)
set -eu
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
mkdir -p gpurun_out/r05_final
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -c springcraft_amd/csrc/twostage.hip -o /tmp/twostage_cs.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall \
  -Wno-unused-function -I include -mllvm -pragma-unroll-threshold=1000000 -DCHASE_STAMPS
$HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_stamps.so /tmp/twostage_cs.o \
  $(ls springcraft_amd/csrc/obj/*.o | grep -v twostage.o)
rm -f gpurun_out/r05_final/chase_stamps.txt
for shape in "1000 32" "512 64" "2000 1"; do
  SPRINGCRAFT_HIP_LIB=$PWD/springcraft_amd/libspringcraft_hip_stamps.so timeout -k 10 300 python tools/chase_stamps.py $shape 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_final/chase_stamps.txt
done
