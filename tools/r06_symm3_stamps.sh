#!/bin/bash
# round 6: how long k_symm3's loader waves wait for their LDS-DMA groups (diagnostic build, -DSYMM3_STAMPS)
# build here (no GPU):  bash tools/r06_symm3_stamps.sh build ; on the GPU box:  bash tools/r06_symm3_stamps.sh
cd ${GRAFT_REPO_ROOT:-.}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if [ "$1" = build ]; then
  python springcraft_amd/csrc/build.py > /dev/null
  $HIPCC -c springcraft_amd/csrc/symm3.hip -o /tmp/symm3_stamps.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -I include -DSYMM3_STAMPS
  $HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_symm3_stamps.so /tmp/symm3_stamps.o $(ls springcraft_amd/csrc/obj/*.o | grep -v symm3.o)
  ls -la springcraft_amd/libspringcraft_hip_symm3_stamps.so
  exit 0
fi
mkdir -p gpurun_out/r06
SPRINGCRAFT_HIP_LIB=$PWD/springcraft_amd/libspringcraft_hip_symm3_stamps.so timeout -k 10 300 python tools/symm3_stamps.py 2000 64 2>gpurun_out/r06/symm3_stamps_err.txt | tee gpurun_out/r06/symm3_stamps.txt
