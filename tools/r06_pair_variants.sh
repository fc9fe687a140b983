#!/bin/bash
# round 6: diagnostic variants of k_bulge_pair (stamps build).  Build here (no GPU needed):  bash tools/r06_pair_variants.sh build
# run on the GPU box:  bash tools/r06_pair_variants.sh
cd ${GRAFT_REPO_ROOT:-.}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if [ "$1" = build ]; then
  python springcraft_amd/csrc/build.py > /dev/null
  for var in CAP168 NODMA; do
    $HIPCC -c springcraft_amd/csrc/twostage.hip -o /tmp/twostage_$var.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall \
      -Wno-unused-function -I include -mllvm -pragma-unroll-threshold=1000000 -DPAIR_STAMPS -DPAIR_VAR_$var &
  done
  wait
  for var in CAP168 NODMA; do
    $HIPCC -shared -fPIC --offload-arch=gfx950 -o springcraft_amd/libspringcraft_hip_var_$var.so /tmp/twostage_$var.o \
      $(ls springcraft_amd/csrc/obj/*.o | grep -v twostage.o)
  done
  ls -la springcraft_amd/*.so
  exit 0
fi
mkdir -p gpurun_out/r06
run() {  # lib, loader
  SPRINGCRAFT_PAIR_LOADER=$2 SPRINGCRAFT_HIP_LIB=$PWD/springcraft_amd/$1 timeout -k 10 300 python tools/pair_stamps.py 2000 64 2>gpurun_out/r06/var_err.txt
}
if [ -f springcraft_amd/libspringcraft_hip_var_CAP168.so ]; then
echo "== 512-thread form compiled for 168 registers"; run libspringcraft_hip_var_CAP168.so 0
echo "== loader waves present, nothing requested (results wrong by construction)"; run libspringcraft_hip_var_NODMA.so 1
fi
