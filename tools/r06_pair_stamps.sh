#!/bin/bash
# round 6: per-phase stamps of k_bulge_pair with and without its loader waves (diagnostic library, -DPAIR_STAMPS)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
for v in 0 1; do
SPRINGCRAFT_PAIR_LOADER=$v SPRINGCRAFT_HIP_LIB=$PWD/springcraft_amd/libspringcraft_hip_stamps.so timeout -k 10 300 python tools/pair_stamps.py 2000 64 > gpurun_out/r06/pair_stamps_$v.txt 2>gpurun_out/r06/pair_stamps_err_$v.txt; echo "rc $?"; cat gpurun_out/r06/pair_stamps_$v.txt
done
