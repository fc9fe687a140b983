#!/bin/bash
# stamps build of the role-split GEMM probe only
set -u
mkdir -p gpurun_out/r05_probe1
O=gpurun_out/r05_probe1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DSTAMPS ${EXTRA:-} tools/probe_gemm3.hip -o /tmp/probe_gemm3s > $O/build.txt 2>&1 || { cat $O/build.txt; exit 1; }
timeout -k 10 300 /tmp/probe_gemm3s > $O/probe_gemm3_stamps.txt 2>&1; echo "probe rc $?" >> $O/probe_gemm3_stamps.txt
cat $O/probe_gemm3_stamps.txt
