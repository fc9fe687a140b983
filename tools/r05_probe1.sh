#!/bin/bash
# round 5: the role-split GEMM probe (tools/probe_gemm3.hip) beside k_gemm2 on the same box
set -u
mkdir -p gpurun_out/r05_probe1
O=gpurun_out/r05_probe1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_gemm3.hip -o /tmp/probe_gemm3 > $O/build.txt 2>&1 || { cat $O/build.txt; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DSTAMPS tools/probe_gemm3.hip -o /tmp/probe_gemm3s >> $O/build.txt 2>&1 || { cat $O/build.txt; exit 1; }
timeout -k 10 300 /tmp/probe_gemm3 > $O/probe_gemm3.txt 2>&1; echo "probe rc $?" >> $O/probe_gemm3.txt
cat $O/probe_gemm3.txt
timeout -k 10 300 /tmp/probe_gemm3s > $O/probe_gemm3_stamps.txt 2>&1; echo "probe rc $?" >> $O/probe_gemm3_stamps.txt
cat $O/probe_gemm3_stamps.txt
if [ "${1:-}" != "nolib" ]; then
timeout -k 10 300 python tools/gemm3_compare.py > $O/gemm2_same_shapes.txt 2>&1; echo "rc $?" >> $O/gemm2_same_shapes.txt
cat $O/gemm2_same_shapes.txt
fi
