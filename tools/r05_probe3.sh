#!/bin/bash
# C-wave variants of the role-split GEMM probe (tools/probe_gemm3.hip, CW_VARIANT), plain and stamps builds
set -u
mkdir -p gpurun_out/r05_probe3
O=gpurun_out/r05_probe3
for v in ${VARIANTS:-0 1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DCW_VARIANT=$v ${EXTRA:-} tools/probe_gemm3.hip -o /tmp/pg3_v$v > $O/build_$v.txt 2>&1 || { cat $O/build_$v.txt; continue; }
  echo "== CW_VARIANT $v" | tee -a $O/variants.txt
  timeout -k 10 200 /tmp/pg3_v$v 2>&1 | grep -v "^device" | tee -a $O/variants.txt
done
if [ -n "${STAMPV:-}" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DSTAMPS -DCW_VARIANT=$STAMPV ${EXTRA:-} tools/probe_gemm3.hip -o /tmp/pg3_s > $O/build_s.txt 2>&1
  echo "== stamps, CW_VARIANT $STAMPV" | tee -a $O/variants.txt
  timeout -k 10 200 /tmp/pg3_s 2>&1 | grep -v "^device" | cut -c1-400 | tee -a $O/variants.txt
fi
