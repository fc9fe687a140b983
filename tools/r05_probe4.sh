#!/bin/bash
# compile-time variants of tools/probe_gemm3.hip: one line of flags per variant in $1 (file), plain runs, K = 256 / 6144 cases shown
set -u
mkdir -p gpurun_out/r05_probe4
O=gpurun_out/r05_probe4
: > $O/variants.txt
i=0
while IFS= read -r flags; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $flags tools/probe_gemm3.hip -o /tmp/pg4_$i > $O/build_$i.txt 2>&1 || { echo "== [$flags] build failed" | tee -a $O/variants.txt; continue; }
  echo "== [$flags]" | tee -a $O/variants.txt
  timeout -k 10 200 /tmp/pg4_$i 2>&1 | grep -v "^device" | cut -c1-460 | tee -a $O/variants.txt
  if grep -q "Memory access fault" $O/variants.txt; then echo "FAULT - stopping"; exit 1; fi
done < "$1"
