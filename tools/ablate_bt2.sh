#!/bin/bash
# Ablation of k_bt2_apply on the GPU box: rebuild twostage.hip with BT2_DBG = 1 (no fragment DMA), 4 (one MFMA in ten),
# 8 (no workgroup barriers), 16 (no fragment reads from LDS) or sums of these and time the bench step; results are wrong
# by construction.  (2 = no Z traffic lets the compiler delete the products as dead code: not meaningful.)
set -u
cd ${GRAFT_REPO_ROOT:-.}
. tools/ab_lib.sh
ab_keep springcraft_amd/csrc/twostage.hip          # (the file is not changed; the EXIT trap rebuilds the library without extra flags)
for d in ${1:-0 1 4 5}; do
  touch springcraft_amd/csrc/twostage.hip
  ab_build "-DBT2_DBG=$d" || continue
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/abl_$d.json 2>/dev/null
  echo "BT2_DBG=$d $(python tools/show_bench.py gpurun_out/abl_$d.json | sed 's/.*bt2 /bt2 /')"
done
