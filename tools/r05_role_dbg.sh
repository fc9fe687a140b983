#!/bin/bash
# which part of k_bt2_role's Z waves breaks config 4?  (SPRINGCRAFT_BT2_ROLE_DBG bit 0: plain stores, bit 1: general DMA path)
mkdir -p gpurun_out/r05_role
for d in 0 1 2 3; do
  SPRINGCRAFT_BT2_ROLE=1 SPRINGCRAFT_BT2_ROLE_DBG=$d timeout -k 10 200 python -m pytest tests/test_batched_configs_gpu.py -x -q -k "config4 or config3_batched" 2>&1 | grep -E "passed|failed" | sed "s/^/dbg $d: /"
done
