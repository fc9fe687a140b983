#!/bin/bash
# k_panel_coop on every test of the eigensolver (forced down to 128-row panels), then C5 / latencies / C3 by the default rule
set -u
mkdir -p gpurun_out/r05_coop
O=gpurun_out/r05_coop
SPRINGCRAFT_QR_COOP_MIN=128 timeout -k 10 800 python -m pytest tests/test_two_stage_gpu.py tests/test_eigh_gpu.py tests/test_batched_configs_gpu.py -x -q -rf > $O/test_all.txt 2>&1; echo "rc $?" >> $O/test_all.txt; tail -5 $O/test_all.txt
grep -q "rc 0" $O/test_all.txt || exit 1
ENVS="SPRINGCRAFT_QR_COOP=1" bash tools/r05_coop.sh
