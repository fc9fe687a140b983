#!/bin/bash
# The eigensolver tests under every switch that selects an alternative code path (on the GPU box):  bash tools/test_matrix.sh
set -u
T="tests/test_two_stage_gpu.py tests/test_eigh_gpu.py tests/test_batched_configs_gpu.py tests/test_gemm_gpu.py"
# (in parts, a gpurun call is limited to 20 minutes:  bash tools/test_matrix.sh <part> [parts = 2]; no argument: everything)
HALF=${1:-0}
PARTS=${2:-2}
K=0
# (a failing row stays in the output with the names of the tests that failed: -rf prints them above the summary line)
run() { K=$((K+1)); if [ $HALF -ne 0 ] && [ $((K % PARTS)) -ne $((HALF % PARTS)) ]; then return; fi; echo "== $*"; env "$@" timeout -k 10 600 python -m pytest $T -x -q -rf 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|error" | tail -4; }
run SPRINGCRAFT_BULGE_PERSISTENT=0 SPRINGCRAFT_BULGE_STREAMS=1 SPRINGCRAFT_STAGE1_STREAMS=1
run SPRINGCRAFT_BULGE_PERSISTENT=0 SPRINGCRAFT_BULGE_STREAMS=3 SPRINGCRAFT_STAGE1_STREAMS=3
run SPRINGCRAFT_BULGE_PERSISTENT=2
run SPRINGCRAFT_GEMM_NO_LOWER_GRID=1 SPRINGCRAFT_GEMM_NO_BALANCE=1 SPRINGCRAFT_GEMM_NO_PAIR=1
run SPRINGCRAFT_GEMM2_TILE=3
run SPRINGCRAFT_GEMM2_TILE=1
run SPRINGCRAFT_BT2_NW=4
run SPRINGCRAFT_NO_AUX=1
run SPRINGCRAFT_QR_UNBLOCKED=1 SPRINGCRAFT_QR_WG=0
run SPRINGCRAFT_SYMM_SPLIT=4 SPRINGCRAFT_BT2_WAVE=1
run SPRINGCRAFT_BT2_WAVE=0
run SPRINGCRAFT_PANEL_PAIRS=0
run SPRINGCRAFT_QR_WG=0
run SPRINGCRAFT_QR_WG=1
run SPRINGCRAFT_BULGE_PERSISTENT=2 SPRINGCRAFT_BULGE_PAIR=2
run SPRINGCRAFT_BULGE_PAIR=0
run SPRINGCRAFT_TWO_STAGE=1
run SPRINGCRAFT_TWO_STAGE=0
run SPRINGCRAFT_GEMM3=0
run SPRINGCRAFT_GEMM3=2 SPRINGCRAFT_GEMM3_LOWER=1
run SPRINGCRAFT_GEMM3=2 SPRINGCRAFT_GEMM3_ORDER=0 SPRINGCRAFT_GEMM3_W=0
run SPRINGCRAFT_BULGE_NO_EARLY=1 SPRINGCRAFT_BULGE_PAIR=0
run SPRINGCRAFT_BULGE_PERSISTENT=2 SPRINGCRAFT_BULGE_PAIR=0 SPRINGCRAFT_BULGE_NO_EARLY=1
run SPRINGCRAFT_QR_COOP=0
run SPRINGCRAFT_QR_COOP_MIN=128
run SPRINGCRAFT_BT2_ROLE=1
