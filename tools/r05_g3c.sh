#!/bin/bash
set -u
mkdir -p gpurun_out/r05_g3
O=gpurun_out/r05_g3
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k gemm3 > $O/test_gemm3.txt 2>&1; echo "rc $?" >> $O/test_gemm3.txt; tail -3 $O/test_gemm3.txt
grep -q "rc 0" $O/test_gemm3.txt || exit 1
SPRINGCRAFT_GEMM3=2 timeout -k 10 900 python -m pytest tests/test_two_stage_gpu.py tests/test_batched_configs_gpu.py tests/test_eigh_gpu.py -x -q > $O/test_solver_gemm3.txt 2>&1; echo "rc $?" >> $O/test_solver_gemm3.txt; tail -3 $O/test_solver_gemm3.txt
grep -q "rc 0" $O/test_solver_gemm3.txt || exit 1
bash tools/r05_ab_env.sh SPRINGCRAFT_GEMM3_W=0 SPRINGCRAFT_GEMM3_W=1
