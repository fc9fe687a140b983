#!/bin/bash
set -u
mkdir -p gpurun_out/r05_g3
O=gpurun_out/r05_g3
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k gemm3 > $O/test_gemm3.txt 2>&1; echo "rc $?" >> $O/test_gemm3.txt; tail -3 $O/test_gemm3.txt
grep -q "rc 0" $O/test_gemm3.txt || exit 1
SPRINGCRAFT_GEMM3_ORDER=0 timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k gemm3 > $O/test_gemm3_flat.txt 2>&1; echo "rc $?" >> $O/test_gemm3_flat.txt; tail -2 $O/test_gemm3_flat.txt
timeout -k 10 600 python tools/gemm3_shapes.py 32 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee $O/shapes.txt
