#!/bin/bash
# k_gemm3 in the library: its unit tests, the eigensolver tests that run through it, a C3 bench A/B (SPRINGCRAFT_GEMM3 = 0 / 1)
set -u
mkdir -p gpurun_out/r05_g3
O=gpurun_out/r05_g3
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -x -q -k gemm3 > $O/test_gemm3.txt 2>&1; echo "rc $?" >> $O/test_gemm3.txt; tail -5 $O/test_gemm3.txt
if grep -q "Memory access fault" $O/test_gemm3.txt; then exit 1; fi
grep -q "rc 0" $O/test_gemm3.txt || exit 1
SPRINGCRAFT_GEMM3=2 timeout -k 10 900 python -m pytest tests/test_two_stage_gpu.py tests/test_batched_configs_gpu.py -x -q > $O/test_solver_gemm3.txt 2>&1; echo "rc $?" >> $O/test_solver_gemm3.txt; tail -5 $O/test_solver_gemm3.txt
grep -q "rc 0" $O/test_solver_gemm3.txt || exit 1
for v in 0 1; do
  SPRINGCRAFT_GEMM3=$v timeout -k 10 600 python bench.py --steps 4 --warmup 2 > $O/bench_gemm3_$v.json 2> $O/bench_gemm3_$v.err; echo "bench $v rc $?"
  python tools/show_bench.py $O/bench_gemm3_$v.json 2>/dev/null | head -30
done
