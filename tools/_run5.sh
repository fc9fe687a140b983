for pc in -1 0 3 4; do
  export SPRINGCRAFT_GEMM_WG_PER_CU=$pc
  echo "== WG_PER_CU=$pc"
  for sh in "24000 24000 128 1 11 1" "48000 6000 256 0 11 1" "48000 6000 128 0 12 1" "6000 6000 6000 0 11 0" "1030 517 333 0 10 1"; do
    echo "$sh: $(timeout -k 10 120 python tools/gemm_pmc.py $sh 2>&1 | tail -1)"
  done
done > gpurun_out/r02_gemm_persist.txt 2>&1
cat gpurun_out/r02_gemm_persist.txt
unset SPRINGCRAFT_GEMM_WG_PER_CU
timeout -k 10 300 python -m pytest tests/test_eigh_gpu.py tests/test_two_stage_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_o.json 2> gpurun_out/r02_bench_o.err; python tools/show_bench.py gpurun_out/r02_bench_o.json
