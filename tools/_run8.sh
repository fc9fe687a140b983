run() {
  for sh in "24000 24000 128 1 11 1" "24000 24000 128 1 12 1" "48000 6000 256 0 11 1" "48000 6000 128 0 12 1" "6000 6000 6000 0 11 0" "1030 517 333 0 10 1" "999 999 77 1 10 1"; do
    echo "$sh: $(timeout -k 10 120 python tools/gemm_pmc.py $sh 2>&1 | tail -1)"
  done
}
for g in 3 6 0 2 4; do export SPRINGCRAFT_GEMM_GM_LOG2=$g; echo "== GM_LOG2=$g"; run; done
unset SPRINGCRAFT_GEMM_GM_LOG2
timeout -k 10 300 python -m pytest tests/test_eigh_gpu.py tests/test_two_stage_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_q.json 2> gpurun_out/r02_bench_q.err; python tools/show_bench.py gpurun_out/r02_bench_q.json
