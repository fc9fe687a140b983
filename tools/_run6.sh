run() {
  for sh in "24000 24000 128 1 11 1" "48000 6000 256 0 11 1" "48000 6000 128 0 12 1" "24000 24000 128 1 12 1"; do
    echo "$sh: $(timeout -k 10 120 python tools/gemm_pmc.py $sh 2>&1 | tail -1)"
  done
}
echo "== normal"; run
SC_EXTRA_HIPCC_FLAGS=-DGEMM_C_NT python springcraft_amd/csrc/build.py --force > /dev/null
echo "== C non-temporal"; run
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_p.json 2> gpurun_out/r02_bench_p.err; python tools/show_bench.py gpurun_out/r02_bench_p.json
