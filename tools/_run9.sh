run() {
  for sh in "24000 24000 128 1 11 1" "24000 24000 128 1 12 1" "6000 6000 128 1 10 1" "999 999 77 1 10 1"; do
    echo "$sh: $(timeout -k 10 120 python tools/gemm_pmc.py $sh 2>&1 | tail -1)"
  done
}
echo "== lower grid"; run
export SPRINGCRAFT_GEMM_NO_LOWER_GRID=1
echo "== rectangular grid"; run
unset SPRINGCRAFT_GEMM_NO_LOWER_GRID
timeout -k 10 300 python -m pytest tests/test_eigh_gpu.py tests/test_two_stage_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_r.json 2> gpurun_out/r02_bench_r.err; python tools/show_bench.py gpurun_out/r02_bench_r.json
SPRINGCRAFT_GEMM_NO_LOWER_GRID=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_s.json 2> gpurun_out/r02_bench_s.err; python tools/show_bench.py gpurun_out/r02_bench_s.json
python - <<'PY'
import json
for f in "rs":
    d=json.loads(open(f"gpurun_out/r02_bench_{f}.json").read().strip().splitlines()[-1]); p=d["phases_ms_profiled_step"]
    print(f, {k: round(p[k],1) for k in ("syr2k_ms","symm_ms","panel_qr_ms","band_reduction_ms")})
PY
