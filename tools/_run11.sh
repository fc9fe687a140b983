timeout -k 10 300 python -m pytest tests/test_eigh_gpu.py tests/test_two_stage_gpu.py -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_v.json 2> gpurun_out/r02_bench_v.err; python tools/show_bench.py gpurun_out/r02_bench_v.json
SPRINGCRAFT_GEMM_NO_PAIR=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_w.json 2> gpurun_out/r02_bench_w.err; python tools/show_bench.py gpurun_out/r02_bench_w.json
python - <<'PY'
import json
for f in "vw":
    d=json.loads(open(f"gpurun_out/r02_bench_{f}.json").read().strip().splitlines()[-1]); p=d["phases_ms_profiled_step"]
    print(f, {k: round(p[k],1) for k in ("syr2k_ms","symm_ms","bt1_w_ms","bt1_update_ms","dc_gemm_ms","band_reduction_ms")})
PY
