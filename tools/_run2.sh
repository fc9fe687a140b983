set -e
for b in 8 16 32 128; do
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --structures-per-gpu $b > gpurun_out/bs_$b.json 2> gpurun_out/bs_$b.err
done
SPRINGCRAFT_NO_AUX=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/noaux.json 2> gpurun_out/noaux.err
python - <<'PY'
import json
for f in ["bs_8","bs_16","bs_32","bs_128","noaux"]:
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    p=d["phases_ms_profiled_step"]
    print(f, d["ms_per_step"], {k:round(v,1) for k,v in p.items() if isinstance(v,float)})
PY
