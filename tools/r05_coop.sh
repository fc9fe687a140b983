#!/bin/bash
# k_panel_coop: its test, then C5 and the single-structure latencies with and without it
set -u
mkdir -p gpurun_out/r05_coop
O=gpurun_out/r05_coop
timeout -k 10 400 python -m pytest tests/test_two_stage_gpu.py -x -q -k "cooperative" > $O/test.txt 2>&1; echo "rc $?" >> $O/test.txt; tail -6 $O/test.txt
grep -q "rc 0" $O/test.txt || exit 1
if grep -q "Memory access fault" $O/test.txt; then exit 1; fi
for e in ${ENVS:-SPRINGCRAFT_QR_COOP=0 SPRINGCRAFT_QR_COOP=1}; do
  env $e timeout -k 10 300 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/c5_$e.json 2> $O/c5_$e.err; echo "[c5 $e] rc $?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/c5_$e.json") if l.startswith('{')][-1])
ph=d['phases_ms_profiled_step']
print("   ms/step %.1f" % d['ms_per_step'], {k: round(v,1) for k,v in ph.items() if k in ('band_reduction_ms','panel_qr_ms','symm_ms','syr2k_ms','bulge_chasing_ms')}, d.get('counters',{}).get('panel_coop_launches'), (d.get('parity_gates') or {}))
PY
  env $e timeout -k 10 200 python tools/latency_phases.py > $O/lat_$e.txt 2>&1; echo "[latency $e] rc $?"
  grep -v amdgpu.ids $O/lat_$e.txt | tail -12
done
