#!/bin/bash
# the partial-spectrum path: its tests, then the C5 line with phases
set -u
mkdir -p gpurun_out/r05_c5
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "subset or range or config5 or partial or c5" 2>&1 | tail -3
timeout -k 10 300 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05_c5/c5.json 2> gpurun_out/r05_c5/c5.err; echo "rc $?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r05_c5/c5.json") if l.startswith('{')][-1])
ph=d['phases_ms_profiled_step']
print("ms/step %.1f" % d['ms_per_step'], {k: round(v,1) for k,v in ph.items() if k in ('band_reduction_ms','bulge_chasing_ms','sturm_ms','stein_ms','cholqr_ms','tridiag_eigen_ms','backtransform_ms')}, d.get('parity_gates'))
PY
