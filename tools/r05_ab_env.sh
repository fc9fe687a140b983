#!/bin/bash
# same-box A/B of the C3 bench step under two environment settings:  bash tools/r05_ab_env.sh "VAR=a" "VAR=b" [steps]
set -u
mkdir -p gpurun_out/r05_ab
O=gpurun_out/r05_ab
S=${3:-4}
i=0
for e in "$1" "$2" "$1" "$2"; do
  i=$((i+1))
  env $e timeout -k 10 600 python bench.py --steps $S --warmup 2 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; echo "[$e] rc $?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_$i.json") if l.startswith('{')][-1])
ph=d['phases_ms_profiled_step']
print("   ms/step %.1f | profiled: band %.0f (symm %.0f syr2k %.0f qr %.0f) chase %.0f dc %.0f (gemm %.0f) bt2 %.0f bt1_w %.0f bt1_update %.0f | gemm3 %s" % (
  d['ms_per_step'], ph['band_reduction_ms'], ph['symm_ms'], ph['syr2k_ms'], ph['panel_qr_ms'], ph['bulge_chasing_ms'], ph['tridiag_eigen_ms'], ph['dc_gemm_ms'], ph['bt2_apply_ms'], ph['bt1_w_ms'], ph['bt1_update_ms'], d['counters'].get('gemm3_launches')))
PY
done
