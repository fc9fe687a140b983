#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r04k}
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1 || { tail -30 $OUT/gputest.txt; exit 1; }
tail -2 $OUT/gputest.txt
for v in default 1 default 1; do
  if [ $v = default ]; then unset SPRINGCRAFT_QR_WG; else export SPRINGCRAFT_QR_WG=$v; fi
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c3_qr_$v.json 2> $OUT/c3.err || { tail -5 $OUT/c3.err; exit 1; }
  python - $OUT/c3_qr_$v.json "QR_WG=$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=d["phases_ms_profiled_step"]
print(sys.argv[2], "ms/step", d["ms_per_step"], "panel_qr", round(t.get("panel_qr_ms",0),1), "band", round(t.get("band_reduction_ms",0),1), "bulge", round(t.get("bulge_chasing_ms",0),1), "dc_gemm", round(t.get("dc_gemm_ms",0),1), "gflop", round(t.get("dc_gemm_gflop",0),1))
PY
done
echo "session done"
