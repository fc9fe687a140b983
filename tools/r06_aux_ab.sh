#!/bin/bash
# round 6 (last session): the second stream's work beside the D&C -- on the main stream instead (SPRINGCRAFT_NO_AUX), and with
# the second stream below / above the main stream's priority (SPRINGCRAFT_AUX_PRIORITY = 1 / 2); bench step, same box
mkdir -p gpurun_out/ab
run() {  # label, config, env...
  local label=$1 cfg=$2; shift 2
  env "$@" timeout -k 10 100 python bench.py --config $cfg --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/ab/${cfg}_$label.json 2> gpurun_out/ab/${cfg}_$label.err || exit 1
}
for i in 1 2; do
  run def_$i c3 X=0
  run low_$i c3 SPRINGCRAFT_AUX_PRIORITY=1
  run high_$i c3 SPRINGCRAFT_AUX_PRIORITY=2
done
run noaux_1 c3 SPRINGCRAFT_NO_AUX=1
for cfg in c4 c2; do
  run def_1 $cfg X=0
  run low_1 $cfg SPRINGCRAFT_AUX_PRIORITY=1
  run def_2 $cfg X=0
  run low_2 $cfg SPRINGCRAFT_AUX_PRIORITY=1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/c*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); p=d.get('phases_ms_profiled_step',{})
    print(f.split('/')[-1], round(d['ms_per_step'],1), 'dc', round(p.get('tridiag_eigen_ms',0),1), 'bt', round(p.get('backtransform_ms',0),1), 'tf', round(p.get('dia_tfactor_ms',0),1))
PY
