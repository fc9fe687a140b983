#!/bin/bash
# the other BASELINE configurations (no CPU baseline), phases of the profiled step
set -u
mkdir -p gpurun_out/r05_cfg
O=gpurun_out/r05_cfg
for c in ${CFGS:-c5 c4 c2}; do
  for e in ${ENVS:-SPRINGCRAFT_GEMM3=1}; do
  env $e timeout -k 10 400 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/$c.json 2> $O/$c.err; echo "[$c $e] rc $?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/$c.json") if l.startswith('{')][-1])
ph=d['phases_ms_profiled_step']
print("   %s: %.1f %s, ms/step %.1f" % ("$c", d['value'], d['unit'], d['ms_per_step']))
print("   ", {k: round(v,1) for k,v in ph.items() if isinstance(v,(int,float)) and k.endswith('_ms')})
print("   ", d.get('counters'))
PY
  done
done
