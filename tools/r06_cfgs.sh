#!/bin/bash
# round 6: the other BASELINE configurations under environment settings (same box):  ENVS="A=1 B=2" CFGS="c5 c2" bash tools/r06_cfgs.sh
set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06_cfg
O=gpurun_out/r06_cfg
IFS=';' read -ra EL <<< "${ENVS:-X=0}"
for c in ${CFGS:-c5 c4 c2}; do
  for e in "${EL[@]}"; do
  env $e timeout -k 10 400 python bench.py --config $c --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline > $O/$c.json 2> $O/$c.err; echo "[$c $e] rc $?"
  python - <<PY
import json
d=json.loads([l for l in open("$O/$c.json") if l.startswith('{')][-1])
ph=d['phases_ms_profiled_step']
r=d.get('rooflines',{})
print("   %s: %.1f %s, ms/step %.1f" % ("$c", d['value'], d['unit'], d['ms_per_step']))
print("   ", {k: round(v,1) for k,v in ph.items() if isinstance(v,(int,float)) and k.endswith('_ms')})
print("   ", {k: (v.get('frac'), v.get('ms')) for k,v in r.items() if isinstance(v,dict)})
PY
  done
done
