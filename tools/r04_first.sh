#!/bin/bash
# Round 4, first GPU session: the GPU suite on the new build, then A/B lines of the bulge-chase forms.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1 || { tail -30 $OUT/gputest.txt; exit 1; }
tail -3 $OUT/gputest.txt
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; return 1; }
  python tools/show_bench.py $OUT/$name.json 2>/dev/null | head -3 || true
  python - $OUT/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=d["phases_ms_profiled_step"]
print(sys.argv[2], "ms/step", d["ms_per_step"], "bulge", round(t.get("bulge_chasing_ms",0),1), "band", round(t.get("band_reduction_ms",0),1), "counters", d["counters"])
PY
}
run c3_default X=1 -- &&
run c3_pair SPRINGCRAFT_BULGE_PERSISTENT=2 -- &&
run c3_chase SPRINGCRAFT_BULGE_PERSISTENT=2 SPRINGCRAFT_BULGE_PAIR=0 -- &&
run c4_pair X=1 -- --config c4 &&
run c4_chase SPRINGCRAFT_BULGE_PAIR=0 -- --config c4 &&
run c2_pair X=1 -- --config c2 &&
run c2_chase SPRINGCRAFT_BULGE_PAIR=0 -- --config c2
echo "session done"
