#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r04o}
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_two_stage_gpu.py tests/test_batched_configs_gpu.py tests/test_ragged_gpu.py -m gpu -x -q > $OUT/gputest.txt 2>&1 || { tail -30 $OUT/gputest.txt; exit 1; }
tail -2 $OUT/gputest.txt
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; return 1; }
  python - $OUT/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=d["phases_ms_profiled_step"]
print(sys.argv[2], "ms/step", d["ms_per_step"], "bulge", round(t.get("bulge_chasing_ms",0),1), "band", round(t.get("band_reduction_ms",0),1), "counters", d["counters"])
PY
}
run c4_early X=1 -- --config c4 &&
run c4_noearly SPRINGCRAFT_BULGE_NO_EARLY=1 -- --config c4 &&
run c2_early X=1 -- --config c2 &&
run c2_noearly SPRINGCRAFT_BULGE_NO_EARLY=1 -- --config c2
timeout -k 10 200 python tools/latency_phases.py 2>&1 | grep -v amdgpu | grep "two_stage=True"
echo "session done"
