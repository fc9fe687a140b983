#!/bin/bash
# Round 4: A/B of the bulge-chase forms on one box (pair form vs one sweep per workgroup vs per-wavefront launches) plus
# the per-step stamps of the pair form from the -DPAIR_STAMPS build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r04c}
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_two_stage_gpu.py tests/test_batched_configs_gpu.py -m gpu -x -q > $OUT/gputest.txt 2>&1 || { tail -30 $OUT/gputest.txt; exit 1; }
tail -2 $OUT/gputest.txt
run() {  # name, env..., -- args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; return 1; }
  python - $OUT/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t=d["phases_ms_profiled_step"]
print(sys.argv[2], "ms/step", d["ms_per_step"], "bulge", round(t.get("bulge_chasing_ms",0),1), "band", round(t.get("band_reduction_ms",0),1), "counters", d["counters"])
PY
}
run c3_default X=1 -- &&
run c3_pair SPRINGCRAFT_BULGE_PERSISTENT=2 -- &&
run c4_pair X=1 -- --config c4 &&
run c4_chase SPRINGCRAFT_BULGE_PAIR=0 -- --config c4 &&
run c2_pair X=1 -- --config c2 &&
run c2_chase SPRINGCRAFT_BULGE_PAIR=0 -- --config c2
exit 0
(timeout -k 10 200 python tools/pair_stamps.py 2000 64 && timeout -k 10 100 python tools/pair_stamps.py 1000 32 && timeout -k 10 100 python tools/pair_stamps.py 2000 8) > $OUT/pair_stamps.txt 2>&1
grep -v amdgpu.ids $OUT/pair_stamps.txt
echo "session done"
