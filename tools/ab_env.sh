#!/bin/bash
# A/B on one GPU box through environment switches: tools/ab_env.sh [bench args --] "VAR=0" "VAR=1" ...  (each run twice,
# interleaved; boxes differ by a few percent, runs on one box by ~1 %)
set -u
cd ${GRAFT_REPO_ROOT:-.}
args=""
if [[ "${1:-}" == --* ]]; then args="$1"; shift; fi
for rep in 1 2; do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 $args > gpurun_out/ab_env.json 2>/dev/null
    echo "[$v] $(python tools/show_bench.py gpurun_out/ab_env.json)"
  done
done
