#!/bin/bash
# A/B on one GPU box: time the bench step with each of the given versions of twostage.hip in turn (twice, interleaved:
# boxes differ by a few percent, runs on one box by ~1 %); the tree's version is restored on every exit.
#   tools/ab_twostage.sh a.hip b.hip
set -u
cd ${GRAFT_REPO_ROOT:-.}
. tools/ab_lib.sh
ab_keep springcraft_amd/csrc/twostage.hip
for rep in 1 2; do
  for v in "$@"; do
    cp "$v" springcraft_amd/csrc/twostage.hip
    ab_build "" || continue
    timeout -k 10 120 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/ab.json 2>/dev/null
    echo "[$v] $(python tools/show_bench.py gpurun_out/ab.json | sed 's/.*tri /tri /')"
  done
done
