#!/bin/bash
# tools/ab_twostage.sh for another bench configuration:  tools/ab_twostage_cfg.sh c4 a.hip b.hip
set -u
cd ${GRAFT_REPO_ROOT:-.}
. tools/ab_lib.sh
CFG=$1; shift
ab_keep springcraft_amd/csrc/twostage.hip
for rep in 1 2; do
  for v in "$@"; do
    cp "$v" springcraft_amd/csrc/twostage.hip
    ab_build "" || continue
    timeout -k 10 120 python bench.py --config $CFG --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/ab.json 2>/dev/null
    echo "[$v] $(python tools/show_bench.py gpurun_out/ab.json)"
  done
done
