#!/bin/bash
# Bench configuration <cfg> with twostage.hip rebuilt under each extra flag set:  tools/variants_cfg.sh c4 "-DSC_QR_IB=8" "-DSC_QR_IB=16"
set -u
cd ${GRAFT_REPO_ROOT:-.}
. tools/ab_lib.sh
CFG=$1; shift
ab_keep springcraft_amd/csrc/twostage.hip
for rep in 1 2; do
  for flags in "$@"; do
    touch springcraft_amd/csrc/twostage.hip
    ab_build "$flags" || continue
    timeout -k 10 120 python bench.py --config $CFG --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/varc.json 2>/dev/null
    echo "[$flags] $(python tools/show_bench.py gpurun_out/varc.json | sed 's/.*modes\/s //')"
  done
done
