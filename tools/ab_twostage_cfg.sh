#!/bin/bash
# tools/ab_twostage.sh for another bench configuration:  tools/ab_twostage_cfg.sh c4 a.hip b.hip
set -u
cd ${GRAFT_REPO_ROOT:-.}
CFG=$1; shift
cp springcraft_amd/csrc/twostage.hip /tmp/twostage_keep.hip
for rep in 1 2; do
  for v in "$@"; do
    cp "$v" springcraft_amd/csrc/twostage.hip
    python springcraft_amd/csrc/build.py > /dev/null 2>&1
    timeout -k 10 120 python bench.py --config $CFG --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/ab.json 2>/dev/null
    echo "[$v] $(python tools/show_bench.py gpurun_out/ab.json)"
  done
done
cp /tmp/twostage_keep.hip springcraft_amd/csrc/twostage.hip
python springcraft_amd/csrc/build.py > /dev/null 2>&1
