#!/bin/bash
# Time the bench step with twostage.hip rebuilt under each of the given extra hipcc flag sets (GPU box), e.g.
#   tools/variants_bt2.sh "-DBT2_LAGSEL=1" "-DBT2_LAGSEL=2" "-DBT2_DBG=8"
# (BT2_DBG builds give wrong results by construction; the library is rebuilt without extra flags on exit; a flag set that
# does not compile is reported and skipped, never timed as the previous build.)
set -u
cd ${GRAFT_REPO_ROOT:-.}
. tools/ab_lib.sh
ab_keep springcraft_amd/csrc/twostage.hip
i=0
for flags in "$@"; do
  touch springcraft_amd/csrc/twostage.hip
  ab_build "$flags" || continue
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/var_$i.json 2>/dev/null
  echo "[$flags] $(python tools/show_bench.py gpurun_out/var_$i.json | sed 's/.*bt2 /bt2 /')"
  i=$((i+1))
done
