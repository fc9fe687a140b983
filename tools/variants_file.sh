#!/bin/bash
# Time the bench step with ONE source file rebuilt under each of the given extra hipcc flag sets (GPU box):
#   tools/variants_file.sh springcraft_amd/csrc/stedc.hip "" "-DDC_EXACT_DIV"
# (the library is rebuilt without extra flags at the end, also when the run is interrupted)
set -u
cd ${GRAFT_REPO_ROOT:-.}
. tools/ab_lib.sh
F=$1; shift
ab_keep $F
i=0
for rep in 1 2; do
  for flags in "$@"; do
    touch $F
    ab_build "$flags" || continue
    timeout -k 10 120 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/varf_$i.json 2>/dev/null
    echo "[$flags] $(python tools/show_bench.py gpurun_out/varf_$i.json | sed 's/.*modes\/s //')"
    i=$((i+1))
  done
done
