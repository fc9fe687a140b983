#!/bin/bash
# A/B of two versions of ONE source file on one GPU box:  tools/ab_file.sh <path in tree> <version a> <version b> [bench args]
set -u
cd ${GRAFT_REPO_ROOT:-.}
. tools/ab_lib.sh
F=$1; A=$2; B=$3; shift 3
ab_keep $F          # restored (and the library rebuilt) on every exit
for rep in 1 2; do
  for v in $A $B; do
    cp "$v" $F
    ab_build "" || continue
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > gpurun_out/abf.json 2>/dev/null
    echo "[$v] $(python tools/show_bench.py gpurun_out/abf.json | sed 's/.*modes\/s //')"
  done
done
