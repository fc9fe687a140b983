#!/bin/bash
# same-box A/B of two prebuilt libraries on the default bench step:  bash tools/quick_ab_lib.sh <lib a> <lib b>
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04q
for r in 1 2; do
for lib in "$@"; do
SPRINGCRAFT_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04q/ab.json 2>gpurun_out/r04q/err.txt && echo "[$lib] $(python tools/show_bench.py gpurun_out/r04q/ab.json | sed 's/.*modes\/s //')"
done
done
