#!/bin/bash
# same-box A/B of environment settings on the default bench step:  bash tools/quick_env_ab.sh "A=1" "A=2 B=3" ...
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04q
for r in 1 2; do
for e in "$@"; do
env $e timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04q/ab.json 2>gpurun_out/r04q/err.txt && echo "[$e] $(python tools/show_bench.py gpurun_out/r04q/ab.json | sed 's/.*modes\/s //')"
done
done
