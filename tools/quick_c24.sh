#!/bin/bash
# quick look at the latency-bound configurations (C4, C2) and one N = 2000 structure on the two-stage path
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04q
timeout -k 10 400 python -m pytest tests/test_two_stage_gpu.py tests/test_batched_configs_gpu.py -m gpu -x -q 2>&1 | tail -1
for c in c4 c2; do
timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r04q/$c.json 2>gpurun_out/r04q/err.txt && python tools/show_bench.py gpurun_out/r04q/$c.json | sed 's/.*modes\/s //'
done
timeout -k 10 200 python tools/latency_phases.py 2>&1 | grep "two_stage=True" | cut -c1-60
