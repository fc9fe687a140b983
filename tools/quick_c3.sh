#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04q
timeout -k 10 300 python -m pytest tests/test_two_stage_gpu.py -m gpu -x -q 2>&1 | tail -1
for r in 1 2; do
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04q/c3_$r.json 2>gpurun_out/r04q/err.txt && python tools/show_bench.py gpurun_out/r04q/c3_$r.json
done
