#!/bin/bash
# round 6: k_bulge_pair -- the tests that run the pair form (default, loader waves, no early look), then a same-box A/B
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
for env in "X=0" "SPRINGCRAFT_PAIR_LOADER=1" "SPRINGCRAFT_PAIR_EARLY=0"; do
env $env timeout -k 10 400 python -m pytest tests/test_two_stage_gpu.py -m gpu -x -q -k "pair or chase" > gpurun_out/r06/pair_tests.txt 2>&1
echo "[$env] two-stage tests rc $? $(tail -1 gpurun_out/r06/pair_tests.txt)"
env $env timeout -k 10 300 python -m pytest tests/test_batched_configs_gpu.py -m gpu -x -q -k "pair_chase" > gpurun_out/r06/pair_tests2.txt 2>&1
echo "[$env] c3 pair test rc $? $(tail -1 gpurun_out/r06/pair_tests2.txt)"
done
for r in 1 2; do
for env in "SPRINGCRAFT_PAIR_EARLY=0" "SPRINGCRAFT_PAIR_EARLY=1" "SPRINGCRAFT_PAIR_LOADER=1"; do
env $env timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06/ab.json 2>gpurun_out/r06/err.txt && echo "[$env] $(python tools/show_bench.py gpurun_out/r06/ab.json)"
done
done
