#!/bin/bash
# round 6: k_symm3 in the band reduction -- GEMM unit tests, the two-stage solver tests, then a same-box A/B of the bench step
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
timeout -k 10 500 python -m pytest tests/test_gemm_gpu.py tests/test_two_stage_gpu.py -m gpu -x -q -k "symm3 or two_stage or random or special or golden or batched" > gpurun_out/r06/t_symm3_all.txt 2>&1
echo "tests rc $? $(tail -1 gpurun_out/r06/t_symm3_all.txt)"
for r in 1 2; do
for e in "SPRINGCRAFT_SYMM3=0" "SPRINGCRAFT_SYMM3=1" "SPRINGCRAFT_SYMM3=1 SPRINGCRAFT_SYMM3_WGS=256"; do
env $e timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06/ab.json 2>gpurun_out/r06/err.txt && echo "[$e] $(python tools/show_bench.py gpurun_out/r06/ab.json | sed 's/.*modes\/s //') $(python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/r06/ab.json') if l.startswith('{')][-1]
r=d.get('rooflines',{})
print('symm', r.get('symm',{}).get('frac'), r.get('symm',{}).get('ms'), 'syr2k', r.get('syr2k',{}).get('frac'))
PY
)"
done
done
