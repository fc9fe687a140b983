#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r04f}
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1 || { tail -30 $OUT/gputest.txt; exit 1; }
tail -2 $OUT/gputest.txt
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err || { tail -5 $OUT/c3.err; exit 1; }
python tools/show_bench.py $OUT/c3.json
export SPRINGCRAFT_HIP_LIB=$ROOT/springcraft_amd/libspringcraft_hip_stamps.so
(timeout -k 10 200 python tools/pair_stamps.py 2000 64 && timeout -k 10 100 python tools/pair_stamps.py 1000 32 && timeout -k 10 100 python tools/pair_stamps.py 2000 8) > $OUT/pair_stamps.txt 2>&1
grep -v amdgpu.ids $OUT/pair_stamps.txt
echo "session done"
