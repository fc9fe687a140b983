#!/bin/bash
# Round-3 final numbers on one GPU box: the four bench lines, the single-structure latencies, then the rocprofv3
# kernel statistics of the default bench command.  Everything lands in gpurun_out/r03_final/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_final
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "c3 done"
for c in c2 c4 c5; do
  timeout -k 10 300 python bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err || exit 1
  echo "$c done"
done
timeout -k 10 200 python tools/latency_phases.py > $OUT/latency.txt 2>&1 || exit 1
echo "latency done"
bash tools/r03_profiles.sh stats || exit 1
cp gpurun_out/r03_prof/*stats*.csv $OUT/ 2>/dev/null
ls $OUT
