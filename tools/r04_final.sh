#!/bin/bash
# Round-4 evidence on one GPU box, in parts (a gpurun call is limited to 20 minutes):
#   bash tools/r04_final.sh a   the -m gpu suite with durations, the four bench lines (C3 with the CPU baseline), latencies
#   bash tools/r04_final.sh b   rocprofv3 kernel statistics of the default bench command, PMC passes of k_bulge_pair at the
#                               benchmarked batch, per-step stamps of the pair chase (-DPAIR_STAMPS build)
#   bash tools/r04_final.sh c   two-rank shared-GPU rehearsal WITH the CPU baseline next to the one-rank line, the one- /
#                               two-stage crossover
#   bash tools/r04_final.sh d1 | d2   the chase sweep (d1) and the test matrix (tools/test_matrix.sh), in two halves
# Everything lands in gpurun_out/r04_final/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_final
mkdir -p $OUT
cd $ROOT
part=${1:-a}
if [ $part = a ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -q --durations=15 > $OUT/gputest_durations.txt 2>&1 || { tail -30 $OUT/gputest_durations.txt; exit 1; }
  tail -3 $OUT/gputest_durations.txt
  timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python tools/show_bench.py $OUT/bench.json
  for c in c2 c4 c5; do
    timeout -k 10 300 python bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err || { tail -5 $OUT/bench_$c.err; exit 1; }
    python tools/show_bench.py $OUT/bench_$c.json
  done
  timeout -k 10 200 python tools/latency_phases.py > $OUT/latency.txt 2>&1 || exit 1
  grep -v amdgpu.ids $OUT/latency.txt
elif [ $part = b ]; then
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench --output-format csv -- \
     python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err) || { tail -5 $OUT/bench_under_rocprof.err; exit 1; }
  rm -f $OUT/prof/*kernel_trace.csv $OUT/prof/*/*kernel_trace.csv
  cp $(ls $OUT/prof/*kernel_stats.csv $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_bench.csv
  head -8 $OUT/rocprofv3_kernel_stats_bench.csv
  bash tools/pmc_kernel.sh k_bulge_pair r04_bulge || exit 1
  cp -r gpurun_out/pmc_r04_bulge $OUT/ 2>/dev/null
  [ -f $ROOT/springcraft_amd/libspringcraft_hip_stamps.so ] || bash tools/build_stamps_lib.sh > /dev/null   # (hipcc cross-compiles: also before the call)
  SPRINGCRAFT_HIP_LIB=$ROOT/springcraft_amd/libspringcraft_hip_stamps.so timeout -k 10 200 python tools/pair_stamps.py 2000 64 > $OUT/pair_stamps.txt 2>&1
  grep -v amdgpu.ids $OUT/pair_stamps.txt
elif [ $part = c ]; then
  # the same (small) workload on one rank and on two ranks sharing the GPU: the CPU baseline must come out the same
  timeout -k 10 300 python bench.py --structures-per-gpu 8 --steps 2 --warmup 1 > $OUT/rehearsal_1rank.json 2> $OUT/rehearsal_1rank.err || { tail -5 $OUT/rehearsal_1rank.err; exit 1; }
  SPRINGCRAFT_BENCH_SHARE_GPUS=1 timeout -k 10 400 python bench.py --gpus 2 --structures-per-gpu 8 --steps 2 --warmup 1 > $OUT/rehearsal_2ranks.json 2> $OUT/rehearsal_2ranks.err || { tail -5 $OUT/rehearsal_2ranks.err; exit 1; }
  python - $OUT/rehearsal_1rank.json $OUT/rehearsal_2ranks.json <<'PY'
import json, sys
for p in sys.argv[1:]:
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    c = d["cpu_baseline"]
    print(p.split("/")[-1], "n_gpus", d["n_gpus"], "value", d["value"], "host_binding", d["config"]["host_binding"], "| cpu_baseline", c["value"], "cores", c["cores"], c["sample"][:90])
PY
  SPRINGCRAFT_BENCH_SHARE_GPUS=1 timeout -k 10 300 python bench.py --config c4 --gpus 2 --structures-per-gpu 8 --steps 2 --warmup 1 > $OUT/rehearsal_c4_2ranks.json 2> $OUT/rehearsal_c4_2ranks.err || { tail -5 $OUT/rehearsal_c4_2ranks.err; exit 1; }
  python tools/show_bench.py $OUT/rehearsal_c4_2ranks.json
  timeout -k 10 500 python tools/crossover.py > $OUT/two_stage_crossover.txt 2>&1
  cat $OUT/two_stage_crossover.txt
elif [ $part = d1 ]; then
  timeout -k 10 500 python tools/bulge_sweep.py > $OUT/bulge_sweep.txt 2>&1
  cat $OUT/bulge_sweep.txt
  bash tools/test_matrix.sh 1 > $OUT/test_matrix_1.txt 2>&1
  cat $OUT/test_matrix_1.txt
else
  bash tools/test_matrix.sh 2 > $OUT/test_matrix_2.txt 2>&1
  cat $OUT/test_matrix_2.txt
fi
echo "part $part done"
