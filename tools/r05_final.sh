#!/bin/bash
# Round-5 evidence on one GPU box, in parts (a gpurun call is limited to 20 minutes):
#   bash tools/r05_final.sh a   the -m gpu suite with durations, the four bench lines (C3 with the CPU baseline), latencies
#   bash tools/r05_final.sh b   rocprofv3 kernel statistics of the default bench command; PMC passes (counters only, separate
#                               runs) of k_bt2_apply -- bytes and MFMA-pipe occupancy at the benchmarked batch -- and the
#                               MFMA-pipe occupancy of k_gemm3
#   bash tools/r05_final.sh b2  rocprofv3 kernel statistics of bench.py --config c5
#   bash tools/r05_final.sh c1 | c2 | c3   the test matrix (tools/test_matrix.sh), in three parts
#   bash tools/r05_final.sh d   k_gemm3 beside k_gemm2 on the batched shapes of the step, the chase sweep with the points
#                               beyond the round-4 range, one-stage against two-stage over (N, batch), per-phase stamps of
#                               k_bulge_chase at the shapes of C4, C2 and one N = 2000 structure
#   bash tools/r05_final.sh e   k_panel_coop on / off (C5, single-structure latencies); k_bt2_role beside k_bt2_apply on the
#                               bench step, the stamps of its MFMA waves, the shader clock under both kernels
# Everything lands in gpurun_out/r05_final/; what is to be judged is copied to profiles/r05_* (profiles/README.md).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05_final
mkdir -p $OUT
cd $ROOT
part=${1:-a}
if [ $part = a ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q --durations=15 > $OUT/gputest_durations.txt 2>&1 || { tail -30 $OUT/gputest_durations.txt; exit 1; }
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
  bash tools/pmc_kernel.sh k_bt2_apply r05_bt2 || exit 1
  bash tools/pmc_mfma.sh k_bt2_apply r05_bt2 > $OUT/bt2_pmc_mfma.txt 2>&1 || exit 1
  bash tools/pmc_mfma.sh k_gemm3 r05_gemm3 > $OUT/gemm3_pmc_mfma.txt 2>&1 || exit 1
  grep -h run_mfma $OUT/bt2_pmc_mfma.txt $OUT/gemm3_pmc_mfma.txt | cut -c1-400
  python3 tools/pmc_to_json.py gpurun_out/pmc_r05_bt2 k_bt2_apply 6000 64 > $OUT/bt2_pmc_fetch_write.json
  grep -E "traffic_over|l2_hit|hbm_bytes" $OUT/bt2_pmc_fetch_write.json
  python3 tools/pmc_summary.py gpurun_out/pmc_r05_bt2 > $OUT/bt2_pmc_summary.txt 2>&1
elif [ $part = b2 ]; then
  # per-kernel totals of the C5 line (one n = 24 000 matrix, 106 modes): k_panel_coop among them
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o bench --output-format csv -- \
     python3 $ROOT/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_under_rocprof.json 2> $OUT/bench_c5_under_rocprof.err) || { tail -5 $OUT/bench_c5_under_rocprof.err; exit 1; }
  rm -f $OUT/prof_c5/*kernel_trace.csv $OUT/prof_c5/*/*kernel_trace.csv
  cp $(ls $OUT/prof_c5/*kernel_stats.csv $OUT/prof_c5/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_bench_c5.csv
  head -12 $OUT/rocprofv3_kernel_stats_bench_c5.csv | cut -c1-220
elif [ $part = c1 ] || [ $part = c2 ] || [ $part = c3 ]; then
  k=${part#c}
  bash tools/test_matrix.sh $k 3 > $OUT/test_matrix_$k.txt 2>&1
  cat $OUT/test_matrix_$k.txt
elif [ $part = c4 ]; then
  # the row that failed in part c2 (STAGE1_STREAMS=3: the part of the batch on the main stream was factored by k_panel_coop,
  # the others by the single-workgroup kernels -- same structure, other last bits), after the fix (no k_panel_coop when the
  # batch is split over streams)
  T="tests/test_two_stage_gpu.py tests/test_eigh_gpu.py tests/test_batched_configs_gpu.py tests/test_gemm_gpu.py"
  { echo "== (re-run after the fix) SPRINGCRAFT_BULGE_PERSISTENT=0 SPRINGCRAFT_BULGE_STREAMS=3 SPRINGCRAFT_STAGE1_STREAMS=3";
    SPRINGCRAFT_BULGE_PERSISTENT=0 SPRINGCRAFT_BULGE_STREAMS=3 SPRINGCRAFT_STAGE1_STREAMS=3 timeout -k 10 600 python -m pytest $T -x -q -rf 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|error" | tail -4; } > $OUT/test_matrix_4.txt
  cat $OUT/test_matrix_4.txt
elif [ $part = e ]; then
  ENVS="SPRINGCRAFT_QR_COOP=0 SPRINGCRAFT_QR_COOP=1" bash tools/r05_coop.sh > $OUT/panel_coop_ab.txt 2>&1 || { tail -5 $OUT/panel_coop_ab.txt; exit 1; }
  grep -v amdgpu.ids $OUT/panel_coop_ab.txt
  bash tools/r05_ab_env.sh SPRINGCRAFT_BT2_ROLE=0 SPRINGCRAFT_BT2_ROLE=1 2 > $OUT/bt2_role_ab.txt 2>&1
  cat $OUT/bt2_role_ab.txt
  bash tools/r05_role_stamps.sh > /dev/null 2>&1; cp gpurun_out/r05_role/stamps.txt $OUT/bt2_role_stamps.txt; cat $OUT/bt2_role_stamps.txt
  rm -f gpurun_out/r05_role/clock.txt; bash tools/r05_bt2_clock.sh > /dev/null 2>&1; cp gpurun_out/r05_role/clock.txt $OUT/bt2_clock.txt; cat $OUT/bt2_clock.txt
else
  timeout -k 10 300 python tools/gemm3_shapes.py 32 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $OUT/gemm3_shapes.txt
  cat $OUT/gemm3_shapes.txt
  timeout -k 10 500 python tools/bulge_sweep.py ext > $OUT/bulge_sweep_ext.txt 2>&1
  cat $OUT/bulge_sweep_ext.txt
  timeout -k 10 400 python tools/crossover.py 2>&1 | grep -v amdgpu.ids > $OUT/two_stage_crossover.txt
  cat $OUT/two_stage_crossover.txt
  bash tools/r05_chase_stamps.sh > /dev/null 2>&1 || true
  cat $OUT/chase_stamps.txt
fi
echo "part $part done"
