#!/bin/bash
# Round-6 evidence on one GPU box, in parts (a gpurun call is limited to 20 minutes):
#   bash tools/r06_final.sh a   the -m gpu suite with durations, the four bench lines (C3 with the CPU baseline), latencies
#   bash tools/r06_final.sh ab  the default bench line and the rocprofv3 kernel statistics of the same command on ONE box
#   bash tools/r06_final.sh b   rocprofv3 kernel statistics of the default bench command; PMC passes (counters only, separate
#                               runs) of k_bt2_apply (bytes, MFMA-pipe occupancy at the benchmarked batch), of k_symm3 and
#                               of k_bulge_pair (bytes)
#   bash tools/r06_final.sh c1 | c2 | c3 | c4   the test matrix (tools/test_matrix.sh), in four parts; c0: the rows that failed
#                               in their first run (after the fixes) and the two rows added later
#   bash tools/r06_final.sh d   k_bulge_pair: per-phase stamps without / with loader waves and with the early look, the
#                               diagnostic variants (512-thread form under the 768-thread form's register budget; loader
#                               waves that request nothing), same-box A/B of the three forms on the bench step
#   bash tools/r06_final.sh e   the spread chase on / off (C5, n = 12000, one N = 2000 structure); k_symm3: micro-bench (random and zero-like "hot" operands), loader-wave stamps, on / off in
#                               the bench step and in C5 / C4 / C2; trailing update on 224 / 256 workgroups; D&C per level
#   bash tools/r06_final.sh f   two-rank rehearsal on the shared GPU (bench.py --gpus 2, gloo)
#   bash tools/r06_final.sh g   one structure at a time (k_sytrd_resident): accuracy and times against the launches per column
#                               and numpy, both take-over routes, per-segment stamps (tools/build_res_stamps_lib.sh first),
#                               the pause before the first poll, one- vs two-stage for one matrix, back-to-back stress,
#                               rocprofv3 kernel statistics of single solves, the second stream on / off
#   bash tools/r06_final.sh c5  the rows of the test matrix added with it
# Everything lands in gpurun_out/r06_final/; what is to be judged is copied to profiles/r06_* (tools/r06_collect.sh).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_final
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
  timeout -k 10 300 python tools/latency_phases.py > $OUT/latency.txt 2>&1 || exit 1
  grep -v amdgpu.ids $OUT/latency.txt
elif [ $part = ab ]; then
  # the default bench line and the rocprofv3 kernel statistics of the same command on ONE box (boxes differ by 3 - 5 %)
  timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
  python tools/show_bench.py $OUT/bench.json
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench --output-format csv -- \
     python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err) || { tail -5 $OUT/bench_under_rocprof.err; exit 1; }
  rm -f $OUT/prof/*kernel_trace.csv $OUT/prof/*/*kernel_trace.csv
  cp $(ls -t $OUT/prof/*kernel_stats.csv $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_bench.csv
  head -4 $OUT/rocprofv3_kernel_stats_bench.csv | cut -c1-200
elif [ $part = b ]; then
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench --output-format csv -- \
     python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err) || { tail -5 $OUT/bench_under_rocprof.err; exit 1; }
  rm -f $OUT/prof/*kernel_trace.csv $OUT/prof/*/*kernel_trace.csv
  cp $(ls $OUT/prof/*kernel_stats.csv $OUT/prof/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_bench.csv
  head -8 $OUT/rocprofv3_kernel_stats_bench.csv | cut -c1-200
  bash tools/pmc_kernel.sh k_bt2_apply r06_bt2 || exit 1
  bash tools/pmc_mfma.sh k_bt2_apply r06_bt2 > $OUT/bt2_pmc_mfma.txt 2>&1 || exit 1
  python3 tools/pmc_to_json.py gpurun_out/pmc_r06_bt2 k_bt2_apply 6000 64 > $OUT/bt2_pmc_fetch_write.json
  grep -E "traffic_over|l2_hit|hbm_bytes" $OUT/bt2_pmc_fetch_write.json
  python3 tools/pmc_summary.py gpurun_out/pmc_r06_bt2 > $OUT/bt2_pmc_summary.txt 2>&1
  bash tools/pmc_kernel.sh k_symm3 r06_symm3 > /dev/null 2>&1 || exit 1
  bash tools/pmc_mfma.sh "k_symm3|k_gemm3" r06_symm3 > $OUT/symm3_pmc_mfma.txt 2>&1 || exit 1
  python3 tools/pmc_summary.py gpurun_out/pmc_r06_symm3 | cut -c1-700 > $OUT/symm3_pmc_summary.txt 2>&1
  cat $OUT/symm3_pmc_summary.txt | cut -c1-300
  bash tools/pmc_kernel.sh k_bulge_pair r06_bulge > /dev/null 2>&1 || exit 1
  python3 tools/pmc_to_json.py gpurun_out/pmc_r06_bulge k_bulge_pair 6000 64 > $OUT/bulge_pmc_fetch_write.json
  grep -E "traffic_over|l2_hit|hbm_bytes" $OUT/bulge_pmc_fetch_write.json
elif [ $part = c1 ] || [ $part = c2 ] || [ $part = c3 ] || [ $part = c4 ]; then
  k=${part#c}
  bash tools/test_matrix.sh $k 4 > $OUT/test_matrix_$k.txt 2>&1
  cat $OUT/test_matrix_$k.txt
elif [ $part = c0 ]; then
  T="tests/test_two_stage_gpu.py tests/test_eigh_gpu.py tests/test_batched_configs_gpu.py tests/test_gemm_gpu.py"
  row() { echo "== $*"; env "$@" timeout -k 10 600 python -m pytest $T -x -q -rf 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|error" | tail -4; }
  { row SPRINGCRAFT_BULGE_PERSISTENT=0 SPRINGCRAFT_BULGE_STREAMS=1 SPRINGCRAFT_STAGE1_STREAMS=1
    row SPRINGCRAFT_BULGE_PERSISTENT=0 SPRINGCRAFT_BULGE_STREAMS=3 SPRINGCRAFT_STAGE1_STREAMS=3
    row SPRINGCRAFT_SYMM3=0
    row SPRINGCRAFT_BULGE_SPREAD=1
    row SPRINGCRAFT_BULGE_SPREAD=0; } > $OUT/test_matrix_0.txt 2>&1
  cat $OUT/test_matrix_0.txt
elif [ $part = d ]; then
  bash tools/r06_pair_stamps.sh > $OUT/pair_stamps.txt 2>&1
  { echo "== early look (SPRINGCRAFT_PAIR_EARLY=1), no loader waves";
    SPRINGCRAFT_PAIR_EARLY=1 SPRINGCRAFT_HIP_LIB=$PWD/springcraft_amd/libspringcraft_hip_stamps.so timeout -k 10 300 python tools/pair_stamps.py 2000 64 2>/dev/null; } >> $OUT/pair_stamps.txt
  bash tools/r06_pair_variants.sh >> $OUT/pair_stamps.txt 2>&1
  grep -v "^rc 0$" $OUT/pair_stamps.txt
  bash tools/quick_env_ab.sh "SPRINGCRAFT_PAIR_LOADER=0" "SPRINGCRAFT_PAIR_LOADER=1" "SPRINGCRAFT_PAIR_EARLY=1" > $OUT/pair_ab.txt 2>&1
  cat $OUT/pair_ab.txt
elif [ $part = e ]; then
  { python tools/symm3_bench.py; SPRINGCRAFT_SYMM3_DBG_HOT=1 python tools/symm3_bench.py 32 5888 1 64 3008 1; } 2>/dev/null > $OUT/symm3_bench.txt
  cat $OUT/symm3_bench.txt
  bash tools/r06_symm3_stamps.sh > $OUT/symm3_stamps.txt 2>&1; SPRINGCRAFT_SYMM3_DBG_HOT=1 bash tools/r06_symm3_stamps.sh >> $OUT/symm3_stamps.txt 2>&1
  cat $OUT/symm3_stamps.txt
  bash tools/quick_env_ab.sh "SPRINGCRAFT_SYMM3=0 SPRINGCRAFT_GEMM3_LOWER=0" "SPRINGCRAFT_SYMM3=0" "SPRINGCRAFT_GEMM3_LOWER=0" "X=default" "SPRINGCRAFT_SYMM3_WGS=256 SPRINGCRAFT_GEMM3_LOWER_WGS=256" > $OUT/symm3_syr2k_ab.txt 2>&1
  cat $OUT/symm3_syr2k_ab.txt
  bash tools/quick_env_ab.sh "SPRINGCRAFT_SYMM3_WGS=208 SPRINGCRAFT_GEMM3_LOWER_WGS=208" "SPRINGCRAFT_SYMM3_WGS=216 SPRINGCRAFT_GEMM3_LOWER_WGS=216" "X=default224" "SPRINGCRAFT_SYMM3_WGS=240 SPRINGCRAFT_GEMM3_LOWER_WGS=240" "SPRINGCRAFT_STAGE1_STREAMS=3" > $OUT/syr2k_wgs.txt 2>&1
  cat $OUT/syr2k_wgs.txt
  ENVS="SPRINGCRAFT_SYMM3=0;SPRINGCRAFT_SYMM3=1" bash tools/r06_cfgs.sh > $OUT/symm3_cfgs.txt 2>&1
  grep -v "^    {" $OUT/symm3_cfgs.txt
  { ENVS="SPRINGCRAFT_BULGE_SPREAD=0;X=default" CFGS="c5" bash tools/r06_cfgs.sh 2>&1 | grep -v "^    {.bt2"
    echo "== a few matrices, one XCD per matrix (SPRINGCRAFT_BULGE_SPREAD=0) against all XCDs for every matrix (=1): tools/spread_sweep.py"
    for sp in 0 1; do SPRINGCRAFT_BULGE_SPREAD=$sp timeout -k 10 400 python tools/spread_sweep.py 2>/dev/null; done
    echo "== n = 12000 (bench.py --config c5 --n-atoms 4000), SPRINGCRAFT_BULGE_SPREAD = 0 / default"
    for e in SPRINGCRAFT_BULGE_SPREAD=0 X=default; do env $e timeout -k 10 300 python bench.py --config c5 --n-atoms 4000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  $e', round(d['ms_per_step'],1), 'ms per solve, bulge chasing', round(d['phases_ms_profiled_step'].get('bulge_chasing_ms',0),1))"; done; } > $OUT/spread_chase.txt 2>&1
  cat $OUT/spread_chase.txt
  bash tools/r06_dc_levels.sh > $OUT/dc_levels.txt 2>&1
  cat $OUT/dc_levels.txt
  bash tools/r06_probe_i8.sh > /dev/null 2>&1; cp gpurun_out/r06/probe_i8_emulation.txt $OUT/probe_i8_emulation.txt; tail -12 $OUT/probe_i8_emulation.txt
elif [ $part = g ]; then
  timeout -k 10 600 python tools/resident_check.py 128 300 900 1536 2048 2560 3072 3300 2>&1 | grep -v amdgpu.ids > $OUT/resident_check.txt || exit 1
  cat $OUT/resident_check.txt | cut -c1-220
  SPRINGCRAFT_HIP_LIB=$ROOT/springcraft_amd/libspringcraft_hip_res_stamps.so timeout -k 10 300 python tools/resident_check.py --stamps 300 1536 2048 3000 2>&1 | grep "n=" > $OUT/resident_stamps.txt
  cat $OUT/resident_stamps.txt
  { for d in 0 8 16 24 32 48 64; do echo "== SPRINGCRAFT_RESIDENT_DELAY=$d"; SPRINGCRAFT_RESIDENT_DELAY=$d timeout -k 10 200 python tools/resident_check.py 300 1536 2048 2>&1 | grep "per-column" | cut -c1-75; done; } > $OUT/resident_delay.txt 2>&1
  cat $OUT/resident_delay.txt
  timeout -k 10 900 python tools/crossover.py --single 683 1000 1500 1700 2000 2200 2400 2600 2>&1 | grep "N=" > $OUT/two_stage_crossover.txt
  cat $OUT/two_stage_crossover.txt
  { timeout -k 10 300 python tools/resident_stress.py 512 30 20; timeout -k 10 300 python tools/resident_stress.py 300 30 20; timeout -k 10 300 python tools/resident_stress.py 1000 10 10; } 2>&1 | grep "N=" > $OUT/resident_stress.txt
  cat $OUT/resident_stress.txt
  { for a in 0 1 0 1; do for n in 100 300 512 1000; do SPRINGCRAFT_AUX_SINGLE=$a timeout -k 10 100 python tools/single_solves.py $n 20 2>&1 | grep "N=" | sed "s/^/SPRINGCRAFT_AUX_SINGLE=$a /"; done; done; } > $OUT/aux_single_ab.txt 2>&1
  cat $OUT/aux_single_ab.txt
  for n in 100 512 1000; do
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_single/n$n -o s --output-format csv -- \
       python3 $ROOT/tools/single_solves.py $n 20 > $OUT/single_under_rocprof_$n.txt 2>&1) || { tail -5 $OUT/single_under_rocprof_$n.txt; exit 1; }
    rm -f $OUT/prof_single/n$n/*kernel_trace.csv $OUT/prof_single/n$n/*/*kernel_trace.csv
    cp $(ls -t $OUT/prof_single/n$n/*kernel_stats.csv $OUT/prof_single/n$n/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/rocprofv3_kernel_stats_single_n$n.csv
    grep "N=" $OUT/single_under_rocprof_$n.txt; head -3 $OUT/rocprofv3_kernel_stats_single_n$n.csv | cut -c1-160
  done
elif [ $part = c5 ]; then
  T="tests/test_two_stage_gpu.py tests/test_eigh_gpu.py tests/test_batched_configs_gpu.py tests/test_gemm_gpu.py tests/test_resident_gpu.py"
  row() { echo "== $*"; env "$@" timeout -k 10 600 python -m pytest $T -x -q -rf 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|error" | tail -4; }
  { row SPRINGCRAFT_RESIDENT=0
    row SPRINGCRAFT_AUX_SINGLE=0
    row SPRINGCRAFT_RESIDENT_WGS=256 SPRINGCRAFT_RESIDENT_DELAY=0
    row SPRINGCRAFT_RESIDENT_MAX=2048
    row SPRINGCRAFT_TWO_STAGE=1
    row SPRINGCRAFT_TWO_STAGE=0; } > $OUT/test_matrix_5.txt 2>&1
  cat $OUT/test_matrix_5.txt
else
  SPRINGCRAFT_BENCH_SHARE_GPUS=1 timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/rehearsal_2ranks.json 2> $OUT/rehearsal_2ranks.err; echo "rehearsal rc $?"
  python tools/show_bench.py $OUT/rehearsal_2ranks.json
  SPRINGCRAFT_BENCH_SHARE_GPUS=1 timeout -k 10 400 python bench.py --gpus 2 --config c4 --steps 2 --warmup 1 > $OUT/rehearsal_c4_2ranks.json 2> $OUT/rehearsal_c4_2ranks.err; echo "rehearsal c4 rc $?"
  python tools/show_bench.py $OUT/rehearsal_c4_2ranks.json
fi
echo "part $part done"
