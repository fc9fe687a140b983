#!/bin/bash
# gpurun_out/r05_final/* (written by tools/r05_final.sh on the GPU box)  ->  profiles/r05_*
set -eu
cd $(dirname $0)/..
F=gpurun_out/r05_final
P=profiles
line() { grep '^{' $1 | tail -1; }
line $F/bench.json > $P/r05_bench.json
for c in c2 c4 c5; do line $F/bench_$c.json > $P/r05_bench_$c.json; done
cp $F/latency.txt $P/r05_latency.txt
cp $F/gputest_durations.txt $P/r05_gputest_durations.txt
cp $F/rocprofv3_kernel_stats_bench.csv $P/r05_rocprofv3_kernel_stats_bench.csv
[ -f $F/rocprofv3_kernel_stats_bench_c5.csv ] && cp $F/rocprofv3_kernel_stats_bench_c5.csv $P/r05_rocprofv3_kernel_stats_bench_c5.csv
cp $F/bt2_pmc_fetch_write.json $P/r05_bt2_pmc_fetch_write.json
cp $F/bt2_pmc_summary.txt $P/r05_bt2_pmc_summary.txt
grep -h "^run_mfma" $F/bt2_pmc_mfma.txt > $P/r05_bt2_pmc_mfma.txt
grep -h "^run_mfma" $F/gemm3_pmc_mfma.txt > $P/r05_gemm3_pmc_mfma.txt
{ echo "# bash tools/r05_final.sh c1 | c2 | c3 (tools/test_matrix.sh in three parts) on the final build; 27 rows."
  echo "# One row FAILED in the first run of the matrix this round (build of commit 'Pair chase by size: ...'):"
  echo "#   == SPRINGCRAFT_BULGE_PERSISTENT=0 SPRINGCRAFT_BULGE_STREAMS=3 SPRINGCRAFT_STAGE1_STREAMS=3"
  echo "#   FAILED tests/test_batched_configs_gpu.py::test_config3_batched_automatic_path"
  echo "#   1 failed, 138 passed, 2 skipped in 50.49s"
  echo "# ('same structure at two batch positions must give identical eigenvalues'): with the batch split over three streams the"
  echo "# part on the main stream was factored by k_panel_coop, the parts on the side streams by the single-workgroup panel"
  echo "# kernels -- both right, other reduction trees, other last bits.  Fix (twostage.hip): no k_panel_coop when the batch is"
  echo "# split over streams; test_cooperative_panel_qr follows that override.  The rows below are the re-run of ALL rows after it."
  cat $F/test_matrix_1.txt $F/test_matrix_2.txt $F/test_matrix_3.txt | grep -v "^part"; } > $P/r05_test_matrix.txt
cp $F/gemm3_shapes.txt $P/r05_gemm3_shapes.txt
cp $F/bulge_sweep_ext.txt $P/r05_bulge_sweep_ext.txt
cp $F/two_stage_crossover.txt $P/r05_two_stage_crossover.txt
cp $F/chase_stamps.txt $P/r05_chase_stamps.txt
grep -v "amdgpu.ids" $F/panel_coop_ab.txt > $P/r05_panel_coop_ab.txt
cp $F/bt2_role_ab.txt $P/r05_bt2_role_ab.txt
cp $F/bt2_role_stamps.txt $P/r05_bt2_role_stamps.txt
cp $F/bt2_clock.txt $P/r05_bt2_clock.txt
ls -la $P/r05_* | awk '{print $5, $9}'
