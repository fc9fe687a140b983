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
cp $F/bt2_pmc_fetch_write.json $P/r05_bt2_pmc_fetch_write.json
cp $F/bt2_pmc_summary.txt $P/r05_bt2_pmc_summary.txt
grep -h "^run_mfma" $F/bt2_pmc_mfma.txt > $P/r05_bt2_pmc_mfma.txt
grep -h "^run_mfma" $F/gemm3_pmc_mfma.txt > $P/r05_gemm3_pmc_mfma.txt
{ cat $F/test_matrix_1.txt $F/test_matrix_2.txt $F/test_matrix_3.txt | grep -v "^part"
  echo "# The row that failed above (test_config3_batched_automatic_path: \"same structure at two batch positions must give"
  echo "# identical eigenvalues\") did so because with SPRINGCRAFT_STAGE1_STREAMS=3 the part of the batch on the main stream was"
  echo "# factored by k_panel_coop and the parts on the side streams by the single-workgroup panel kernels: both right, other"
  echo "# reduction trees, other last bits.  Fix (twostage.hip): no k_panel_coop when the batch is split over streams.  Re-run:"
  grep -v "^part" $F/test_matrix_4.txt; } > $P/r05_test_matrix.txt
cp $F/gemm3_shapes.txt $P/r05_gemm3_shapes.txt
cp $F/bulge_sweep_ext.txt $P/r05_bulge_sweep_ext.txt
cp $F/two_stage_crossover.txt $P/r05_two_stage_crossover.txt
cp $F/chase_stamps.txt $P/r05_chase_stamps.txt
grep -v "amdgpu.ids" $F/panel_coop_ab.txt > $P/r05_panel_coop_ab.txt
cp $F/bt2_role_ab.txt $P/r05_bt2_role_ab.txt
cp $F/bt2_role_stamps.txt $P/r05_bt2_role_stamps.txt
cp $F/bt2_clock.txt $P/r05_bt2_clock.txt
ls -la $P/r05_* | awk '{print $5, $9}'
