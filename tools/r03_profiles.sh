#!/bin/bash
# Round-3 evidence on the GPU box (one gpurun call):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (kernel-stats CSV -> gpurun_out/r03_prof/)
#   2. PMC passes (counters only, separate runs) of k_bt2_apply at the benchmarked batch of 64
#   3. PMC FETCH_SIZE / WRITE_SIZE of the bulge chase as ONE dispatch (persistent k_bulge_chase, 8 x N = 2000): the
#      per-wavefront form issues ~12 000 dispatches per matrix batch, under which the counter collection of round 2 died
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r03_prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r03_prof -o bench --output-format csv -- \
  python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/r03_prof/bench_under_rocprof.json 2> $ROOT/gpurun_out/r03_prof/bench_under_rocprof.err || exit 1
rm -f $ROOT/gpurun_out/r03_prof/*kernel_trace.csv $ROOT/gpurun_out/r03_prof/*/*kernel_trace.csv   # (hundreds of MB; the stats CSV is what is kept)
echo "kernel stats done"
[ "${1:-all}" = "stats" ] && exit 0
cd $ROOT
bash tools/pmc_kernel.sh k_bt2_apply r03_bt2 || exit 1
bash tools/pmc_kernel.sh k_bulge_chase r03_bulge --structures-per-gpu 8 || exit 1
