#!/bin/bash
# rocprofv3 kernel statistics of one bench configuration:  bash tools/stats_config.sh c4   ->  gpurun_out/stats_<cfg>.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/stats_$CFG -o s --output-format csv -- \
  python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/stats_$CFG.json 2> $ROOT/gpurun_out/stats_$CFG.err || exit 1
cp $(ls /tmp/stats_$CFG/*kernel_stats.csv /tmp/stats_$CFG/*/*kernel_stats.csv 2>/dev/null | head -1) $ROOT/gpurun_out/stats_$CFG.csv
