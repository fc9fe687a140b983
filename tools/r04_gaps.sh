#!/bin/bash
# rocprofv3 --kernel-trace of the default bench command, reduced on the box to the gap statistics (the trace itself is
# hundreds of MB):  bash tools/r04_gaps.sh  ->  gpurun_out/r04_trace_gaps.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04_gaps
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/r04_gaps -o bench --output-format csv -- \
  python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/r04_gaps/bench_under_trace.json 2> $ROOT/gpurun_out/r04_gaps/bench_under_trace.err || exit 1
F=$(ls /tmp/r04_gaps/*kernel_trace.csv /tmp/r04_gaps/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $ROOT/tools/trace_gaps.py "$F" > $ROOT/gpurun_out/r04_trace_gaps.txt 2>&1
python3 $ROOT/tools/show_bench.py $ROOT/gpurun_out/r04_gaps/bench_under_trace.json >> $ROOT/gpurun_out/r04_trace_gaps.txt 2>&1
cat $ROOT/gpurun_out/r04_trace_gaps.txt
