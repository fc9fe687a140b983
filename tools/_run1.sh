cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_m
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_d.txt 2>&1 && \
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_m -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-runs 1 > $R/gpurun_out/r02_bench_m.json 2> $R/gpurun_out/r02_bench_m.err)
tail -3 gpurun_out/r02_gputest_d.txt; cat gpurun_out/r02_bench_m.json
