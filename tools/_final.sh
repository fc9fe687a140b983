R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputest_f.txt 2>&1 || { tail -20 gpurun_out/r02_gputest_f.txt; exit 1; }
tail -2 gpurun_out/r02_gputest_f.txt
timeout -k 10 400 python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err || exit 1
python tools/show_bench.py gpurun_out/r02_bench_final.json
timeout -k 10 400 python bench.py --config c4 --no-cpu-baseline > gpurun_out/r02_bench_c4.json 2> gpurun_out/r02_bench_c4.err || exit 1
rm -rf gpurun_out/prof_final; mkdir -p gpurun_out/prof_final
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02_bench_prof.json 2> $R/gpurun_out/r02_bench_prof.err
cd $R
python tools/show_bench.py gpurun_out/r02_bench_prof.json
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()"
