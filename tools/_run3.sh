set -e
SC_EXTRA_HIPCC_FLAGS=-DGEMM_STAMPS python springcraft_amd/csrc/build.py --force > /dev/null
timeout -k 10 300 python tools/gemm_trace.py > gpurun_out/r02_gemm_trace.txt 2>&1
cat gpurun_out/r02_gemm_trace.txt
