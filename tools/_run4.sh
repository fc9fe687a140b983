set -e
SC_EXTRA_HIPCC_FLAGS=-DBULGE_STAMPS python springcraft_amd/csrc/build.py --force > /dev/null
for b in 64 32 8; do timeout -k 10 200 python tools/bulge_stamps.py $b; done > gpurun_out/r02_bulge_stamps.txt 2>&1
cat gpurun_out/r02_bulge_stamps.txt
