#!/bin/bash
# round 6: the int8-emulation probe (tools/probe_i8.hip) + the NumPy error emulation of the slice scheme
set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
O=gpurun_out/r06/probe_i8_emulation.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_i8.hip -o /tmp/probe_i8 > gpurun_out/r06/probe_i8_build.txt 2>&1 || { cat gpurun_out/r06/probe_i8_build.txt; exit 1; }
{ timeout -k 10 300 /tmp/probe_i8; echo "probe rc $?"; echo; echo "error of the slice scheme (tools/models/ozaki_error.py, NumPy):"; python tools/models/ozaki_error.py; } > $O 2>&1
cat $O
