#!/bin/bash
# MFMA-pipe occupancy of one kernel of the default bench step (counters only):  bash tools/pmc_mfma.sh <kernel regex> <tag>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
KERNEL=$1; TAG=$2; shift 2
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "$KERNEL" -d $OUT/run_mfma -o r --output-format csv -- \
  python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $OUT/run_mfma.log 2>&1
rm -f $OUT/run_mfma/*kernel_trace.csv
python3 $ROOT/tools/pmc_summary.py $OUT
