#!/bin/bash
# PMC passes (counters only) of one kernel of the default bench step:  bash tools/pmc_kernel.sh <kernel regex> <tag> [bench args]
# pass 1: SQ counters; pass 2: FETCH_SIZE; pass 3: WRITE_SIZE + TCC hit / miss.  Results under gpurun_out/pmc_<tag>/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
KERNEL=$1; TAG=$2; shift 2
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "$KERNEL" -d $OUT/run_$name -o r --output-format csv -- \
    python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 $BENCH_ARGS > $OUT/run_$name.log 2>&1
  echo "pass $name: $(tail -c 300 $OUT/run_$name.log | tr '\n' ' ' | cut -c1-200)"
}
BENCH_ARGS="$*"
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 $ROOT/tools/pmc_summary.py $OUT
