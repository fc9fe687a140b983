#!/bin/bash
# SQ counters of the GEMM kernels on a few shapes (two rocprofv3 runs per shape; counters only, no other trace domain).
#   bash tools/pmc_gemm.sh        (from the repo root, on the GPU box; results under gpurun_out/pmc_gemm/)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r m n k mode tile beta; do
  i=$((i+1))
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE \
    --kernel-trace --kernel-include-regex "k_gemm" -d $OUT/run${i}a -o r --output-format csv -- \
    python3 $ROOT/tools/gemm_pmc.py $m $n $k $mode $tile $beta > $OUT/run${i}a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES \
    --kernel-trace --kernel-include-regex "k_gemm" -d $OUT/run${i}b -o r --output-format csv -- \
    python3 $ROOT/tools/gemm_pmc.py $m $n $k $mode $tile $beta > $OUT/run${i}b.log 2>&1
  echo "run$i: $m $n $k mode $mode tile $tile beta $beta: $(tail -1 $OUT/run${i}a.log) / $(tail -1 $OUT/run${i}b.log)"
done <<'LIST'
6000 6000 6000 0 11 0
6000 6000 6000 0 12 0
48000 6000 256 0 12 1
24000 24000 128 1 12 1
LIST
python3 $ROOT/tools/pmc_summary.py $OUT
