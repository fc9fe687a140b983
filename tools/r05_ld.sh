#!/bin/bash
mkdir -p gpurun_out/r05_g3
for e in "" "SC_DBG_LDA=6144" "SC_DBG_LDC=6144" "SC_DBG_LDA=6144 SC_DBG_LDC=6144" "SC_DBG_LDA=6144 SC_DBG_LDC=6144 SC_DBG_LDB=6144" "SC_DBG_LDA=6016 SC_DBG_LDC=6016 SC_DBG_LDB=6016" "SC_DBG_LDA=6272 SC_DBG_LDC=6272 SC_DBG_LDB=6272"; do
  echo "== $e"
  env $e timeout -k 10 300 python tools/gemm3_shapes.py 32 ld 2>&1 | grep -v "amdgpu.ids\|^gfx950" | cut -c1-200
done 2>&1 | tee gpurun_out/r05_g3/shapes_ld.txt
