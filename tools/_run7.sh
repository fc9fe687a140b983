run() {
  for sh in "24000 24000 128 1 12 1" "48000 6000 256 0 12 1" "48000 6000 128 0 12 1" "6000 6000 6000 0 12 0" "1030 517 333 0 12 1" "999 999 77 1 12 1"; do
    echo "$sh: $(timeout -k 10 120 python tools/gemm_pmc.py $sh 2>&1 | tail -1)"
  done
}
echo "== PF=1"; run
export SPRINGCRAFT_GEMM_PF=2
echo "== PF=2"; run
