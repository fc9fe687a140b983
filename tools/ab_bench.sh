#!/bin/bash
# A/B/... of several builds of libspringcraft_hip.so on the SAME GPU box (devices differ by several %):
#   tools/ab_bench.sh <rounds> <lib_a.so> <lib_b.so> ...
R=$1; shift
for r in $(seq 1 $R); do
  i=0
  for L in "$@"; do
    SPRINGCRAFT_HIP_LIB=$L python bench.py --no-cpu-baseline --steps 2 > gpurun_out/ab_${i}_$r.log 2>/dev/null
    i=$((i+1))
  done
done
i=0
for L in "$@"; do echo "== $L"; python tools/show_bench.py gpurun_out/ab_${i}_*.log | sed 's/^[^:]*: //'; i=$((i+1)); done
