#!/bin/bash
# K slices of the SYMM in the middle regime (6 x n = 6000: 564 tiles, automatic: 4 slices): even against odd
for sp in ${SPLITS:-4 5}; do SPRINGCRAFT_SYMM_SPLIT=$sp timeout -k 10 100 python bench.py --structures-per-gpu 6 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ph=d['phases_ms_profiled_step']; print('6 x 6000, split $sp:', round(d['ms_per_step'],1), 'ms/step  symm', round(ph['symm_ms'],1), 'band', round(ph['band_reduction_ms'],1))"; done
