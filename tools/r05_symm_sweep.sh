for sp in ${SPLITS:-7 8 9 10 12}; do SPRINGCRAFT_SYMM_SPLIT=$sp timeout -k 10 200 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ph=d['phases_ms_profiled_step']; print('split $sp', round(d['ms_per_step'],1), 'symm', round(ph['symm_ms'],1), 'band', round(ph['band_reduction_ms'],1))"; done
