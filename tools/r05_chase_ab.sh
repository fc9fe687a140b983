for c in c4 c2; do timeout -k 10 200 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$c', round(d['ms_per_step'],1), 'ms/step  chase', round(d['phases_ms_profiled_step']['bulge_chasing_ms'],1), (d.get('parity_gates') or {}).get('pass'))"; done
timeout -k 10 200 python tools/latency_phases.py 2>&1 | grep "two_stage=True" | cut -c1-140
