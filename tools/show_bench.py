"""Pretty-print the essentials of bench.py JSON lines (stdin or files)."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        ph = d.get("phases_ms_profiled_step") or {}
        r = d["roofline"]
        head = (f"{path}: {d['value']:.0f} modes/s  {d['ms_per_step']:.1f} ms/step  B={d['config']['structures_per_gpu_per_step']} "
                f"{r['kernel']} {r['achieved']} {r['unit']} ({r['frac']})  tri {ph.get('tridiag_ms', 0):.0f} "
                f"dc {ph.get('tridiag_eigen_ms', 0):.0f} bt {ph.get('backtransform_ms', 0):.0f}")
        if ph.get("two_stage"):
            print(head + f"  stage1 {ph['band_reduction_ms']:.0f} stage2 {ph['bulge_chasing_ms']:.0f} bt2 {ph['bt2_apply_ms']:.0f} ms "
                  f"(executed {r.get('executed_tflops')} TF)")
        else:
            print(head + f"  symv {ph.get('symv_ms', 0):.0f} syr2k {ph.get('syr2k_ms', 0):.0f} ms "
                  f"({ph.get('syr2k_tflops')} TF, {ph.get('syr2k_frac_of_f64_mfma_peak')})")
