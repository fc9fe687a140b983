"""Bulge-chasing stage over (N, batch): pair form of the persistent chase (k_bulge_pair), one sweep per workgroup
(k_bulge_chase), one launch per wavefront (k_bulge_step).   python tools/bulge_sweep.py [quick]"""
import os
import subprocess
import sys

code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
import springcraft_amd as sc
from springcraft_amd.batch import DeviceBatchSolver
n_atoms, B = int(sys.argv[1]), int(sys.argv[2])
box = 5.0 * n_atoms ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(n_atoms, B, sc.InvariantForceField(13.0))
solver.ctx.set_two_stage(True)
solver.solve(coord); torch.cuda.synchronize()
solver.set_profiling(True)
solver.solve(coord); torch.cuda.synchronize()
t = solver.last_timings()
print(f"{t['bulge_chasing_ms']:.1f}")
'''
MODES = (("pair", {"SPRINGCRAFT_BULGE_PERSISTENT": "2", "SPRINGCRAFT_BULGE_PAIR": "2", "SPRINGCRAFT_BULGE_PAIR_MAX": "1000000"}),
         ("sweep/wg", {"SPRINGCRAFT_BULGE_PERSISTENT": "2", "SPRINGCRAFT_BULGE_PAIR": "0"}),
         ("per-wavefront", {"SPRINGCRAFT_BULGE_PERSISTENT": "0"}))
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
ext = len(sys.argv) > 1 and sys.argv[1] == "ext"   # beyond the round-4 range: more matrices per XCD than pair workgroups fit
points = [(500, 64), (1000, 16), (1000, 32), (1000, 64), (2000, 8), (2000, 16), (2000, 32), (2000, 64)] if quick else [
    (n_atoms, B) for n_atoms in (342, 500, 1000, 2000) for B in (4, 8, 16, 32, 64)]
if ext:
    points = [(2000, 64), (2000, 96), (1000, 128), (1000, 256), (500, 256), (342, 512)]
for n_atoms, B in points:
    if n_atoms * n_atoms * 9 * B * 8 * 4 > 200e9:
        continue
    row = []
    for name, extra in MODES:
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code, str(n_atoms), str(B)], capture_output=True, text=True, env=env, timeout=300)
        row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "fail")
    print(f"N={n_atoms:5d} n={3 * n_atoms:5d} B={B:3d} (B n / 128 = {B * 3 * n_atoms // 128:5d}): " +
          "   ".join(f"{name} {v:>7s} ms" for (name, _), v in zip(MODES, row)), flush=True)
