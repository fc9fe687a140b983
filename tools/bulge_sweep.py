"""Bulge-chasing stage: persistent chase vs one launch per wavefront over (N, batch).  python tools/bulge_sweep.py"""
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
for n_atoms in (342, 500, 1000, 2000):
    for B in (4, 8, 16, 32, 64):
        if n_atoms * n_atoms * 9 * B * 8 * 4 > 200e9:
            continue
        row = []
        for p in ("2", "0"):   # 2 = persistent chase forced, 0 = off (1, the default, chooses by batch * n)
            env = dict(os.environ, SPRINGCRAFT_BULGE_PERSISTENT=p)
            r = subprocess.run([sys.executable, "-c", code, str(n_atoms), str(B)], capture_output=True, text=True, env=env, timeout=300)
            row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "fail")
        print(f"N={n_atoms:5d} n={3 * n_atoms:5d} B={B:3d}: persistent {row[0]:>8s} ms   per-wavefront launches {row[1]:>8s} ms", flush=True)
