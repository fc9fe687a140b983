"""Phase times of config C4's per-GPU share (32 structures of N = 1000, InvariantForceField 13 A) through the batched solver.
python tools/c4_phases.py [n_atoms] [batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
box = 5.0 * n_atoms ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(n_atoms, B, sc.InvariantForceField(13.0))
solver.solve(coord); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    solver.solve(coord)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
solver.set_profiling(True)
solver.solve(coord); torch.cuda.synchronize()
t = solver.last_timings()
print(f"N={n_atoms} B={B}: {dt * 1e3:.1f} ms per step = {B / dt:.1f} solves/s;", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in t.items()})
