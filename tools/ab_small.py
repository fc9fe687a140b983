"""Latency-regime timing: single-structure ANM eigensolves at small/medium N through the batched device API."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

cases = ((512, 1), (1000, 1), (1000, 8), (2000, 1), (2000, 16))
if len(sys.argv) > 1:   # N:B pairs
    cases = tuple(tuple(int(x) for x in a.split(":")) for a in sys.argv[1:])
for n_atoms, B in cases:
    box = 5.0 * n_atoms ** (1 / 3)
    coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda()
    solver = DeviceBatchSolver(n_atoms, B, sc.InvariantForceField(13.0) if n_atoms < 2000 else sc.HinsenForceField())
    solver.solve(coord)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        solver.solve(coord)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"N={n_atoms:5d} B={B:2d}: {dt * 1e3:8.1f} ms/step  {3 * n_atoms * B / dt:9.0f} modes/s", flush=True)
    del solver
