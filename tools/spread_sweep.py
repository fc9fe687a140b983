"""Bulge chasing of a few matrices, one XCD per matrix against all XCDs for every matrix (run once per setting of
SPRINGCRAFT_BULGE_SPREAD):  python tools/spread_sweep.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402
import springcraft_amd as sc  # noqa: E402

for n_atoms, B in ((700, 1), (1000, 1), (1400, 1), (2000, 1), (2000, 2), (2000, 4), (1000, 4), (3000, 1), (3000, 3)):
    box = 5.0 * n_atoms ** (1.0 / 3.0)
    coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda().contiguous()
    solver = DeviceBatchSolver(n_atoms, B, sc.InvariantForceField(13.0))
    solver.ctx.set_two_stage(True)
    solver.solve(coord)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        solver.solve(coord)
    e1.record()
    torch.cuda.synchronize()
    solver.set_profiling(True)
    solver.solve(coord)
    torch.cuda.synchronize()
    t = solver.last_timings()
    print(f"[{os.environ.get('SPRINGCRAFT_BULGE_SPREAD', 'default')}] {B} x N = {n_atoms} (n = {3 * n_atoms}): {e0.elapsed_time(e1) / 3:.1f} ms per solve, "
          f"bulge chasing {t.get('bulge_chasing_ms', 0):.1f} ms")
    del solver
