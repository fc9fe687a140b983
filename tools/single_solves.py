"""ONE structure at a time, `reps` solves of N atoms on the device (for rocprofv3 --kernel-trace --stats: the kernels of
the reference's own calling pattern, anm.py:150-167 -> nma.py:61).   python tools/single_solves.py [N = 512] [reps = 20]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = torch.from_numpy((np.random.RandomState(0).rand(n_atoms, 3) * 5.0 * n_atoms ** (1 / 3))[None]).cuda()
s = DeviceBatchSolver(n_atoms, 1, sc.InvariantForceField(13.0))
s.solve(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    s.solve(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
print(f"N={n_atoms}: {dt:.2f} ms per solve ({reps} solves enqueued back to back); resident_launches "
      f"{s.ctx.counter('resident_launches')}, resident_takeovers {s.ctx.counter('resident_takeovers')}", flush=True)
