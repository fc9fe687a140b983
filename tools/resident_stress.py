"""Many single-structure solves enqueued back to back (no synchronisation in between, as a caller of the device API
would): time of every batch of `reps` solves and the context's resident_launches / resident_takeovers counters -- a batch
that takes many times the others' time or a take-over count above zero is a roll call of k_sytrd_resident that failed.
    python tools/resident_stress.py [N = 512] [batches = 30] [reps = 20]"""
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
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
x = torch.from_numpy((np.random.RandomState(0).rand(n_atoms, 3) * 5.0 * n_atoms ** (1 / 3))[None]).cuda()
s = DeviceBatchSolver(n_atoms, 1, sc.InvariantForceField(13.0))
s.solve(x)
torch.cuda.synchronize()
ts = []
for b in range(batches):
    t0 = time.perf_counter()
    for _ in range(reps):
        s.solve(x)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / reps * 1e3)
ts = np.array(ts)
print(f"N={n_atoms}: {batches} x {reps} solves: per solve min {ts.min():.2f} median {np.median(ts):.2f} max {ts.max():.2f} ms; "
      f"resident_launches {s.ctx.counter('resident_launches')}, resident_takeovers {s.ctx.counter('resident_takeovers')}", flush=True)
