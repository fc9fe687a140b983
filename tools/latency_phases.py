"""Single-structure latency (the reference's own calling pattern, anm.py:150-167 -> nma.py:61): wall-clock and phase
breakdown of ONE structure at a time, one-stage and two-stage path, N = 512 and N = 2000."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402


def coords(n, seed):
    return np.random.RandomState(seed).rand(n, 3) * 5.0 * n ** (1 / 3)


for n_atoms, ff in ((512, sc.InvariantForceField(13.0)), (2000, sc.HinsenForceField())):
    x = torch.from_numpy(coords(n_atoms, 0)[None]).cuda()
    for mode in (False, True):
        s = DeviceBatchSolver(n_atoms, 1, ff)
        s.ctx.set_two_stage(mode)
        s.solve(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            s.solve(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        s.set_profiling(True)
        s.solve(x)
        torch.cuda.synchronize()
        t = s.last_timings()
        s.set_profiling(False)
        print(f"N={n_atoms} two_stage={mode}: {min(ts) * 1e3:.1f} ms unprofiled; profiled phases: "
              + ", ".join(f"{k}={v:.1f}" if isinstance(v, float) else f"{k}={v}" for k, v in t.items()), flush=True)
        del s
    c = coords(n_atoms, 0)
    sc.ANM(c, ff).eigen()
    t0 = time.perf_counter()
    sc.ANM(c, ff).eigen()
    print(f"N={n_atoms} host API ANM.eigen(): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
