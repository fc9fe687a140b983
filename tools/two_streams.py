"""Experiment: one solver with batch B vs G solvers with batch B/G, each on its OWN stream and driven by its own host
thread, started with a phase offset so that the MFMA-bound phases of one group meet the bandwidth- / latency-bound phases
of another.    python tools/two_streams.py [n_atoms] [B]"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
box = 5.0 * n_atoms ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda()
ff = sc.HinsenForceField()


def run(groups, reps=3, offset_frac=0.5):
    per = B // groups
    streams = [torch.cuda.Stream() for _ in range(groups)]
    solvers = []
    for g in range(groups):
        with torch.cuda.stream(streams[g]):
            solvers.append(DeviceBatchSolver(n_atoms, per, ff))
    parts = [coord[g * per:(g + 1) * per].contiguous() for g in range(groups)]
    torch.cuda.synchronize()
    # warm-up (workspace allocation) and the duration of one group's step alone
    t0 = time.perf_counter()
    solvers[0].solve(parts[0]); solvers[0].ctx.synchronize()
    for g in range(1, groups):
        solvers[g].solve(parts[g]); solvers[g].ctx.synchronize()
    t0 = time.perf_counter()
    solvers[0].solve(parts[0]); solvers[0].ctx.synchronize()
    t_one = time.perf_counter() - t0

    def work(g):
        time.sleep(g * offset_frac * t_one / max(groups - 1, 1) if groups > 1 else 0.0)
        for _ in range(reps):
            solvers[g].solve(parts[g])
            solvers[g].ctx.synchronize()

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(g,)) for g in range(groups)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"N={n_atoms} B={B} groups={groups} (one group alone {t_one * 1e3:.0f} ms): {dt * 1e3:8.1f} ms per {B} structures  "
          f"{3 * n_atoms * B / dt:9.0f} modes/s", flush=True)
    del solvers


for g in (1, 2, 4):
    run(g)
