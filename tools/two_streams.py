"""Experiment: one solver with batch B vs G solvers with batch B/G on G streams driven by G host threads."""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
box = 5.0 * n_atoms ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda()
ff = sc.HinsenForceField()


def run(groups, reps=2):
    per = B // groups
    solvers = [DeviceBatchSolver(n_atoms, per, ff) for _ in range(groups)]
    parts = [coord[g * per:(g + 1) * per].contiguous() for g in range(groups)]

    def work(g):
        solvers[g].solve(parts[g])
        solvers[g].ctx.synchronize()

    def step():
        if groups == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(g,)) for g in range(groups)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        torch.cuda.synchronize()

    step()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    dt = (time.perf_counter() - t0) / reps
    print(f"N={n_atoms} B={B} groups={groups}: {dt * 1e3:8.1f} ms/step  {3 * n_atoms * B / dt:9.0f} modes/s", flush=True)
    del solvers


for g in (1, 2, 4):
    run(g)
