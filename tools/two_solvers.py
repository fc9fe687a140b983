"""Do two half batches on two streams (two contexts) beat one batch?  python tools/two_solvers.py [total] [n_atoms] [parts]
The phases of a solve load different units (band reduction / back-transformations: matrix cores; bulge chase, divide &
conquer: latency and L2) -- two solves that drift apart in phase could fill each other's holes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
box = 5.0 * N ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(N, 3) * box for s in range(B)])).cuda()


def run(parts, offset_ms=0.0, steps=3):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    solvers = []
    for q, s in enumerate(streams):
        with torch.cuda.stream(s):
            solvers.append(DeviceBatchSolver(N, B // parts, sc.HinsenForceField()))
    chunks = [coord[q * (B // parts):(q + 1) * (B // parts)].contiguous() for q in range(parts)]
    torch.cuda.synchronize()

    def step():
        for q, s in enumerate(streams):
            with torch.cuda.stream(s):
                solvers[q].solve(chunks[q])
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if offset_ms > 0:      # the later parts start late once: afterwards the parts stay out of phase
        for q, s in enumerate(streams[1:]):
            with torch.cuda.stream(s):
                torch.cuda._sleep(int(offset_ms * 1e-3 * (q + 1) * 2.1e9))
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0 - 1e-3 * offset_ms * (parts - 1)) / steps
    print(f"{parts} part(s) x {B // parts}, offset {offset_ms} ms: {1e3 * dt:8.1f} ms per {B} structures = {B / dt:6.2f} solves/s "
          f"(the offset subtracted once)", flush=True)
    del solvers


run(1, steps=5)
run(P, steps=5)
run(P, offset_ms=600.0, steps=5)
run(P, offset_ms=1200.0, steps=5)
run(1, steps=5)
