"""Two half batches on two streams, driven by two host threads, against one full batch: do the latency-bound phases of one
half (panel QR, bulge chase) hide behind the GEMM phases of the other?  python tools/pipeline_probe.py [n_atoms] [batch] [parts]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ff = sc.HinsenForceField() if N == 2000 else sc.InvariantForceField(13.0)
box = 5.0 * N ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(N, 3) * box for s in range(B)])).cuda()
steps = 4

one = DeviceBatchSolver(N, B, ff)
one.solve(coord); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one.solve(coord)
torch.cuda.synchronize()
t_one = (time.perf_counter() - t0) / steps * 1e3
w_ref = one.w.clone()
del one
torch.cuda.empty_cache()

streams = [torch.cuda.Stream() for _ in range(P)]
bounds = [(B * p // P, B * (p + 1) // P) for p in range(P)]
solvers = []
for p in range(P):
    with torch.cuda.stream(streams[p]):
        solvers.append(DeviceBatchSolver(N, bounds[p][1] - bounds[p][0], ff))
parts = [coord[lo:hi].contiguous() for lo, hi in bounds]
torch.cuda.synchronize()


def drive(p, k):
    with torch.cuda.stream(streams[p]):
        for _ in range(k):
            solvers[p].solve(parts[p])
        streams[p].synchronize()


def run(k):
    th = [threading.Thread(target=drive, args=(p, k)) for p in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join()


run(1)
t0 = time.perf_counter()
run(steps)
t_two = (time.perf_counter() - t0) / steps * 1e3
w_two = torch.cat([s.w for s in solvers])
print(f"N={N} batch {B}: one solver {t_one:.1f} ms per step; {P} solvers on {P} streams / host threads {t_two:.1f} ms per step "
      f"({100 * (t_one / t_two - 1):+.1f} %); eigenvalues equal: {bool(torch.equal(w_ref, w_two))}", flush=True)
