"""Wall-clock of the five BASELINE.json configurations on one MI355X (host API unless noted)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device buffers for the batched configurations)

import springcraft_amd as sc  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402


def coords(n, seed):
    return np.random.RandomState(seed).rand(n, 3) * 5.0 * n ** (1 / 3)


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


# C1: 1l2y GNM 7 A (plumbing)
ca = np.load(os.path.join(ROOT, "tests", "golden", "generated", "structures.npz"))["1l2y_coord"]
t = best(lambda: sc.GNM(ca, sc.InvariantForceField(7.0)).eigen())
print(f"C1  1l2y GNM 7 A, eigen(): {t * 1e3:.2f} ms", flush=True)

# C2: N = 512 ANM 13 A, single structure, host API (coordinates in, eigenpairs out over PCIe)
c = coords(512, 0)
t = best(lambda: sc.ANM(c, sc.InvariantForceField(13.0)).eigen())
print(f"C2  N=512 ANM 13 A, ANM.eigen() host API: {t * 1e3:.1f} ms  ({1536 / t:.0f} modes/s)", flush=True)
s = DeviceBatchSolver(512, 1, sc.InvariantForceField(13.0))
x = torch.from_numpy(c[None]).cuda()
t = best(lambda: (s.solve(x), torch.cuda.synchronize()))
print(f"C2  N=512, device-resident, 1 structure: {t * 1e3:.1f} ms; ", end="")
s = DeviceBatchSolver(512, 64, sc.InvariantForceField(13.0))
x = torch.from_numpy(np.stack([coords(512, i) for i in range(64)])).cuda()
t = best(lambda: (s.solve(x), torch.cuda.synchronize()))
print(f"64 structures: {t * 1e3:.1f} ms ({64 * 1536 / t:.0f} modes/s)", flush=True)
del s

# C3: N = 2000 Hinsen: single structure via the host API; the batched number is bench.py's
c = coords(2000, 0)
t = best(lambda: sc.ANM(c, sc.HinsenForceField()).eigen(), reps=2)
print(f"C3  N=2000 Hinsen, ANM.eigen() host API (288 MB of eigenvectors over PCIe): {t * 1e3:.0f} ms", flush=True)

# C4: 256 x N = 1000 ANM 13 A on ONE GPU (the 8-GPU run shards 32 per GPU): 4 steps of 64 and 8 steps of 32
for b in (32, 64):
    s = DeviceBatchSolver(1000, b, sc.InvariantForceField(13.0))
    x = torch.from_numpy(np.stack([coords(1000, i) for i in range(b)])).cuda()
    t = best(lambda: (s.solve(x), torch.cuda.synchronize()), reps=2)
    print(f"C4  N=1000 ANM 13 A, {b} structures per step: {t * 1e3:.0f} ms/step -> 256 structures in {256 / b * t:.2f} s "
          f"({b * 3000 / t:.0f} modes/s); one GPU's share of the 8-GPU run (32 structures): {32 / b * t:.2f} s", flush=True)
    del s

# C5: N = 8000, modes 0..105
c = coords(8000, 0)
t = best(lambda: sc.ANM(c, sc.InvariantForceField(13.0)).eigen(subset_by_index=(0, 105)), reps=1)
print(f"C5  N=8000 ANM 13 A, modes 0..105: {t:.2f} s", flush=True)
