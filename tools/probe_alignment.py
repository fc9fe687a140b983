"""Does the SYMV rate depend on where the matrix buffer sits?  Sweeps the base offset of the batch of matrices."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import springcraft_amd as sc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

n_atoms, B = 2000, 4
n = 3 * n_atoms
box = 5.0 * n_atoms ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(n_atoms, B, sc.HinsenForceField())
big = torch.empty(B * n * n + (1 << 22), dtype=torch.float64, device="cuda")
print("base address %x" % big.data_ptr())
solver.set_profiling(True)
for off_bytes in (0, 256, 1024, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 0):
    view = big[off_bytes // 8: off_bytes // 8 + B * n * n].view(B, n, n)
    solver.matrix = view
    solver.solve(coord)
    torch.cuda.synchronize()
    t = solver.last_timings()
    m = np.arange(n - 1, 1, -1, dtype=np.float64)
    alg = float(np.sum(m * (m + 1) / 2) * 8.0) * B
    print(f"offset {off_bytes:8d} B  addr%2MiB={(view.data_ptr() % (2 << 20)):8d}  symv {t['symv_ms']:7.1f} ms  {alg / t['symv_ms'] / 1e6:7.0f} GB/s",
          flush=True)
