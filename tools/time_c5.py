"""Config C5 (N = 8000 C-alpha, 13 A cutoff, modes 0..105) through both tridiagonalisation paths."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import springcraft_amd as sc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402

n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
coord = np.random.RandomState(0).rand(n_atoms, 3) * 5.0 * n_atoms ** (1 / 3)
ctx = _hip.context()
res = {}
for mode in (False, True, False, True):
    ctx.set_two_stage(mode)
    t0 = time.perf_counter()
    w, v = sc.ANM(coord, sc.InvariantForceField(13.0)).eigen(subset_by_index=(0, 105))
    dt = time.perf_counter() - t0
    res[mode] = w
    print(f"N={n_atoms} two_stage={mode}: {dt:.3f} s   w[6:9]={w[6:9]}", flush=True)
print("max rel diff between the paths:", np.abs(res[True][6:] - res[False][6:]).max() / np.abs(res[False]).max())
