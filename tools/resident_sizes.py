"""k_sytrd_resident over many orders, the boundaries of its chunks (256 columns) and rows (P, 8 / 12 per workgroup) among
them: eigenvalues against LAPACK, eigenvectors by residual and orthogonality.   python tools/resident_sizes.py [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from springcraft_amd import nma, _hip  # noqa: E402

ctx = _hip.context()
ctx.set_two_stage(False)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rs = np.random.RandomState(seed)
edges = [128, 129, 191, 192, 193, 255, 256, 257, 383, 511, 512, 513, 767, 769, 1023, 1024, 1025, 1279, 1535, 1537, 1791, 1793, 2047,
         2048, 2049, 2303, 2305, 2559, 2560, 2561, 2815, 2817, 3071, 3072, 3073, 3135, 3137]
sizes = edges + [int(x) for x in rs.randint(130, 3400, size=12)]
worst = 0.0
t0 = time.time()
for n in sizes:
    a = rs.randn(n, n)
    a = a + a.T
    l0 = ctx.counter("resident_launches")
    w, v = nma.eigh(a)
    took = ctx.counter("resident_launches") - l0
    wr = np.linalg.eigvalsh(a)
    scale = np.abs(wr).max()
    ev = np.abs(w - wr).max() / scale
    res = np.abs(a @ v.T - v.T * w[None, :]).max() / scale
    orth = np.abs(v @ v.T - np.eye(n)).max()
    worst = max(worst, ev, res, orth)
    flag = "" if max(ev, res, orth) < 1e-11 and took == 1 else "   <-- CHECK"
    print(f"n={n:5d}: |dw| {ev:.1e}  resid {res:.1e}  orth {orth:.1e}  launches {took}{flag}", flush=True)
print(f"{len(sizes)} orders in {time.time() - t0:.0f} s, worst figure {worst:.1e}, take-overs {ctx.counter('resident_takeovers')}")
