"""Stage-by-stage check of the two-stage tridiagonalisation against NumPy (needs a GPU)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
os.environ.setdefault("SPRINGCRAFT_TWO_STAGE", "1")
import springcraft_amd as sc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402

L = _hip.lib()
ctx = _hip.context()
L.sc_dbg_two_stage.restype = C.c_int
L.sc_dbg_two_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


def stages(a):
    n = len(a)
    band = np.zeros((n, 128))
    d = np.zeros(n)
    e = np.zeros(n)
    rc = L.sc_dbg_two_stage(ctx.handle, _hip.ptr(np.ascontiguousarray(a)), n, _hip.ptr(band), _hip.ptr(d), _hip.ptr(e))
    assert rc == 0, (rc, L.sc_last_error(ctx.handle))
    # band[j, dd] = AB(j + dd, j)
    bm = np.zeros((n, n))
    for dd in range(65):
        idx = np.arange(n - dd)
        bm[idx + dd, idx] = band[idx, dd]
        bm[idx, idx + dd] = band[idx, dd]
    beyond = np.abs(band[:, 65:]).max()
    t = np.diag(d) + np.diag(e[:-1], 1) + np.diag(e[:-1], -1)
    return bm, t, beyond


sizes = [int(x) for x in sys.argv[1:]] or [256, 257, 300, 511, 640, 1000]
for n in sizes:
    rs = np.random.RandomState(n)
    a = rs.randn(n, n)
    a = a + a.T
    w_ref = np.linalg.eigvalsh(a)
    scale = np.abs(w_ref).max()
    bm, t, beyond = stages(a)
    e1 = np.abs(np.linalg.eigvalsh(bm) - w_ref).max() / scale
    e2 = np.abs(np.linalg.eigvalsh(t) - w_ref).max() / scale
    t0 = time.perf_counter()
    w, v = sc.nma.eigh(a)
    dt = time.perf_counter() - t0
    ev = np.abs(w - w_ref).max() / scale
    res = np.abs(a @ v.T - v.T * w).max() / scale
    orth = np.abs(v @ v.T - np.eye(n)).max()
    print(f"n={n:5d} band eig err {e1:.2e} (beyond-band {beyond:.1e})  tri eig err {e2:.2e}  "
          f"full: eig {ev:.2e} resid {res:.2e} orth {orth:.2e}  {dt*1e3:.0f} ms", flush=True)
