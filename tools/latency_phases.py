"""Single-structure latency (the reference's own calling pattern, anm.py:150-167 -> nma.py:61): wall-clock and phase
breakdown of ONE structure at a time at N = 100, 300, 512, 1000, 2000 -- the automatic path, the one-stage path with and
without the one-launch reduction (k_sytrd_resident), the two-stage path from N = 512 -- the host API ANM.eigen() and
numpy.linalg.eigh of the same Hessian on the box's host cores."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

L = _hip.lib()
L.sc_dbg_set_resident.restype = C.c_int
L.sc_dbg_set_resident.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]


def coords(n, seed):
    return np.random.RandomState(seed).rand(n, 3) * 5.0 * n ** (1 / 3)


sizes = [int(a) for a in sys.argv[1:]] or [100, 300, 512, 1000, 2000]
for n_atoms in sizes:
    ff = sc.HinsenForceField() if n_atoms >= 2000 else sc.InvariantForceField(13.0)
    x = torch.from_numpy(coords(n_atoms, 0)[None]).cuda()
    variants = [("automatic", None, -1), ("one-stage, launches per column", False, 0), ("one-stage, resident", False, 1)]
    if n_atoms >= 512:
        variants.append(("two-stage", True, -1))
    for name, two, resident in variants:
        s = DeviceBatchSolver(n_atoms, 1, ff)
        s.ctx.set_two_stage(two)
        s.ctx.check(L.sc_dbg_set_resident(s.ctx.handle, resident, 0, 0))
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
        print(f"N={n_atoms} {name}: {min(ts) * 1e3:.2f} ms unprofiled; profiled phases: "
              + ", ".join(f"{k}={v:.1f}" if isinstance(v, float) else f"{k}={v}" for k, v in t.items()), flush=True)
        del s
    c = coords(n_atoms, 0)
    sc.ANM(c, ff).eigen()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        sc.ANM(c, ff).eigen()
        ts.append(time.perf_counter() - t0)
    h = sc.ANM(c, ff).hessian
    t0 = time.perf_counter()
    np.linalg.eigh(h)
    t_np = time.perf_counter() - t0
    print(f"N={n_atoms} host API ANM.eigen(): {min(ts) * 1e3:.2f} ms; numpy.linalg.eigh of the same {3 * n_atoms} x {3 * n_atoms} "
          f"Hessian on the host ({os.cpu_count()} cores visible): {t_np * 1e3:.1f} ms", flush=True)
