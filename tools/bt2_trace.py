"""Diagnostic build only (SC_EXTRA_HIPCC_FLAGS=-DBT2_TRACE): issue time of every MFMA of one diamond of k_bt2_apply
(workgroup 0, all eight waves), after one bench-sized solve.  python tools/bt2_trace.py [structures] [n_atoms]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
box = 5.0 * N ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(N, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(N, B, sc.HinsenForceField())
solver.solve(coord)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 2 * 81))()
L = _hip.lib()
L.sc_dbg_bt2_trace.restype = C.c_int
rc = L.sc_dbg_bt2_trace(buf)
a = np.array(buf, dtype=np.float64).reshape(8, 2, 81)
print("rc", rc)
t0 = a[:, 0, 0].min()
for h in range(2):
    print(f"half {h}: cycles between consecutive MFMA issues (rows = waves 0 .. 7; 80 steps; last = end of half)")
    for wv in range(8):
        d = np.diff(a[wv, h])
        print(f"  w{wv} start {a[wv, h, 0] - t0:7.0f} total {a[wv, h, 80] - a[wv, h, 0]:6.0f}: " + " ".join(f"{x:4.0f}" for x in d))
