"""Diagnostic build only (-DBT2_STAMPS -DBT2_ROLE_STAMPS, SPRINGCRAFT_BT2_ROLE=1): where an MFMA wave of k_bt2_role spends
a diamond's cycles, after one bench-sized solve.  python tools/bt2_role_stamps.py [structures] [n_atoms]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
box = 5.0 * N ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(N, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(N, B, sc.HinsenForceField())
solver.solve(coord)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 8 * 17))()
rc = _hip.lib().sc_dbg_bt2_stamps(buf)
a = np.array(buf, dtype=np.float64).reshape(64, 8, 17)[:, :4]
print("rc", rc, "diamonds per wave", a[0, 0, 16])
names = {}
for q in range(4):
    names[4 * q] = f"quarter {q}: MFMAs 0 .. 35 (+ the tail before)"
    names[4 * q + 1] = f"quarter {q}: wait in the barrier that opens quarter {(q + 1) % 4}"
names[14] = "quarter 3: fragments of the next diamond + MFMAs 36 .. 39"
names[15] = "slide"
per = a[:, :, :16] / np.maximum(a[:, :, 16:17], 1)
tot = per.sum(-1).mean()
for i in sorted(names):
    print(f"{names[i]:62s} mean {per[:, :, i].mean():8.0f} cyc  min {per[:, :, i].min():8.0f}  max {per[:, :, i].max():8.0f}  {100 * per[:, :, i].mean() / tot:5.1f} %")
print(f"{'per diamond (160 MFMAs = 10 240 pipe cycles)':62s} mean {tot:8.0f} cyc")
