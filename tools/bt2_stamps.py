"""Diagnostic build only (SC_EXTRA_HIPCC_FLAGS=-DBT2_STAMPS): per-wave cycle sums of the segments of k_bt2_apply's
diamond loop, after one bench-sized solve.  python tools/bt2_stamps.py [structures] [n_atoms]"""
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
buf = (C.c_ulonglong * (64 * 8 * 17))()
rc = _hip.lib().sc_dbg_bt2_stamps(buf)
a = np.array(buf, dtype=np.float64).reshape(64, 8, 17)
print("rc", rc, "diamonds per wave", a[0, 0, 16])
names = []
for h in range(2):
    names += [f"half {h}: arrive at its barrier (since the last stamp)", f"half {h}: barrier wait", f"half {h}: MFMA steps  0 .. 15",
              f"half {h}: MFMA steps 16 .. 31", f"half {h}: MFMA steps 32 .. 47", f"half {h}: MFMA steps 48 .. 63",
              f"half {h}: MFMA steps 64 .. 79"]
names += ["vmcnt wait after half 1", "slide"]
per = a[:, :, :16] / np.maximum(a[:, :, 16:17], 1)
tot = per.sum(-1).mean()
for i, nm in enumerate(names):
    print(f"{nm:52s} mean {per[:, :, i].mean():8.0f} cyc  min {per[:, :, i].min():8.0f}  max {per[:, :, i].max():8.0f}  {100 * per[:, :, i].mean() / tot:5.1f} %")
print(f"{'per diamond':52s} mean {tot:8.0f} cyc")
print("per wave of workgroup 0 (rows = waves, columns = the sixteen segments):")
for wv in range(8):
    print("  wave", wv, " ".join(f"{per[0, wv, i]:6.0f}" for i in range(16)), f"  sum {per[0, wv].sum():7.0f}")
