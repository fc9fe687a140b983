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
buf = (C.c_ulonglong * (64 * 8 * 9))()
rc = _hip.lib().sc_dbg_bt2_stamps(buf)
a = np.array(buf, dtype=np.float64).reshape(64, 8, 9)
print("rc", rc, "diamonds per wave", a[0, 0, 8])
names = ["slide end -> barrier 0 arrive (loop top)", "barrier 0 wait", "half 0: minis 3, 2 (+ DMA issue, deferred stores)",
         "vmcnt wait after half 0", "barrier 1 wait", "half 1: minis 1, 0 (+ DMA issue, row loads)",
         "vmcnt wait after half 1", "slide (shift, scatter)"]
per = a[:, :, :8] / np.maximum(a[:, :, 8:9], 1)
tot = per.sum(-1).mean()
for i, nm in enumerate(names):
    print(f"{nm:40s} mean {per[:, :, i].mean():9.0f} cyc  min {per[:, :, i].min():9.0f}  max {per[:, :, i].max():9.0f}  {100 * per[:, :, i].mean() / tot:5.1f} %")
print(f"{'per diamond':40s} mean {tot:9.0f} cyc")
