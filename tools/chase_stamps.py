"""Diagnostic build only (-DCHASE_STAMPS): where a task of the persistent bulge chase with one sweep per workgroup
(k_bulge_chase: C2, C4, single structures) spends its cycles.  python tools/chase_stamps.py [n_atoms] [batch]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import springcraft_amd as sc  # noqa: E402
from springcraft_amd import _hip  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
box = 5.0 * N ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(N, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(N, B, sc.InvariantForceField(13.0))
solver.ctx.set_two_stage(True)
L = _hip.lib()
L.sc_dbg_chase_stamps.restype = C.c_int
buf = (C.c_ulonglong * 16)()
solver.solve(coord)
torch.cuda.synchronize()
L.sc_dbg_chase_stamps(buf)          # (clears the sums of the first solve)
solver.set_profiling(True)
solver.solve(coord)
torch.cuda.synchronize()
rc = L.sc_dbg_chase_stamps(buf)
t = solver.last_timings()
a = np.array(buf, dtype=np.float64)
n = 3 * N
print(f"rc {rc}  {B} x n = {n}: bulge chasing {t['bulge_chasing_ms']:.1f} ms of the profiled solve; chase form counters:",
      {k: solver.ctx.counter(k) for k in ("chase_launches", "chase_pair_launches", "stepwise_chases")})
names = ["", "wait for the predecessor sweep (poll + barrier)", "loads of both blocks + first products, to the first barrier",
         "right update of E + the new reflector", "column sums + u", "second wait (early hand-off)",
         "E stores issued + D into LDS", "D products", "w", "D update + stores issued", "store drain + barrier", "publish"]
wt = max(a[15], 1.0)
tot = a[1:12].sum() / wt
for i in range(1, 12):
    print(f"  {names[i]:62s} {a[i] / wt:8.0f} cycles  {100 * a[i] / wt / tot:5.1f} %")
print(f"  {'per task':62s} {tot:8.0f} cycles = {tot / 2.4e3:.2f} us at 2.4 GHz;  {wt / 4:.0f} tasks;"
      f"  the dependent chain of one matrix: 2 n = {2 * n} tasks")
