"""Diagnostic build only (SC_EXTRA_HIPCC_FLAGS=-DPAIR_STAMPS): where a step of the pair chase (k_bulge_pair) spends its
cycles.   python tools/pair_stamps.py [n_atoms] [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd import _hip  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402
import springcraft_amd as sc  # noqa: E402

n_atoms = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
box = 5.0 * n_atoms ** (1.0 / 3.0)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda().contiguous()
solver = DeviceBatchSolver(n_atoms, B, sc.HinsenForceField())
solver.ctx.set_two_stage(True)
L = _hip.lib()
L.sc_dbg_set_chase(solver.ctx.handle, 3, 0)           # persistent chase always, pair form
buf = (C.c_ulonglong * 16)()
solver.solve(coord)
torch.cuda.synchronize()
L.sc_dbg_pair_stamps(buf)
solver.set_profiling(True)
solver.solve(coord)
torch.cuda.synchronize()
rc = L.sc_dbg_pair_stamps(buf)
t = solver.last_timings()
v = [int(x) for x in buf]
steps = max(1, v[9])   # the stamps cover the common steps (pair_step_full) only
names = ["wait+go [0]", "E loads issued / B: slot reads + store drain [1]", "D loads issued, E right + reflector [2,3]",
         "column sums + u + D image [4,5]", "E left + D products [6]", "w [7]", "D update + stores"]
print(f"rc {rc}  N = {n_atoms} x {B}: bulge chasing {t['bulge_chasing_ms']:.1f} ms, {steps} steps ({v[9]} with both teams at work), "
      f"counters launches {solver.ctx.counter('chase_launches')} timeouts {solver.ctx.counter('chase_timeouts')}")
tot = 0
for k, name in enumerate(names):
    print(f"  {name:72s} {v[k] / steps:8.0f} cycles")
    tot += v[k]
print(f"  {'step':32s} {tot / steps:8.0f} cycles")
print(f"  inside [1]: team A, E loads issued + vp read {v[12] / steps:.0f}; team B (thread 256): slot reads {v[10] / steps:.0f}, "
      f"store drain {v[11] / steps:.0f}; team B, D update + stores {v[13] / steps:.0f} cycles")
