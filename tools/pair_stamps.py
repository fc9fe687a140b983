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
buf = (C.c_ulonglong * 48)()
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
names = ["wait + barrier [0]", "block reads (A: loads or landed image, B: slots + store drain) + [1]", "E right + reflector [2,3]",
         "column sums + u + D image [4,5]", "E left + D products [6]", "w [7]", "D update (+ stores)"]
print(f"rc {rc}  N = {n_atoms} x {B}  loader waves {os.environ.get('SPRINGCRAFT_PAIR_LOADER', '0')}: bulge chasing "
      f"{t['bulge_chasing_ms']:.1f} ms, {v[8]} steps ({v[9]} common), "
      f"counters launches {solver.ctx.counter('chase_launches')} timeouts {solver.ctx.counter('chase_timeouts')}")
print(f"  {'cycles per common step':72s} {'thread 0 (A)':>12s} {'thread 256 (B)':>14s}")
tot = [0, 0]
for k, name in enumerate(names):
    b = v[16 + k] if k != 6 else v[16 + 13]
    print(f"  {name:72s} {v[k] / steps:12.0f} {b / steps:14.0f}")
    tot[0] += v[k]
    tot[1] += b
print(f"  {'step':72s} {tot[0] / steps:12.0f} {tot[1] / steps:14.0f}")
print(f"  inside [1]: team A, block reads issued + vp read {v[12] / steps:.0f}; team B: slot reads {v[26] / steps:.0f}, "
      f"store drain {v[27] / steps:.0f} cycles")
print(f"  thread 256 inside [0]: {v[16 + 7] / steps:.0f} cycles in the block that looks at the predecessor pair's counter; it polls in "
      f"{v[16 + 14] / max(1, v[8]):.2f} of the steps (early look: {os.environ.get('SPRINGCRAFT_PAIR_EARLY', '0')})")
if v[40]:
    ls = v[40]
    ln = ["wait for this step's E pieces", "barrier [0]", "barriers [1] [2]", "wait for this step's D pieces", "barriers [3] [4]",
          "E requests issued", "barriers [5] [6] [7]", "D requests issued (M0 write waits for the E pieces)"]
    print(f"  loader wave 8, lane 0, {ls} steps that fetch and are served:")
    for k, name in enumerate(ln):
        print(f"    {name:70s} {v[32 + k] / ls:12.0f}")
