"""Diagnostic build only (SC_EXTRA_HIPCC_FLAGS=-DBULGE_STAMPS): where a bulge-chasing task spends its cycles, at the bench's
batch.  python tools/bulge_stamps.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from springcraft_amd import _hip  # noqa: E402
from springcraft_amd.batch import DeviceBatchSolver  # noqa: E402
import springcraft_amd as sc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_atoms = 2000
box = 5.0 * n_atoms ** (1.0 / 3.0)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda().contiguous()
solver = DeviceBatchSolver(n_atoms, B, sc.HinsenForceField())
L = _hip.lib()
buf = (C.c_ulonglong * 6)()
solver.solve(coord)
torch.cuda.synchronize()
L.sc_dbg_bulge_stamps(buf)
solver.set_profiling(True)
solver.solve(coord)
torch.cuda.synchronize()
rc = L.sc_dbg_bulge_stamps(buf)
t = solver.last_timings()
e_in, e_out, d_in, end, tasks, tasks_k = [int(x) for x in buf]
print(f"rc {rc} batch {B}: bulge chasing {t['bulge_chasing_ms']:.1f} ms, {tasks} tasks ({tasks_k} with an off-diagonal block)")
print(f"  mean cycles since task start: E in LDS {e_in / max(1, tasks_k):.0f}, E stored {e_out / max(1, tasks_k):.0f}, "
      f"D in LDS {d_in / max(1, tasks):.0f}, end {end / max(1, tasks):.0f}")
cu_cycles = t['bulge_chasing_ms'] * 1e-3 * 2.4e9 * 256
print(f"  task-cycles / (CU-cycles of the phase) = mean concurrent tasks per CU: {end / cu_cycles:.2f}")
