"""Diagnostic build only (-DSYMM3_STAMPS): where the loader waves of k_symm3 wait during one C3 bench step."""
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
buf = (C.c_ulonglong * 4)()
solver.solve(coord)
torch.cuda.synchronize()
L.sc_dbg_symm3_stamps(buf)
solver.set_profiling(True)
solver.solve(coord)
torch.cuda.synchronize()
rc = L.sc_dbg_symm3_stamps(buf)
t = solver.last_timings()
w, bar, loop, n = [int(x) for x in buf]
print(f"rc {rc}  N = {n_atoms} x {B}: symm {t.get('symm_ms', 0):.1f} ms; first loader wave of every workgroup: "
      f"{100.0 * w / max(1, loop):.1f} % of its loop waiting for its group to land ({w / max(1, n):.0f} cycles per waited step, "
      f"a step = {loop / max(1, 2 * n):.0f} cycles), {100.0 * bar / max(1, loop):.1f} % at the step barriers")
