"""Diagnostic build only (-DBT2_CLOCK): the shader clock during the Q2 application of bench-sized solves.
python tools/bt2_clock.py [structures] [n_atoms] [solves]"""
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
K = int(sys.argv[3]) if len(sys.argv) > 3 else 6
box = 5.0 * N ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(N, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(N, B, sc.HinsenForceField())
L = _hip.lib()
L.sc_dbg_bt2_clock.restype = C.c_int
for it in range(K):
    solver.solve(coord)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 2)()
    rc = L.sc_dbg_bt2_clock(buf)
    cyc, ticks = buf[0], buf[1]
    print(f"solve {it}: rc {rc}  {cyc} shader cycles in {ticks / 100.0:.0f} us  ->  {cyc / max(ticks, 1) * 100.0:.0f} MHz "
          f"(ROLE={os.environ.get('SPRINGCRAFT_BT2_ROLE', '0')})", flush=True)
