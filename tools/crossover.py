"""One-stage vs two-stage tridiagonalisation over (N, batch): ms per step of the whole solve.  python tools/crossover.py"""
import subprocess
import sys

code = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import springcraft_amd as sc
from springcraft_amd.batch import DeviceBatchSolver
n_atoms, B, two = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
box = 5.0 * n_atoms ** (1 / 3)
coord = torch.from_numpy(np.stack([np.random.RandomState(s).rand(n_atoms, 3) * box for s in range(B)])).cuda()
solver = DeviceBatchSolver(n_atoms, B, sc.InvariantForceField(13.0))
solver.ctx.set_two_stage(bool(two))
solver.solve(coord); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    solver.solve(coord)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3 * 1e3:.1f}")
'''
ROWS = ((4000, 1), (3000, 1), (2500, 1), (2000, 1), (2000, 2), (2000, 3), (2000, 4), (1500, 1), (1500, 2), (1000, 1), (1000, 4), (1000, 8), (1000, 16), (500, 8), (500, 16), (500, 32), (500, 64), (342, 64), (171, 256))
if len(sys.argv) > 1 and sys.argv[1] == "--single":   # one structure at a time (round 6: the resident one-stage reduction)
    ROWS = tuple((int(x), 1) for x in sys.argv[2:]) or ((683, 1), (1000, 1), (1200, 1), (1400, 1), (1500, 1), (1700, 1), (2000, 1), (2500, 1))
for n_atoms, B in ROWS:
    row = []
    for two in (0, 1):
        r = subprocess.run([sys.executable, "-c", code, str(n_atoms), str(B), str(two)], capture_output=True, text=True, timeout=300)
        row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "fail")
    n = 3 * n_atoms
    auto = "two" if (n >= 512 and B * n * n >= max(1.7e7, 5.0e3 * n)) else "one"   # eigh.hip:two_stage_for
    if B == 1:
        auto = "two" if n > 7000 else "one"   # (one matrix, with k_sytrd_resident)
    print(f"N={n_atoms:5d} n={n:5d} B={B:3d}: one-stage {row[0]:>8s} ms   two-stage {row[1]:>8s} ms   (automatic: {auto}-stage)", flush=True)
