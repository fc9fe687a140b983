"""Cubic lattices (exactly degenerate ENM spectra: 3000 eigenvalues, 10 distinct values) through both paths.  python tools/lattice_degeneracy.py"""
import numpy as np, sys
sys.path.insert(0, ".")
import springcraft_amd as sc
from oracle import enm_oracle as orc
from springcraft_amd import _hip
ctx = _hip.context()
for m, spacing, cut in ((8, 3.8, 5.5), (10, 3.8, 4.0), (12, 4.0, 7.0)):
    g = np.arange(m) * spacing
    coord = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    h, _ = orc.compute_hessian(coord, orc.invariant_ff(cut))
    wr = np.linalg.eigvalsh(h)
    n = len(h)
    for two in (False, True):
        ctx.set_two_stage(two)
        w, v = sc.ANM(coord, sc.InvariantForceField(cut)).eigen()
        print(f"lattice {m}^3 (n={n}) cutoff {cut} two={two}: eig {np.abs(w - wr).max() / wr.max():.1e} res {np.abs(h @ v.T - v.T * w[None, :]).max() / wr.max():.1e} orth {np.abs(v @ v.T - np.eye(n)).max():.1e}  distinct(1e-9) {len(np.unique(np.round(wr / wr.max(), 9)))}")
