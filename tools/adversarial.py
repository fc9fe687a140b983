"""Adversarial inputs through sc.nma.eigh on both tridiagonalisation paths (graph Laplacians with several components,
Hilbert, Wilkinson, arrowhead, collinear chain, NaN / Inf entries).  python tools/adversarial.py"""
import numpy as np, sys
sys.path.insert(0, ".")
import springcraft_amd as sc
from springcraft_amd import _hip
ctx = _hip.context()
rs = np.random.RandomState(3)
def check(name, a, two):
    ctx.set_two_stage(two)
    try:
        w, v = sc.nma.eigh(a)
    except Exception as e:
        print(f"{name:28s} two={two}: raised {type(e).__name__}: {str(e)[:80]}"); return
    if not np.isfinite(a).all():
        print(f"{name:28s} two={two}: returned, finite w {np.isfinite(w).all()} finite v {np.isfinite(v).all()}"); return
    wr = np.linalg.eigvalsh(a); scale = max(np.abs(wr).max(), 1e-300); n = len(a)
    print(f"{name:28s} two={two}: eig {np.abs(w-wr).max()/scale:.1e} res {np.abs(a@v.T - v.T*w[None,:]).max()/scale:.1e} orth {np.abs(v@v.T-np.eye(n)).max():.1e}")
n = 700
# graph Laplacians with many components and isolated vertices (integer entries, exact degeneracies)
adj = np.zeros((n, n)); 
for lo, hi in ((0, 200), (200, 450), (450, 460)):
    blk = (rs.rand(hi-lo, hi-lo) < 0.3).astype(float); blk = np.triu(blk, 1); adj[lo:hi, lo:hi] = blk + blk.T
lap = np.diag(adj.sum(1)) - adj
path = np.diag(np.r_[1, 2*np.ones(n-2), 1]) - np.diag(np.ones(n-1), 1) - np.diag(np.ones(n-1), -1)
hilbert = 1.0 / (np.arange(n)[:, None] + np.arange(n)[None, :] + 1.0)
wilk = np.diag(np.abs(np.arange(n) - n//2).astype(float)) + np.diag(np.ones(n-1), 1) + np.diag(np.ones(n-1), -1)
arrow = np.diag(np.arange(1, n+1, dtype=float)); arrow[0, :] = 1; arrow[:, 0] = 1
coll = np.c_[np.arange(n//3, dtype=float)*3.8, np.zeros(n//3), np.zeros(n//3)]
h_coll, _ = sc.compute_hessian(coll, sc.InvariantForceField(8.0))
nan_m = rs.standard_normal((n, n)); nan_m = nan_m + nan_m.T; nan_m[5, 3] = nan_m[3, 5] = np.nan
inf_m = rs.standard_normal((n, n)); inf_m = inf_m + inf_m.T; inf_m[7, 2] = inf_m[2, 7] = np.inf
for two in (False, True):
    check("laplacian, 3 comps + isolated", lap, two)
    check("path graph", path, two)
    check("hilbert", hilbert, two)
    check("wilkinson", wilk, two)
    check("arrowhead", arrow, two)
    check("collinear chain Hessian", h_coll, two)
    check("NaN entry", nan_m, two)
    check("Inf entry", inf_m, two)
