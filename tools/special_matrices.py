"""ones(n, n), low rank, clustered spectra, extreme scales through the two-stage path.  python tools/special_matrices.py"""
import numpy as np, sys
sys.path.insert(0, ".")
import springcraft_amd as sc
from springcraft_amd import _hip
ctx = _hip.context(); ctx.set_two_stage(True)
n = 640
rs = np.random.RandomState(1)
q, _ = np.linalg.qr(rs.standard_normal((n, n)))
cases = {
  "zeros": np.zeros((n, n)), "identity": np.eye(n), "diag": np.diag(np.arange(n, dtype=float)),
  "ones (rank 1)": np.ones((n, n)), "rank 5": (lambda b: b @ b.T)(rs.standard_normal((n, 5))),
  "clustered": q @ np.diag(np.repeat([1.0, 2.0, 2.0 + 1e-13, 5.0], n // 4)) @ q.T,
  "tiny scale": 1e-200 * (lambda a: a + a.T)(rs.standard_normal((n, n))),
  "huge scale": 1e200 * (lambda a: a + a.T)(rs.standard_normal((n, n))),
  "banded": np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1) + 2 * np.eye(n),
}
for name, a in cases.items():
    a = 0.5 * (a + a.T)
    w, v = sc.nma.eigh(a)
    wr = np.linalg.eigvalsh(a)
    scale = max(np.abs(wr).max(), 1e-300)
    res = np.abs(a @ v.T - v.T * w[None, :]).max() / scale
    orth = np.abs(v @ v.T - np.eye(n)).max()
    print(f"{name:14s} eig err {np.abs(w - wr).max() / scale:.1e}  residual {res:.1e}  orth {orth:.1e}")
    if not (np.abs(w - wr).max() <= 1e-11 * scale and res <= 1e-10 and orth <= 1e-10): print("   ^^^ FAIL")
print("special matrices ok")
