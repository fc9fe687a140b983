"""Deflation-heavy and degenerate matrices through k_sytrd_resident and the parallel deflation set-up of the D&C, at orders
where a single solve takes them (1536: rows in LDS, 2600: rows in registers): eigenvalues against LAPACK, residual,
orthogonality.   python tools/resident_structured.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from springcraft_amd import nma, _hip  # noqa: E402

ctx = _hip.context()
ctx.set_two_stage(False)
rs = np.random.RandomState(0)
worst = 0.0
for n in (1536, 2600):
    q, _ = np.linalg.qr(rs.randn(n, n))
    kinds = {
        "zero": np.zeros((n, n)),
        "identity": np.eye(n),
        "diag": np.diag(rs.randn(n)),
        "tridiagonal": np.diag(rs.randn(n)) + np.diag(rs.randn(n - 1), 1) + np.diag(rs.randn(n - 1), -1) * 0,
        "ones (rank 1)": np.ones((n, n)),
        "two blocks of ones": np.kron(np.eye(2), np.ones((n // 2, n // 2))),
        "four eigenvalues": (q * np.repeat([1.0, 2.0, 3.0, 4.0], n // 4)) @ q.T,
        "clustered 1 + 1e-13 k": (q * (1.0 + 1e-13 * np.arange(n))) @ q.T,
        "graded 1e-12 .. 1": (q * np.logspace(-12, 0, n)) @ q.T,
        "wilkinson": np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1),
        "six zero modes": None,
    }
    t = kinds["tridiagonal"]
    kinds["tridiagonal"] = np.triu(t) + np.triu(t, 1).T
    b = rs.randn(n, n - 6)
    kinds["six zero modes"] = b @ b.T
    for name, a in kinds.items():
        a = (a + a.T) / 2
        w, v = nma.eigh(a)
        wr = np.linalg.eigvalsh(a)
        scale = max(np.abs(wr).max(), 1e-300)
        ev = np.abs(w - wr).max() / scale
        res = np.abs(a @ v.T - v.T * w[None, :]).max() / scale
        orth = np.abs(v @ v.T - np.eye(n)).max()
        worst = max(worst, ev, res, orth)
        flag = "" if max(ev, res, orth) < 1e-11 else "   <-- CHECK"
        print(f"n={n} {name:24s}: |dw| {ev:.1e}  resid {res:.1e}  orth {orth:.1e}{flag}", flush=True)
print(f"worst figure {worst:.1e}; resident launches {ctx.counter('resident_launches')}, take-overs {ctx.counter('resident_takeovers')}")
