"""Diagnostic battery for the device eigensolver (prints a table; used during bring-up)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from springcraft_amd import nma, _hip  # noqa: E402
from oracle import enm_oracle as orc  # noqa: E402


def check(name, a, vectors=True):
    n = len(a)
    t0 = time.time()
    try:
        if vectors:
            w, v = nma.eigh(a)
        else:
            w = nma.eigh(a, eigenvectors=False)
            v = None
    except Exception as e:  # noqa: BLE001
        print(f"{name:28s} n={n:5d} FAILED: {type(e).__name__}: {e}")
        return
    dt = time.time() - t0
    wr = np.linalg.eigvalsh(a)
    scale = max(np.abs(wr).max(), 1e-300)
    ev_err = np.abs(w - wr).max() / scale
    msg = f"{name:28s} n={n:5d} t={dt*1e3:8.1f}ms  |dw|/|w|max={ev_err:.2e}"
    if v is not None:
        al = np.tril(a) + np.tril(a, -1).T
        res = np.abs(al @ v.T - v.T * w[None, :]).max() / scale
        orth = np.abs(v @ v.T - np.eye(n)).max()
        msg += f"  resid={res:.2e}  orth={orth:.2e}"
    if np.isnan(w).any():
        msg += "  NaN!"
    import ctypes as C
    t6 = (C.c_double * 6)()
    ctx = _hip.context()
    _hip.lib().sc_last_eigh_timings(ctx.handle, t6)
    msg += f"  [tri {t6[0]:.1f} dc {t6[1]:.1f} bt {t6[2]:.1f} | symv {t6[3]:.1f} syr2k {t6[4]:.1f} ms]"
    print(msg, flush=True)


def main():
    print(_hip.context().info())
    _hip.lib().sc_ctx_set_profiling(_hip.context().handle, 1)
    rs = np.random.RandomState(0)
    sizes = [int(s) for s in sys.argv[1:]] or [1, 2, 3, 5, 20, 31, 32, 33, 64, 65, 100, 129, 200, 300, 500, 1000, 1536]
    for n in sizes:
        a = rs.randn(n, n)
        a = a + a.T
        check("random", a)
        check("random (values only)", a, vectors=False)
    for n in (64, 200, 500):
        if n > max(sizes):
            continue
        check("identity", np.eye(n))
        check("diag", np.diag(rs.randn(n)))
        t = np.diag(rs.randn(n)) + np.diag(rs.randn(n - 1), 1)
        t = t + np.triu(t, 1).T
        check("tridiagonal", t)
        q, _ = np.linalg.qr(rs.randn(n, n))
        lam = np.repeat(rs.randn(n // 4 + 1), 4)[:n]
        check("4-fold degenerate", (q * lam) @ q.T)
        check("rank-1 + I", np.eye(n) + np.outer(q[:, 0], q[:, 0]) * 5)
        w = np.zeros((n, n))
        idx = np.arange(n - 1)
        w[idx, idx + 1] = w[idx + 1, idx] = 1.0   # glued Wilkinson-like
        w[idx, idx] = np.abs(np.arange(n - 1) - n // 2)
        check("wilkinson", w)
    for n_atoms in (20, 100, 171, 512):
        if 3 * n_atoms > max(sizes) * 3:
            continue
        coord = orc.synthetic_coord(n_atoms, 1)
        h, _ = orc.compute_hessian(coord, orc.invariant_ff(13.0))
        check(f"ANM inv13 N={n_atoms}", h)
        h, _ = orc.compute_hessian(coord, orc.hinsen_ff())
        check(f"ANM hinsen N={n_atoms}", h)
        k, _ = orc.compute_kirchhoff(coord, orc.invariant_ff(13.0))
        check(f"GNM inv13 N={n_atoms}", k)


if __name__ == "__main__":
    main()
