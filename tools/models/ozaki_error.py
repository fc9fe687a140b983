"""
Round 6: the error of an error-free-slice ("Ozaki") emulation of an f64 product on int8 matrix
cores, emulated in NumPy -- s slices of 7 bits per operand, A scaled per row and B per column by powers of two, the slice
products p + q <= s + 1 accumulated exactly (int64 here, int32 on the chip: 6000 x 127^2 < 2^31).  Operands shaped like the
solver's W = V^T Z: V a block of Householder reflectors (unit diagonal, decaying tails), Z orthonormal columns.
    python tools/models/ozaki_error.py
Reported: max |C - C_ref| / (|A| |B|) -- the componentwise measure against which an f64 GEMM has ~K eps / 2 worst case and
~sqrt(K) eps in practice -- and max |C - C_ref| / max |C_ref|.
"""
import numpy as np


def slices(x, s, axis):
    """x = 2^e (sum_p q_p 128^-p + rest), q_p integers in [-127, 127]; e per row (axis = 1) or per column (axis = 0)."""
    amax = np.abs(x).max(axis=axis, keepdims=True)
    e = np.where(amax > 0, np.floor(np.log2(np.where(amax > 0, amax, 1.0))) + 1, 0.0)
    y = x / np.exp2(e)                       # |y| < 1
    out = []
    for _ in range(s):
        y = y * 128.0
        q = np.trunc(y)
        out.append(q.astype(np.int64))
        y = y - q
    return out, e


def emulate(a, b, s):
    qa, ea = slices(a, s, 1)
    qb, eb = slices(b, s, 0)
    c = np.zeros((a.shape[0], b.shape[1]))
    for p in range(s):
        for q in range(s):
            if p + q <= s - 1:               # (0-based: the s (s + 1) / 2 leading pairs)
                c += (qa[p] @ qb[q]).astype(np.float64) * 128.0 ** -(p + q + 2)
    return c * np.exp2(ea) * np.exp2(eb)


def main():
    rs = np.random.RandomState(0)
    k, m, n = 3000, 64, 64
    # reflector block: unit entry on the diagonal band, tails of norm ~ 1 below
    v = rs.standard_normal((k, m)) / np.sqrt(k)
    for j in range(m):
        v[:j * 8, j] = 0.0
        v[j * 8, j] = 1.0
    z, _ = np.linalg.qr(rs.standard_normal((k, n)))
    a = v.T.copy()
    ref = np.asarray(a.astype(np.longdouble) @ z.astype(np.longdouble), dtype=np.float64)
    denom = np.abs(a) @ np.abs(z)
    f64 = a @ z
    print(f"K = {k}: f64 GEMM itself: max err / (|A||B|) = {np.abs(f64 - ref).max() / denom.max():.2e} (eps = 1.1e-16)")
    for s in (5, 6, 7, 8, 9):
        c = emulate(a, z, s)
        err = np.abs(c - ref)
        print(f"  s = {s} slices of 7 bits ({s * (s + 1) // 2:2d} int8 GEMMs): max err / (|A||B|) = {(err / denom).max():.2e}, "
              f"max err / max |C| = {err.max() / np.abs(ref).max():.2e}")


if __name__ == "__main__":
    main()
