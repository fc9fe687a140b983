"""
NumPy specification of the Sturm count in springcraft_amd/csrc/stein.hip:k_sturm_range (round 5): the number of
eigenvalues of the symmetric tridiagonal matrix (d, e) below a shift x, from the sign changes of the division-free
sequence

    p_0 = 1,   p_i = a_i p_{i-1} - b_i p_{i-2},   a_i = (d_i - x) s,   b_i = (e_{i-1} s)^2,   b_1 = 0,

with the rows scaled by s = 1 / (Gershgorin radius + 2 max|e|), the last two members re-normalised by a power of two every
4 rows (the kernel counts rows inside 64-row chunks: the phase differs, the bound does not), and an exact zero replaced by a tiny value of the sign opposite to its predecessor (what dstebz's q = -pivmin does
in the ratio form q_i = p_i / p_{i-1}), so that a decoupled block behind it (b = 0) starts afresh.

`count_ratio` is the ratio form the kernel used before (LAPACK dlaebz's recurrence), kept as the comparison.  What the
product form gives up: rows whose entries and distance to the shift are below ~1e-60 of the matrix' norm underflow the
sequence and then count as decoupled zeros; the ratio form keeps relative accuracy on such graded matrices.
"""
import numpy as np

TINY = 2.0 ** -900


def scale_of(d, e):
    d = np.asarray(d, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    n = len(d)
    el = np.concatenate(([0.0], np.abs(e)))[:n]
    er = np.concatenate((np.abs(e), [0.0]))[:n]
    span = max(abs(float(np.min(d - el - er))), abs(float(np.max(d + el + er)))) if n else 0.0
    emax = float(np.max(np.abs(e))) if len(e) else 0.0
    return 1.0 / max(span + 2.0 * emax, 1e-300)


def count_product(d, e, x, sc=None):
    d = np.asarray(d, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    n = len(d)
    if sc is None:
        sc = scale_of(d, e)
    xs = x * sc
    cnt = 0
    p2, p1 = 1.0, 1.0
    for i in range(n):
        a = d[i] * sc - xs
        b = (e[i - 1] * sc) ** 2 if i > 0 else 0.0
        p0 = float(np.float64(a) * np.float64(p1) - np.float64(b * p2))   # (the kernel fuses the multiply-add)
        if p0 == 0.0:
            p0 = -np.copysign(TINY, p1)
        cnt += int(np.signbit(p0) != np.signbit(p1))
        p2, p1 = p1, p0
        if i % 4 == 3:
            _, ex = np.frexp(max(abs(p1), abs(p2)))
            p1 = float(np.ldexp(p1, -int(ex)))
            p2 = float(np.ldexp(p2, -int(ex)))
    return cnt


def count_ratio(d, e, x):
    d = np.asarray(d, dtype=np.float64)
    e = np.asarray(e, dtype=np.float64)
    n = len(d)
    emax = float(np.max(np.abs(e))) if len(e) else 0.0
    pivmin = max(2.2250738585072014e-308 * max(1.0, emax * emax), 1e-290)
    cnt = 0
    q = d[0] - x
    if abs(q) < pivmin:
        q = -pivmin
    cnt += q < 0.0
    for i in range(1, n):
        q = d[i] - x - e[i - 1] * e[i - 1] / q
        if abs(q) < pivmin:
            q = -pivmin
        cnt += q < 0.0
    return int(cnt)
