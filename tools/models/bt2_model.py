"""
NumPy model of the stage-2 back-transformation kernels (csrc/twostage.hip: k_dia_tfactor2, k_bt2_apply): the
fragment layout, the k-step lists that skip structural zeros, and the T = U^-1 formulation, checked against applying
the 64 Householder reflectors of a diamond one by one.  Pure index arithmetic: it is the specification the HIP
kernels are written against (the MFMA lane maps are restated by `mfma` below).

Round 3: a diamond (64 sweeps at one chase position) is applied as FOUR compact-WY blocks of 16 sweeps ("minis",
sweep tile st = 3, 2, 1, 0 in that order: Q = Q_0 Q_1 Q_2 Q_3, the last acts first).  A mini's reflectors span rows
16 st .. 16 st + 78 = 5 row tiles, for V^T Z and for (V T) W alike, so a diamond costs 4 x (20 + 20) = 160 MFMAs per
16 columns instead of the 80 + 104 of the single 64-sweep block (whose V T is a trapezoid, not a parallelogram):
executed / algorithmic flops 1.25 instead of 1.44.

    python tools/models/bt2_model.py
"""
import numpy as np

KB = 64  # reflector length
KG = 64  # sweeps per diamond
KM = 16  # sweeps per mini


def mfma(a_frag, b_frag, c):
    """v_mfma_f64_16x16x4_f64: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
    c[r][l] is D[i = 4 r + (l >> 4)][j = l & 15]."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a_frag[l]
        B[l >> 4, l & 15] = b_frag[l]
    D = A @ B
    out = c.copy()
    for r in range(4):
        for l in range(64):
            out[r, l] += D[4 * r + (l >> 4), l & 15]
    return out


def steps():
    """Issue order of one diamond = order of the fragments in memory: for st = 3, 2, 1, 0:
    20 x ("p1", st, rt, r) with rt = st + j // 4, r = j % 4          (W = V_st^T Z; B operand = Z tile rt register r)
    20 x ("p2", st, rt, r) with r = j // 5, rt = st + j % 5          (Z -= (V_st T_st) W; B operand = W register r).
    (In device memory the 64 lane values of fragments 2 p and 2 p + 1 are interleaved -- frag_off() in twostage.hip --
    so that a lane fetches both with one 16-byte LDS read; the order of the fragments is the one modelled here.)"""
    out = []
    for st in (3, 2, 1, 0):
        out += [("p1", st, st + j // 4, j % 4) for j in range(20)]
        out += [("p2", st, st + j % 5, j // 5) for j in range(20)]
    return out


def make_diamond(rs, nrows_valid=127, absent=()):
    """V (128 x 64): reflector c in rows c .. c + 63 (unit first entry), tau (64).  `absent`: tau = 0 columns."""
    V = np.zeros((128, KG))
    tau = np.zeros(KG)
    for c in range(KG):
        L = min(KB, max(0, nrows_valid - c))
        if L < 1:
            continue
        v = np.zeros(L)
        v[0] = 1.0
        v[1:] = rs.randn(L - 1) * 0.3
        V[c:c + L, c] = v
        tau[c] = 0.0 if c in absent else 2.0 / (v @ v)
    return V, tau


def t_factor_recurrence(V, tau):
    """LAPACK dlarft (forward, columnwise)."""
    k = V.shape[1]
    G = V.T @ V
    T = np.zeros((k, k))
    for q in range(k):
        T[:q, q] = -tau[q] * (T[:q, :q] @ G[:q, q])
        T[q, q] = tau[q]
    return T


def t_factor_inverse16(V, tau):
    """T of one mini = (diag(1 / tau) + striu(V^T V))^-1 by back substitution, one column per lane; tau = 0 decoupled."""
    G = V.T @ V
    U = np.triu(G, 1)
    dead = tau == 0.0
    U[dead, :] = 0.0
    U[:, dead] = 0.0
    U[np.arange(KM), np.arange(KM)] = np.where(dead, 1.0, 1.0 / np.where(dead, 1.0, tau))
    X = np.zeros((KM, KM))
    for j in range(KM):
        for i in range(j, -1, -1):
            s = (1.0 if i == j else 0.0) - U[i, i + 1:j + 1] @ X[i + 1:j + 1, j]
            X[i, j] = s / U[i, i]
    X[dead, dead] = 0.0
    return X


def fragments(V, tau):
    frags = np.zeros((len(steps()), 64))
    vt = {}
    for st in range(4):
        cols = slice(16 * st, 16 * st + 16)
        vt[st] = V[:, cols] @ t_factor_inverse16(V[:, cols], tau[cols])
    for f, (kind, st, rt, r) in enumerate(steps()):
        for l in range(64):
            if kind == "p1":
                frags[f, l] = V[16 * rt + 4 * r + (l >> 4), 16 * st + (l & 15)]
            else:
                frags[f, l] = -vt[st][16 * rt + (l & 15), 4 * r + (l >> 4)]
    return frags


def apply_kernel(frags, Zwin):
    """One wave: 16 columns of the 128-row window; zt[rt][r][lane] = Z(16 rt + 4 r + (l >> 4), l & 15)."""
    zt = np.zeros((8, 4, 64))
    for rt in range(8):
        for r in range(4):
            for l in range(64):
                zt[rt, r, l] = Zwin[16 * rt + 4 * r + (l >> 4), l & 15]
    w = None
    for f, (kind, st, rt, r) in enumerate(steps()):
        if kind == "p1":
            if f % 40 == 0:
                w = np.zeros((4, 64))               # one accumulator per mini (a dependent chain issues at full rate)
            w = mfma(frags[f], zt[rt, r], w)
        else:
            zt[rt] = mfma(frags[f], w[r], zt[rt])
    out = np.zeros_like(Zwin)
    for rt in range(8):
        for r in range(4):
            for l in range(64):
                out[16 * rt + 4 * r + (l >> 4), l & 15] = zt[rt, r, l]
    return out


def main():
    rs = np.random.RandomState(0)
    assert len(steps()) == 160
    for case, (nv, absent) in enumerate([(127, ()), (127, (0, 5, 63)), (90, (17,)), (40, ()), (127, tuple(range(16, 32)))]):
        V, tau = make_diamond(rs, nv, absent)
        for st in range(4):
            cols = slice(16 * st, 16 * st + 16)
            d = np.abs(t_factor_recurrence(V[:, cols], tau[cols]) - t_factor_inverse16(V[:, cols], tau[cols])).max()
            assert d < 1e-12, (case, st, d)
        Z = rs.randn(128, 16)
        ref = Z.copy()
        for c in range(KG - 1, -1, -1):   # Q Z with Q = H_0 H_1 ... H_63: the last reflector acts first
            v = V[:, c]
            ref -= tau[c] * np.outer(v, v @ ref)
        got = apply_kernel(fragments(V, tau), Z)
        assert np.abs(got - ref).max() < 1e-12, (case, np.abs(got - ref).max())
        # every row tile outside rt = st .. st + 4 really is structurally zero, for V and for V T
        for st in range(4):
            cols = slice(16 * st, 16 * st + 16)
            vt = V[:, cols] @ t_factor_inverse16(V[:, cols], tau[cols])
            for rt in range(8):
                if not st <= rt <= st + 4:
                    assert np.all(V[16 * rt:16 * rt + 16, cols] == 0.0) and np.all(vt[16 * rt:16 * rt + 16] == 0.0)
        print("case", case, "ok: |kernel - reflectors| =", np.abs(got - ref).max())


if __name__ == "__main__":
    main()
