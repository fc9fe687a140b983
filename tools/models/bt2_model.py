"""
NumPy model of the stage-2 back-transformation kernels (csrc/twostage.hip: k_dia_tfactor2, k_bt2_apply): the
fragment layout, the k-step lists that skip structural zeros, and the T = U^-1 formulation, checked against applying
the 64 Householder reflectors of a diamond one by one.  Pure index arithmetic: it is the specification the HIP
kernels are written against (the MFMA lane maps are restated by `mfma` below).

    python tools/models/bt2_model.py
"""
import numpy as np

KB = 64  # reflector length
KG = 64  # sweeps per diamond


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


def p1_steps():
    """product 1 (W1 = V^T Z): (rt, r, st) in issue order f = 4 i + st, i = 4 (rt - st) + r: the four accumulators take
    turns; B operand = Z tile rt register r."""
    return [(st + i // 4, i % 4, st) for i in range(20) for st in range(4)]


def p2_steps():
    """product 2 (Z -= VT W1): (st, r, rt) in issue order; B operand = W1 tile st register r."""
    return [(st, r, rt) for st in range(4) for r in range(4) for rt in range(0, 5 + st)]


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
    G = V.T @ V
    T = np.zeros((KG, KG))
    for q in range(KG):
        T[:q, q] = -tau[q] * (T[:q, :q] @ G[:q, q])
        T[q, q] = tau[q]
    return T


def t_factor_inverse(V, tau):
    """T = (diag(1 / tau) + striu(V^T V))^-1, blocked 16 -> 32 -> 64 as the kernel does; tau = 0 columns decoupled."""
    G = V.T @ V
    U = np.triu(G, 1)
    dead = tau == 0.0
    U[dead, :] = 0.0
    U[:, dead] = 0.0
    U[np.arange(KG), np.arange(KG)] = np.where(dead, 1.0, 1.0 / np.where(dead, 1.0, tau))
    T = np.zeros((KG, KG))
    for b in range(4):   # 16 x 16 diagonal blocks by back substitution, one column per lane
        D = U[16 * b:16 * b + 16, 16 * b:16 * b + 16]
        X = np.zeros((16, 16))
        for j in range(16):
            for i in range(j, -1, -1):
                s = (1.0 if i == j else 0.0) - D[i, i + 1:j + 1] @ X[i + 1:j + 1, j]
                X[i, j] = s / D[i, i]
        T[16 * b:16 * b + 16, 16 * b:16 * b + 16] = X
    for lo in (0, 32):   # level 1
        a, c = slice(lo, lo + 16), slice(lo + 16, lo + 32)
        P = U[a, c] @ T[c, c]
        T[a, c] = -T[a, a] @ P
    a, c = slice(0, 32), slice(32, 64)   # level 2
    P = U[a, c] @ T[c, c]
    T[a, c] = -T[a, a] @ P
    T[dead, dead] = 0.0
    return T


def fragments(V, T):
    VT = V @ T
    f1 = np.zeros((len(p1_steps()), 64))
    for f, (rt, r, st) in enumerate(p1_steps()):
        for l in range(64):
            f1[f, l] = V[16 * rt + 4 * r + (l >> 4), 16 * st + (l & 15)]
    f2 = np.zeros((len(p2_steps()), 64))
    for f, (st, r, rt) in enumerate(p2_steps()):
        for l in range(64):
            f2[f, l] = -VT[16 * rt + (l & 15), 16 * st + 4 * r + (l >> 4)]
    return f1, f2


def apply_kernel(f1, f2, Zwin):
    """One wave: 16 columns of the 128-row window; zt[rt][r][lane] = Z(16 rt + 4 r + (l >> 4), l & 15)."""
    zt = np.zeros((8, 4, 64))
    for rt in range(8):
        for r in range(4):
            for l in range(64):
                zt[rt, r, l] = Zwin[16 * rt + 4 * r + (l >> 4), l & 15]
    w1 = np.zeros((4, 4, 64))
    for f, (rt, r, st) in enumerate(p1_steps()):
        w1[st] = mfma(f1[f], zt[rt, r], w1[st])
    for f, (st, r, rt) in enumerate(p2_steps()):
        zt[rt] = mfma(f2[f], w1[st, r], zt[rt])
    out = np.zeros_like(Zwin)
    for rt in range(8):
        for r in range(4):
            for l in range(64):
                out[16 * rt + 4 * r + (l >> 4), l & 15] = zt[rt, r, l]
    return out


def main():
    rs = np.random.RandomState(0)
    assert len(p1_steps()) == 80 and len(p2_steps()) == 104
    for case, (nv, absent) in enumerate([(127, ()), (127, (0, 5, 63)), (90, (17,)), (40, ())]):
        V, tau = make_diamond(rs, nv, absent)
        T0 = t_factor_recurrence(V, tau)
        T1 = t_factor_inverse(V, tau)
        assert np.abs(T0 - T1).max() < 1e-12, (case, np.abs(T0 - T1).max())
        Z = rs.randn(128, 16)
        ref = Z.copy()
        for c in range(KG - 1, -1, -1):   # Q Z with Q = H_0 H_1 ... H_63: the last reflector acts first
            v = V[:, c]
            ref -= tau[c] * np.outer(v, v @ ref)
        wy = Z - V @ (T0 @ (V.T @ Z))
        assert np.abs(wy - ref).max() < 1e-12
        f1, f2 = fragments(V, T1)
        got = apply_kernel(f1, f2, Z)
        assert np.abs(got - ref).max() < 1e-12, (case, np.abs(got - ref).max())
        # the skipped k-steps really are structural zeros
        VT = V @ T1
        for st in range(4):
            for rt in range(8):
                blk = V[16 * rt:16 * rt + 16, 16 * st:16 * st + 16]
                if not (max(0, rt - 4) <= st <= min(3, rt)):
                    assert np.all(blk == 0.0)
                if rt > 4 + st:
                    assert np.all(VT[16 * rt:16 * rt + 16, 16 * st:16 * st + 16] == 0.0)
        print("case", case, "ok: |T_rec - T_inv| =", np.abs(T0 - T1).max(), " |kernel - reflectors| =", np.abs(got - ref).max())


if __name__ == "__main__":
    main()
