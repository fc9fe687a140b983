"""
NumPy model of k_bulge_pair's data flow (twostage.hip): team A chases sweep sA = 2p with its blocks kept in a ring of
three LDS slots, team B chases sweep sA + 1 two positions behind from the shifted slots and writes the band.  Checked
against the plain task-by-task chase on the band storage (what k_bulge_step does).  CPU only; run it after changing the
kernel's index arithmetic:   python tools/models/bulge_pair_model.py
"""
import numpy as np

KB = 64
LDAB = 128


def chase_len(n, s):
    return (n - 1 - s + KB - 1) // KB


def householder(alpha, xn2):
    if xn2 == 0.0 or alpha * alpha + xn2 < 1e-280:
        return alpha, 0.0, 0.0
    beta = -np.copysign(np.sqrt(alpha * alpha + xn2), alpha)
    return beta, (beta - alpha) / beta, 1.0 / (alpha - beta)


def task_core(first, L, E, D, x, vp, tau_p):
    """E (64,64) zero padded or None, D (64,64) lower (zero padded), x (64,) for the sweep start.  Returns E, D, vn, tau, beta."""
    if not first:
        u = tau_p * (E @ vp)
        E = E - np.outer(u, vp)
        x = E[:, 0].copy()
    x = np.where(np.arange(KB) < L, x, 0.0)
    beta, tau, scale = householder(x[0], float(np.sum(x[1:] ** 2)))
    vn = np.where(np.arange(KB) < L, x * scale, 0.0)
    vn[0] = 1.0
    if not first:
        z = tau * (E.T @ vn)
        E = E - np.outer(vn, z)
        E[:, 0] = 0.0
        E[0, 0] = beta
        E[L:, :] = 0.0
    Dl = np.tril(D)
    Ds = Dl + np.tril(D, -1).T
    pp = tau * (Ds @ vn)
    w = pp - 0.5 * tau * (pp @ vn) * vn
    Dn = np.tril(Ds - np.outer(vn, w) - np.outer(w, vn))
    Dn[L:, :] = 0.0
    return E, Dn, vn, tau, beta


def load_blocks(ab, n, s, k):
    r0 = s + 1 + k * KB
    L = min(KB, n - r0)
    D = np.zeros((KB, KB))
    for j in range(L):
        D[j:L, j] = ab[0:L - j, r0 + j]
    E = None
    if k > 0:
        E = np.zeros((KB, KB))
        for j in range(KB):
            E[:L, j] = ab[KB - j: KB - j + L, r0 - KB + j]
    x = np.zeros(KB)
    if k == 0:
        x[:L] = ab[1:1 + L, s]
    return r0, L, E, D, x


def store_blocks(ab, n, s, k, r0, L, E, D, beta):
    for j in range(L):
        ab[0:L - j, r0 + j] = D[j:L, j]
    if k > 0:
        for j in range(KB):
            ab[KB - j: KB - j + L, r0 - KB + j] = E[:L, j]
    else:
        ab[1:1 + L, s] = 0.0
        ab[1, s] = beta


def reference_chase(ab, n):
    ab = ab.copy()
    for s in range(n - 2):
        vp, tau_p = np.zeros(KB), 0.0
        for k in range(chase_len(n, s)):
            r0, L, E, D, x = load_blocks(ab, n, s, k)
            E, D, vn, tau, beta = task_core(k == 0, L, E, D, x, vp, tau_p)
            store_blocks(ab, n, s, k, r0, L, E, D, beta)
            vp, tau_p = vn, tau
    return ab


def flush(ab, n, sA, m, lenA, slots):
    """Give-up between steps (the kernel's `if (!s_go)` branch): what A left for B goes back to the band."""
    for pos in range(max(m - 2, 0), min(m, lenA)):
        E, D = slots[pos % 3]
        r0 = sA + 1 + pos * KB
        L = min(KB, n - r0)
        i_lo = 1 if (pos == m - 2 and pos >= 1) else 0     # row 0 of the older slot: B's task one up has rewritten it
        if pos > 0:
            for j in range(KB):
                for i in range(i_lo, L):
                    ab[KB + i - j, r0 - KB + j] = E[i, j]
        for j in range(L):
            for i in range(max(j, i_lo), L):
                ab[i - j, r0 + j] = D[i, j]


def pair_chase(ab, n, abort=None, refl=None):
    """abort = (sA, m): give up at the start of step m of the pair that owns sweep sA; returns (band, done counts)."""
    ab = ab.copy()
    done = {}
    sA = 0
    while sA <= n - 3:
        hasB = sA + 1 <= n - 3
        lenA = chase_len(n, sA)
        lenB = chase_len(n, sA + 1) if hasB else 0
        slots = {}
        vA = (np.zeros(KB), 0.0)
        vB = (np.zeros(KB), 0.0)
        nsteps = lenA + 2 if hasB else lenA
        for m in range(nsteps):
            if abort is not None and abort == (sA, m):
                if hasB:
                    flush(ab, n, sA, m, lenA, slots)
                    done[sA] = min(m, lenA)
                    done[sA + 1] = min(max(m - 2, 0), lenB)
                return ab, done
            # ---- team A, position m
            newA = None
            if m < lenA:
                r0, L, E, D, x = load_blocks(ab, n, sA, m)
                E, D, vn, tau, beta = task_core(m == 0, L, E, D, x, vA[0], vA[1])
                vA = (vn, tau)
                if refl is not None:
                    refl[(sA, m)] = vA
                if not hasB:
                    done[sA] = m + 1
                if hasB:
                    newA = (E if E is not None else np.full((KB, KB), np.nan), D)
                    if m == 0:
                        ab[1:1 + L, sA] = 0.0
                        ab[1, sA] = beta
                        ab[0, r0] = D[0, 0]        # row 0 of A's FIRST block has no task of B above it: final already
                    else:
                        ab[KB + 1: KB + L, r0 - KB] = 0.0          # the annihilated entries E(1.., 0)
                else:
                    store_blocks(ab, n, sA, m, r0, L, E, D, beta)
            # ---- team B, position m - 2 (reads the slots as they were at the start of the step)
            k = m - 2
            if hasB and 0 <= k < lenB:
                s = sA + 1
                r0 = s + 1 + k * KB
                L = min(KB, n - r0)
                Ek, Dk = slots[k % 3]
                has_next = k + 1 < lenA
                En, Dn = slots[(k + 1) % 3] if has_next else (None, None)
                D = np.zeros((KB, KB))
                E = np.zeros((KB, KB))
                x = np.zeros(KB)
                for i in range(KB):
                    for j in range(KB):
                        if i < KB - 1:
                            if j <= i:
                                D[i, j] = Dk[i + 1, j + 1]
                            E[i, j] = Ek[i + 1, j + 1] if j < KB - 1 else Dk[i + 1, 0]
                        else:
                            if j <= i:
                                D[i, j] = (En[0, j + 1] if j < KB - 1 else Dn[0, 0]) if has_next else 0.0
                            E[i, j] = En[0, 0] if (has_next and j == KB - 1) else 0.0
                    x[i] = Dk[i + 1, 0] if i < KB - 1 else (En[0, 0] if has_next else 0.0)
                D[L:, :] = 0.0
                E[L:, :] = 0.0
                x[L:] = 0.0
                first = k == 0
                E2, D2, vn, tau, beta = task_core(first, L, None if first else E, D, x, vB[0], vB[1])
                vB = (vn, tau)
                if refl is not None:
                    refl[(s, k)] = vB
                done[s] = k + 1
                store_blocks(ab, n, s, k, r0, L, E2, D2, beta)
            if newA is not None:
                slots[m % 3] = newA
        if hasB:
            done[sA] = lenA
        sA += 2
    return (ab, done) if abort is not None else ab


def main():
    rs = np.random.RandomState(0)
    for n in (70, 130, 131, 193, 200, 258, 321):
        ab = np.zeros((LDAB, n))
        for j in range(n):
            m = min(KB, n - 1 - j)
            ab[0:m + 1, j] = rs.standard_normal(m + 1)
        ref = reference_chase(ab, n)
        got = pair_chase(ab, n)
        assert np.array_equal(ref, got), f"band storage differs somewhere (n = {n})"   # incl. every stale / bulge entry
        err = np.abs(ref[:2] - got[:2]).max()
        # the reduced matrix is tridiagonal: nothing may be left below the first sub-diagonal
        junk = np.abs(got[2:KB + 1]).max()
        print(f"n = {n}: max |d, e difference| = {err:.2e}, left below the sub-diagonal {junk:.2e}")
        assert err < 1e-10 and junk < 1e-10, n
    print("pair model agrees with the task-by-task chase")
    # give-up + take-over: stop at step m of some pair, flush, finish with the per-wavefront tasks from the counts
    n = 200
    ab = np.zeros((LDAB, n))
    for j in range(n):
        mm = min(KB, n - 1 - j)
        ab[0:mm + 1, j] = rs.standard_normal(mm + 1)
    ref = reference_chase(ab, n)
    for sA in (0, 6, 70, 136, 196):
        for m in range(0, chase_len(n, sA) + 2):
            refl = {}
            part, done = pair_chase(ab, n, abort=(sA, m), refl=refl)
            # (the model's take-over needs the reflector of a sweep's last finished task; the kernel reads it from the diamond)
            fin = finish_with(part, n, done, refl)
            assert np.abs(fin[:2] - ref[:2]).max() < 1e-11 and np.abs(fin[2:KB + 1]).max() < 1e-11, (sA, m)
    print("give-up at any step + take-over from the counts reproduces the chase")


def finish_with(ab, n, done, refl):
    ab = ab.copy()
    t_max = 2 * (n - 3) + chase_len(n, n - 3) - 1
    refl = dict(refl)
    for t in range(t_max + 1):
        for s in range(0, n - 2):
            k = t - 2 * s
            if k < 0 or k >= chase_len(n, s) or k < done.get(s, 0):
                continue
            r0, L, E, D, x = load_blocks(ab, n, s, k)
            vp, tau_p = refl.get((s, k - 1), (np.zeros(KB), 0.0))
            E, D, vn, tau, beta = task_core(k == 0, L, E, D, x, vp, tau_p)
            store_blocks(ab, n, s, k, r0, L, E, D, beta)
            refl[(s, k)] = (vn, tau)
    return ab


if __name__ == "__main__":
    main()
