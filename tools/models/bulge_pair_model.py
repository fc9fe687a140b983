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


# ----------------------------------------------------------------------------------------------------------------
# Round 6: the loader waves of k_bulge_pair<1>.  The same chase with the three LDS slots as flat arrays, team A's blocks
# of the common steps arriving by emulated LDS-DMA instructions (destination M0 + immediate + 16 lane, source scalar base
# + per-lane offset + immediate: the kernel's own offset formulas), and every access to a slot placed in the phase of the
# step's barrier schedule it has in the kernel:
#   a  in front of [1]   A reads its blocks (landed image or the band), B reads its shifted blocks from the slots
#   c  [2] .. [3]        both teams write the right-updated E image (row-major, stride 65) into their slot
#   d  [3] .. [4]        both read it (column sums); A reads the landed triangle of D
#   e  [4] .. [5]        both write the row-packed image of D; the loaders request E(m + 1) into B's slot
#   f  [5] .. [6]        A writes its final E, both read the image of D
#   h  behind [7]        A writes its final D; the loaders request D(m + 1) into B's slot
# A request poisons its destination with NaN at once and delivers either at once or as late as the handshake allows
# (E: the next step's [0]; D: the next step's [3]); both must reproduce the task-by-task chase bit for bit, and every
# image a team reads back must still be what it wrote.
KSLOT_E = KB * (KB + 1)
KSLOT_D = 17 * 128
KSLOT = KSLOT_E + KSLOT_D


def pair_cc(jj):
    return 32 * jj - (jj >> 1) * ((jj - 1) >> 1)


def common_step(hasB, m, lenA, sA, n):
    return hasB and m >= 3 and m < lenA and sA + 1 + (m + 1) * KB <= n


def loader_offsets(lw):
    """Per-lane byte offsets of loader wave lw (pair_loader_run)."""
    d0 = 0 if lw == 0 else 1 + 4 * lw
    voff_e = np.zeros((8, 64), dtype=np.int64)
    voff_d = np.zeros((5, 64), dtype=np.int64)
    for lane in range(64):
        lane_off = 16 * lane if lane < 32 else 8 * (LDAB - 1) + 16 * (lane - 32)
        for x in range(8):
            voff_e[x, lane] = lane_off + 16 * (LDAB - 1) * (8 * lw + x) - 1024 * x + 3584
        for x in range(5):
            g = min(64 * (d0 + x) + lane, pair_cc(KB) - 1)
            j = 0
            while pair_cc(j + 1) <= g:
                j += 1
            voff_d[x, lane] = 8 * (j * LDAB + 2 * (g - pair_cc(j))) - 1024 * x + 2048
    assert voff_e.min() >= 0 and voff_d.min() >= 0
    return d0, voff_e, voff_d


class Dma:
    """One LDS-DMA instruction: 64 lanes x 16 bytes."""

    def __init__(self, m0, imm, voff, sbase):
        self.dst = (m0 + imm + 16 * np.arange(64)) // 8
        src = sbase + voff + imm
        assert np.all(src % 8 == 0) and np.all((m0 + imm) % 16 == 0)
        self.src = src // 8

    def poison(self, lds):
        lds[self.dst] = np.nan
        lds[self.dst + 1] = np.nan

    def deliver(self, lds, glob):
        lds[self.dst] = glob[self.src]
        lds[self.dst + 1] = glob[self.src + 1]


def pair_chase_loader(ab, n, late):
    """The pair chase with LDS slots and loader waves; late = deliver every request at the last moment."""
    store = np.ascontiguousarray(ab.T)  # AB(c + d, c) at c * LDAB + d: the kernel's band storage
    ab = store.T                        # (the model's (d, c) indexing: a view of it)
    glob = store.reshape(-1)            # (flat doubles, what the DMA addresses: a view as well)
    assert np.shares_memory(ab, glob)
    lds = np.full(3 * KSLOT, np.nan)
    offs = [loader_offsets(lw) for lw in range(4)]
    sA = 0
    while sA <= n - 3:
        hasB = sA + 1 <= n - 3
        lenA = chase_len(n, sA)
        lenB = chase_len(n, sA + 1) if hasB else 0
        vA = (np.zeros(KB), 0.0)
        vB = (np.zeros(KB), 0.0)
        nsteps = lenA + 2 if hasB else lenA
        pend_e, pend_d = [], []
        for m in range(nsteps):
            sl = lambda pos: (pos % 3) * KSLOT
            # ---- barrier [0]: E of this step has landed
            for q in pend_e:
                q.deliver(lds, glob)
            pend_e = []
            # ---- phase a
            doA = m < lenA
            k = m - 2
            doB = hasB and 0 <= k < lenB
            com = common_step(hasB, m, lenA, sA, n)
            if doA:
                r0, L, E, D, x = load_blocks(ab, n, sA, m)
                if com:
                    Eimg = lds[sl(m): sl(m) + KSLOT_E]
                    El = Eimg[:KB * KB].reshape(KB, KB).T.copy()          # (i, j) at j * 64 + i
                    assert np.array_equal(El, E), (sA, m, "landed E")
                    E = El
            if doB:
                s = sA + 1
                rB = s + 1 + k * KB
                LB = min(KB, n - rB)
                Ek = lds[sl(k): sl(k) + KSLOT_E].reshape(KB, KB + 1)
                Dk = lds[sl(k) + KSLOT_E: sl(k) + KSLOT]
                has_next = k + 1 < lenA
                En = lds[sl(k + 1): sl(k + 1) + KSLOT_E].reshape(KB, KB + 1)
                Dn = lds[sl(k + 1) + KSLOT_E: sl(k + 1) + KSLOT]
                rp = lambda A_, i, j: A_[i * (i + 1) // 2 + j]
                DB = np.zeros((KB, KB))
                EB = np.zeros((KB, KB))
                xB = np.zeros(KB)
                for i in range(min(LB, KB)):
                    for j in range(KB):
                        if i < KB - 1:
                            if j <= i:
                                DB[i, j] = rp(Dk, i + 1, j + 1)
                            EB[i, j] = Ek[i + 1, j + 1] if j < KB - 1 else rp(Dk, i + 1, 0)
                        else:
                            if j <= i:
                                DB[i, j] = (En[0, j + 1] if j < KB - 1 else rp(Dn, 0, 0)) if has_next else 0.0
                            EB[i, j] = En[0, 0] if (has_next and j == KB - 1) else 0.0
                    xB[i] = rp(Dk, i + 1, 0) if i < KB - 1 else (En[0, 0] if has_next else 0.0)
            # the arithmetic of both tasks (registers)
            if doA:
                EA2, DA2, vnA, tauA, betaA = task_core(m == 0, L, E, D, x, vA[0], vA[1])
                vA = (vnA, tauA)
            if doB:
                firstB = k == 0
                EB2, DB2, vnB, tauB, betaB = task_core(firstB, LB, None if firstB else EB, DB, xB, vB[0], vB[1])
                vB = (vnB, tauB)
            # ---- phase c: E images (what is written is a marker: the right-updated block is never needed again here)
            if doA and hasB and m > 0:
                lds[sl(m): sl(m) + KSLOT_E].reshape(KB, KB + 1)[:, :KB] = 1000.0 + m
            if doB and k > 0:
                lds[sl(k): sl(k) + KSLOT_E].reshape(KB, KB + 1)[:, :KB] = 2000.0 + k
            # ---- barrier [3]: D of this step has landed
            for q in pend_d:
                q.deliver(lds, glob)
            pend_d = []
            # ---- phase d
            if doA and hasB and m > 0:
                assert np.all(lds[sl(m): sl(m) + KSLOT_E].reshape(KB, KB + 1)[:, :KB] == 1000.0 + m), (sA, m, "A's E image")
            if doB and k > 0:
                assert np.all(lds[sl(k): sl(k) + KSLOT_E].reshape(KB, KB + 1)[:, :KB] == 2000.0 + k), (sA, m, "B's E image")
            if doA and com:
                Dl = np.zeros((KB, KB))
                Dreg = lds[sl(m) + KSLOT_E: sl(m) + KSLOT]
                for jj in range(KB):
                    for i in range(jj, KB):
                        Dl[i, jj] = Dreg[2 * pair_cc(jj) - jj + i]
                assert np.array_equal(Dl, np.tril(D)), (sA, m, "landed D")
            # ---- phase e: D images; the loaders request E(m + 1)
            fetch = common_step(hasB, m + 1, lenA, sA, n)
            if doA and hasB:
                lds[sl(m) + KSLOT_E: sl(m) + KSLOT_E + KB * (KB + 1) // 2] = 3000.0 + m
            if doB:
                lds[sl(k) + KSLOT_E: sl(k) + KSLOT_E + KB * (KB + 1) // 2] = 4000.0 + k
            r0n = sA + 1 + (m + 1) * KB
            slot_b = sl(m + 1) * 8
            if fetch:
                eb = ((r0n - KB) * LDAB + KB) * 8
                for lw in range(4):
                    d0, voff_e, voff_d = offs[lw]
                    m0 = slot_b + 1024 * 8 * lw + 3584
                    for x in range(8):
                        pend_e.append(Dma(m0, 1024 * x - 3584, voff_e[x], eb))
                for q in pend_e:
                    q.poison(lds)
                if not late:
                    for q in pend_e:
                        q.deliver(lds, glob)
            # ---- phase f
            if doA and hasB:
                assert np.all(lds[sl(m) + KSLOT_E: sl(m) + KSLOT_E + KB * (KB + 1) // 2] == 3000.0 + m), (sA, m, "A's D image")
                if m > 0:
                    lds[sl(m): sl(m) + KSLOT_E].reshape(KB, KB + 1)[:, :KB] = EA2
            if doB:
                assert np.all(lds[sl(k) + KSLOT_E: sl(k) + KSLOT_E + KB * (KB + 1) // 2] == 4000.0 + k), (sA, m, "B's D image")
            # ---- phase h: final D; stores; the loaders request D(m + 1)
            if doA:
                if hasB:
                    Dreg = lds[sl(m) + KSLOT_E: sl(m) + KSLOT]
                    for i in range(KB):
                        Dreg[i * (i + 1) // 2: i * (i + 1) // 2 + i + 1] = DA2[i, :i + 1]
                    if m == 0:
                        ab[1:1 + L, sA] = 0.0
                        ab[1, sA] = betaA
                        ab[0, r0] = DA2[0, 0]
                    else:
                        ab[KB + 1: KB + L, r0 - KB] = 0.0
                else:
                    store_blocks(ab, n, sA, m, r0, L, EA2, DA2, betaA)
            if doB:
                store_blocks(ab, n, sA + 1, k, rB, LB, EB2, DB2, betaB)
            if fetch:
                db = r0n * LDAB * 8
                for lw in range(4):
                    d0, voff_e, voff_d = offs[lw]
                    m0 = slot_b + KSLOT_E * 8 + 1024 * d0 + 2048
                    for x in range(5 if lw == 0 else 4):
                        pend_d.append(Dma(m0, 1024 * x - 2048, voff_d[x], db))
                for q in pend_d:
                    q.poison(lds)
                if not late:
                    for q in pend_d:
                        q.deliver(lds, glob)
        assert not pend_e and not pend_d
        sA += 2
    return np.array(ab)


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
    for n in (200, 321, 450):
        ab = np.zeros((LDAB, n))
        for j in range(n):
            mm = min(KB, n - 1 - j)
            ab[0:mm + 1, j] = rs.standard_normal(mm + 1)
        ref = reference_chase(ab, n)
        for late in (False, True):
            assert np.array_equal(ref, pair_chase_loader(ab, n, late)), (n, late)
    print("loader-wave protocol (LDS slots, emulated LDS-DMA, early and late delivery) agrees with the task-by-task chase")
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
