"""
GPU tests of the one-launch tridiagonalisation of ONE matrix (k_sytrd_resident, tridiag.hip: the rows of the matrix stay
in LDS, one exchange between the workgroups per column) -- the path the reference's own call takes, one structure at a
time (anm.py:150-167 -> nma.py:61) -- against LAPACK on the same matrix, through the Python API -> C ABI.  Gates as in
test_eigh_gpu.py (the solver reaches ~1e-14; pinned at 1e-11).
"""
import ctypes as C

import numpy as np
import pytest

from tests.util import check_eigenvectors

pytestmark = pytest.mark.gpu


@pytest.fixture()
def res():
    """(package, library, context, setter): the debug entry is reset to the default rule afterwards."""
    import os

    import springcraft_amd as sc
    from springcraft_amd import _hip

    if os.environ.get("SPRINGCRAFT_RESIDENT") == "0":
        pytest.skip("SPRINGCRAFT_RESIDENT=0: the kernel under test is switched off")
    L = _hip.lib()
    ctx = _hip.context()
    L.sc_dbg_set_resident.restype = C.c_int
    L.sc_dbg_set_resident.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]

    def counter(name):
        v = C.c_int64(0)
        ctx.check(L.sc_ctx_get_counter(ctx.handle, name.encode(), C.byref(v)))
        return v.value

    def set_(mode, hook=0, wgs=0):
        ctx.check(L.sc_dbg_set_resident(ctx.handle, mode, hook, wgs))

    ctx.set_two_stage(False)
    try:
        yield sc, counter, set_
    finally:
        set_(-1)
        ctx.set_two_stage(None)


def sym(seed, n):
    a = np.random.RandomState(seed).randn(n, n)
    return a + a.T


def check(sc, a, tol=1e-11):
    w, v = sc.nma.eigh(a)
    w_ref = np.linalg.eigvalsh(a)
    assert np.abs(w - w_ref).max() <= tol * max(np.abs(w_ref).max(), 1e-300)
    check_eigenvectors(np.tril(a) + np.tril(a, -1).T, w, v, tol_res=tol, tol_orth=tol)
    return w


@pytest.mark.parametrize("n", [128, 129, 255, 300, 513, 1000, 1537, 2048, 2049, 2561, 3072])
def test_whole_matrix_in_one_launch(res, n):
    """Orders on both sides of every internal boundary: 256 columns per thread chunk, 8 rows per workgroup, odd orders,
    the largest order with the rows in LDS (2048), the two register forms (up to 2560 and up to 3072, the largest that
    fits); eigenvalues only and the partial spectrum take the same reduction."""
    sc, counter, set_ = res
    a = sym(n, n)
    before, t0 = counter("resident_launches"), counter("resident_takeovers")   # (the context's counters are cumulative)
    w = check(sc, a)
    assert counter("resident_launches") == before + 1
    w_only = sc.nma.eigh(a, eigenvectors=False)
    assert np.abs(w_only - w).max() <= 1e-11 * np.abs(w).max()
    ws, vs = sc.nma.eigh(a, subset_by_index=(3, 40))
    assert np.abs(ws - w[3:41]).max() <= 1e-11 * np.abs(w).max()
    assert counter("resident_launches") == before + 3 and counter("resident_takeovers") == t0
    # same eigenvalues as the launches per column give
    set_(0)
    w_col = sc.nma.eigh(a, eigenvectors=False)
    assert counter("resident_launches") == before + 3
    assert np.abs(w_col - w).max() <= 1e-11 * np.abs(w).max()


@pytest.mark.parametrize("wgs", [64, 128, 256])
def test_workgroup_counts(res, wgs):
    """The number of workgroups is a free parameter above order / 8: same results with 64, 128, 256 of them."""
    sc, counter, set_ = res
    a = sym(7, 400)
    set_(1, 0, wgs)
    check(sc, a)


def test_trailing_matrix_of_a_larger_order(res):
    """n = 3300: four panels by the launches per column, the trailing matrix of order 3044 (lower triangle only valid
    behind the SYR2K updates) in one launch."""
    sc, counter, set_ = res
    a = sym(11, 3300)
    before = counter("resident_launches")
    check(sc, a)
    assert counter("resident_launches") == before + 1


def test_scaled_input_and_only_the_lower_triangle(res):
    """A matrix that the solver rescales (largest entry 1e200: the scaling pass touches the lower triangle only, the rows
    are then gathered from it) and junk in NumPy's upper triangle (UPLO = 'L')."""
    sc, counter, set_ = res
    a = sym(3, 500)
    w_ref = np.linalg.eigvalsh(a)
    w, v = sc.nma.eigh(a * 1e200)
    assert np.all(np.isfinite(w)) and np.abs(w / 1e200 - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    junk = a.copy()
    junk[np.triu_indices(500, 1)] = 1e3
    junk[5, 400] = np.nan
    w, v = sc.nma.eigh(junk)
    assert np.abs(w - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    check_eigenvectors(a, w, v, tol_res=1e-11, tol_orth=1e-11)


@pytest.mark.parametrize("n,hook", [(300, 1), (300, 2 + 150), (700, 2 + 0), (2300, 2 + 40), (3200, 1)])
def test_take_over(res, n, hook):
    """The take-over kernel behind every launch: a failed roll call (nothing stored yet; n = 3200: for a trailing matrix)
    and a wait lost in mid-run (the matrix is restored from its other triangle; n = 2300: the register form) still give
    LAPACK's eigenpairs; the event is counted; after a lost wait (at once) or three failed roll calls the context keeps
    to the launches per column until the debug entry re-arms it."""
    import os

    sc, counter, set_ = res
    if hook > 1 and n > int(os.environ.get("SPRINGCRAFT_RESIDENT_MAX", "3072")):
        pytest.skip("with this SPRINGCRAFT_RESIDENT_MAX the matrix is not reduced whole: a lost wait fails the solve (next test)")
    a = sym(n + hook, n)
    set_(1, hook)
    t0, l0 = counter("resident_takeovers"), counter("resident_launches")
    check(sc, a)
    assert counter("resident_takeovers") == t0 + 1 and counter("resident_launches") == l0 + 1
    if hook > 1:
        # a lost wait: the context stays away from the kernel at once (the hook is still set)
        check(sc, a)
        assert counter("resident_takeovers") == t0 + 1 and counter("resident_launches") == l0 + 1
        used = 1
    elif n <= 700:
        # a failed roll call may be a launch whose workgroups were dispatched late: the context gives up at the third
        check(sc, a)
        check(sc, a)
        assert counter("resident_takeovers") == t0 + 3 and counter("resident_launches") == l0 + 3
        assert counter("resident_rollcall_failures") >= 3
        check(sc, a)
        assert counter("resident_takeovers") == t0 + 3 and counter("resident_launches") == l0 + 3
        used = 3
    else:
        used = 1
    set_(1, 0)           # the debug entry re-arms it
    check(sc, a)
    assert counter("resident_takeovers") == t0 + used and counter("resident_launches") == l0 + used + 1


def test_lost_wait_in_a_trailing_matrix_fails_the_solve(res):
    """A trailing matrix cannot be restarted (its upper triangle is stale): a wait lost in mid-run there surfaces as
    LinAlgError through the deferred status, as a QL failure would; the next solve is clean."""
    sc, counter, set_ = res
    a = sym(13, 3200)
    set_(1, 2 + 500)
    with pytest.raises(np.linalg.LinAlgError):
        sc.nma.eigh(a)
    set_(1, 0)
    check(sc, sym(14, 300))


def test_two_streams_of_single_solves_do_not_compete(res):
    """
    Two contexts on two streams enqueue single-structure solves at the same time (orders 2100: rows in registers, one
    workgroup per CU; 1536: rows in LDS): launches of k_sytrd_resident on one device are chained by an event, so neither
    sits in its roll call with half of its workgroups -- no take-over, eigenvalues as from one stream.
    """
    import torch

    sc, counter, set_ = res
    from springcraft_amd.batch import DeviceBatchSolver

    for n_atoms in (700, 512):
        coord = np.random.RandomState(3).rand(2, n_atoms, 3) * 5.0 * n_atoms ** (1 / 3)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        solvers, xs = [], []
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                solvers.append(DeviceBatchSolver(n_atoms, 1, sc.InvariantForceField(13.0)))
                solvers[-1].ctx.set_two_stage(False)     # (whatever SPRINGCRAFT_TWO_STAGE says: the one-stage path is under test)
                xs.append(torch.from_numpy(coord[k][None]).cuda())
        torch.cuda.synchronize()
        for _ in range(4):
            for s, x, st in zip(solvers, xs, streams):
                with torch.cuda.stream(st):
                    s.solve(x)
        torch.cuda.synchronize()
        for k, s in enumerate(solvers):
            w, _ = s.finish()
            why = {c: s.ctx.counter(c) for c in ("resident_launches", "resident_takeovers", "resident_rollcall_failures",
                                                 "resident_lost_waits", "resident_lost_at")}
            assert why["resident_launches"] == 4 and why["resident_takeovers"] == 0, (n_atoms, k, why)
            h = sc.ANM(coord[k], sc.InvariantForceField(13.0)).hessian
            w_ref = np.linalg.eigvalsh(h)
            assert np.abs(w.cpu().numpy()[0] - w_ref).max() <= 1e-11 * np.abs(w_ref).max()


def test_two_host_threads_of_single_solves_do_not_compete(res):
    """
    The same with two HOST THREADS (ctypes releases the GIL: both are inside the library at once), each with its own
    context and stream: the wait for the predecessor, the launch and the record of a solve are one critical section
    (tridiag.hip: g_resident_mu), so two threads cannot both wait for the same predecessor and then run side by side.
    """
    import threading

    import torch

    sc, counter, set_ = res
    from springcraft_amd.batch import DeviceBatchSolver

    n_atoms, rounds = 512, 6
    coord = np.random.RandomState(5).rand(2, n_atoms, 3) * 5.0 * n_atoms ** (1 / 3)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    solvers, xs = [], []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            solvers.append(DeviceBatchSolver(n_atoms, 1, sc.InvariantForceField(13.0)))
            solvers[-1].ctx.set_two_stage(False)
            xs.append(torch.from_numpy(coord[k][None]).cuda())
    torch.cuda.synchronize()
    go, errors = threading.Barrier(2), []

    def work(k):
        try:
            torch.cuda.set_device(0)
            go.wait()
            with torch.cuda.stream(streams[k]):
                for _ in range(rounds):
                    solvers[k].solve(xs[k])
        except Exception as e:   # noqa: BLE001 -- reported by the assertion below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for k, s in enumerate(solvers):
        w, _ = s.finish()
        why = {c: s.ctx.counter(c) for c in ("resident_launches", "resident_takeovers", "resident_rollcall_failures",
                                             "resident_lost_waits")}
        assert why["resident_launches"] == rounds and why["resident_takeovers"] == 0, (k, why)
        w_ref = np.linalg.eigvalsh(sc.ANM(coord[k], sc.InvariantForceField(13.0)).hessian)
        assert np.abs(w.cpu().numpy()[0] - w_ref).max() <= 1e-11 * np.abs(w_ref).max()


@pytest.mark.parametrize("n", [1536, 2600])
@pytest.mark.parametrize("kind", ["ones", "four_eigenvalues", "clustered", "wilkinson", "six_zero_modes", "identity"])
def test_degenerate_spectra_at_single_solve_orders(res, n, kind):
    """Reflectors that vanish (identity), chains of deflation rotations (repeated and clustered eigenvalues: the D&C's
    set-up walks them by one thread, everything else by all threads), exact null spaces -- at the orders a single solve
    takes (1536: rows in LDS; 2600: rows in registers)."""
    sc, counter, set_ = res
    rs = np.random.RandomState(n)
    if kind == "ones":
        a = np.ones((n, n))
    elif kind == "identity":
        a = np.eye(n)
    elif kind == "wilkinson":
        a = np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    elif kind == "six_zero_modes":
        b = rs.randn(n, n - 6)
        a = b @ b.T
    else:
        q, _ = np.linalg.qr(rs.randn(n, n))
        lam = np.repeat([1.0, 2.0, 3.0, 4.0], n // 4) if kind == "four_eigenvalues" else 1.0 + 1e-13 * np.arange(n)
        a = (q * lam) @ q.T
        a = (a + a.T) / 2
    before, t0 = counter("resident_launches"), counter("resident_takeovers")   # (the context's counters are cumulative)
    w, v = sc.nma.eigh(a)
    assert counter("resident_launches") == before + 1 and counter("resident_takeovers") == t0
    w_ref = np.linalg.eigvalsh(a)
    scale = np.abs(w_ref).max()
    assert np.abs(w - w_ref).max() <= 1e-11 * scale
    assert np.abs(a @ v.T - v.T * w[None, :]).max() <= 1e-11 * scale
    assert np.abs(v @ v.T - np.eye(n)).max() <= 1e-11
