"""
GPU parity tests of the two-stage tridiagonalisation path of the eigensolver (csrc/twostage.hip: band reduction,
bulge chasing, diamond back-transformation), forced on with ``Context.set_two_stage(True)``; the same gates as
tests/test_eigh_gpu.py and direct comparison with the one-stage path.
"""
import numpy as np
import pytest

from oracle import enm_oracle as orc
from tests.util import check_eigenvalues, check_eigenvectors, generated, synthetic_coord

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import springcraft_amd
    from springcraft_amd import _hip

    ctx = _hip.context()
    ctx.set_two_stage(True)
    yield springcraft_amd
    ctx.set_two_stage(None)


def sym(rs, n):
    a = rs.randn(n, n)
    return a + a.T


@pytest.mark.parametrize("n", [256, 257, 258, 319, 320, 321, 383, 384, 385, 449, 500, 512, 513, 777, 1000, 1536, 2049])
def test_random_symmetric(sc, n):
    """Orders around multiples of the band width 64 and of the diamond width: short last panels, short last sweeps."""
    a = sym(np.random.RandomState(n), n)
    w, v = sc.nma.eigh(a)
    w_ref = np.linalg.eigvalsh(a)
    assert np.abs(w - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    check_eigenvectors(a, w, v, tol_res=1e-11, tol_orth=1e-11)
    w_only = sc.nma.eigh(a, eigenvectors=False)
    assert np.abs(w_only - w_ref).max() <= 1e-11 * np.abs(w_ref).max()


def test_special_matrices(sc):
    n = 600
    rs = np.random.RandomState(1)
    q, _ = np.linalg.qr(rs.randn(n, n))
    cases = {
        "identity": np.eye(n),
        "zero": np.zeros((n, n)),
        "diagonal": np.diag(np.arange(n, dtype=float)),
        "tridiagonal already": np.diag(rs.randn(n)) + np.diag(rs.randn(n - 1), 1) + np.diag(np.zeros(n - 1), -1),
        "banded (inside the band)": sum(np.diag(rs.randn(n - d), -d) for d in range(0, 40)),
        "rank one + identity": np.eye(n) + np.outer(q[:, 0], q[:, 0]) * 5.0,
        "clustered": (q * np.repeat([1.0, 2.0, 3.0], n // 3)) @ q.T,
        "graded": (q * np.logspace(-12, 3, n)) @ q.T,
    }
    for name, a in cases.items():
        a = np.tril(a) + np.tril(a, -1).T
        w, v = sc.nma.eigh(a)
        w_ref = np.linalg.eigvalsh(a)
        scale = max(np.abs(w_ref).max(), 1e-300)
        assert np.abs(w - w_ref).max() <= 1e-11 * scale, name
        r = np.abs(a @ v.T - v.T * w).max()
        assert r <= 1e-11 * scale, name
        assert np.abs(v @ v.T - np.eye(n)).max() <= 1e-11, name


def test_only_lower_triangle_is_read(sc):
    """np.linalg.eigh(UPLO='L') semantics (nma.py:61): garbage above the diagonal must not matter."""
    n = 400
    a = sym(np.random.RandomState(3), n)
    b = np.tril(a) + np.triu(np.full((n, n), 1e30), 1)
    w, _ = sc.nma.eigh(b)
    assert np.abs(w - np.linalg.eigvalsh(a)).max() <= 1e-11 * np.abs(a).max() * n


@pytest.mark.parametrize("n_atoms,ff", [(171, "inv"), (512, "inv"), (700, "hinsen")])
def test_anm_matches_oracle(sc, n_atoms, ff):
    coord = synthetic_coord(n_atoms, 7)
    if ff == "inv":
        model = sc.ANM(coord, sc.InvariantForceField(13.0))
        h, _ = orc.compute_hessian(coord, orc.invariant_ff(13.0))
    else:
        model = sc.ANM(coord, sc.HinsenForceField())
        h, _ = orc.compute_hessian(coord, orc.hinsen_ff())
    w, v = model.eigen()
    w_ref = np.linalg.eigvalsh(h)
    check_eigenvalues(w, w_ref, 6)
    assert np.abs(w[6:] - w_ref[6:]).max() <= 1e-11 * w_ref.max()
    check_eigenvectors(h, w, v, tol_res=1e-11, tol_orth=1e-11)


def test_config3_n2000_hinsen_golden(sc):
    """BASELINE config C3 through the two-stage path against the eigenvalues of the imported reference."""
    g = generated("c3_n2000_hinsen.npz")
    coord = synthetic_coord(2000, 0)
    w, v = sc.ANM(coord, sc.HinsenForceField()).eigen()
    check_eigenvalues(w, g["nocut_eigenvalues"], 6)
    assert np.abs(v @ v[:64].T - np.eye(6000)[:, :64]).max() <= 1e-10


def test_partial_spectrum(sc):
    n = 1500
    a = sym(np.random.RandomState(11), n)
    w_ref, v_ref = np.linalg.eigh(a)
    for lo, hi in ((0, 20), (700, 760), (1490, 1499)):
        w, v = sc.nma.eigh(a, subset_by_index=(lo, hi))
        assert np.abs(w - w_ref[lo:hi + 1]).max() <= 1e-11 * np.abs(w_ref).max()
        r = np.abs(a @ v.T - v.T * w).max()
        assert r <= 1e-10 * np.abs(w_ref).max()
        assert np.abs(v @ v.T - np.eye(hi - lo + 1)).max() <= 1e-10


def test_batched_solver_matches_one_stage(sc):
    """The batched device API (what bench.py times): both paths on the same 6 structures."""
    import torch

    from springcraft_amd import _hip
    from springcraft_amd.batch import DeviceBatchSolver

    n_atoms, batch = 300, 6
    coord = torch.from_numpy(np.stack([synthetic_coord(n_atoms, s) for s in range(batch)])).cuda()
    solver = DeviceBatchSolver(n_atoms, batch, sc.InvariantForceField(13.0))
    solver.ctx.set_two_stage(True)
    w2, v2 = solver.solve(coord)
    w2, v2 = w2.cpu().numpy().copy(), v2.cpu().numpy().copy()
    solver.ctx.set_two_stage(False)
    w1, _ = solver.solve(coord)
    w1 = w1.cpu().numpy()
    assert np.abs(w1 - w2).max() <= 1e-11 * np.abs(w1).max()
    for b in range(batch):
        h, _ = orc.compute_hessian(coord[b].cpu().numpy(), orc.invariant_ff(13.0))
        check_eigenvectors(h, w2[b], v2[b], tol_res=1e-11, tol_orth=1e-11)


def test_7cal_prody_and_bio3d_goldens(sc):
    """The reference's n = 5328 fixtures (tests/test_anm.py:87-142, :145-334) through the two-stage path."""
    from tests.util import load_csv, structures

    s = structures()
    ca = s["7cal_coord"]
    w, v = sc.ANM(ca, sc.InvariantForceField(13.0)).eigen()
    ref = load_csv("prody_anm_13_ang_cutoff_evals_7cal.csv.gz")
    check_eigenvalues(w, np.concatenate([np.zeros(6), ref[6:]]), 6, rtol=1e-5)
    assert np.abs(v[:32] @ v.T - np.eye(len(w))[:32]).max() <= 1e-10
    # mass-weighted Hinsen Hessian (Bio3D): rtol 5e-3 / atol 2e-3 as in the reference's test
    atoms = sc.AtomArray(len(ca))
    atoms.coord = ca
    atoms.res_name = s["7cal_res_name"]
    masses = load_csv("bio3d_mass_7cal.csv.gz")
    w, _ = sc.ANM(atoms, sc.HinsenForceField(), masses=masses).eigen()
    assert np.allclose(w[6:], load_csv("bio3d_anm_calpha_ff_evals_mw_7cal.csv.gz")[6:], rtol=5e-3, atol=2e-3)


def test_gnm_large(sc):
    """Kirchhoff matrices (integer entries, many equal eigenvalues) through the two-stage path."""
    n_atoms = 1100
    coord = synthetic_coord(n_atoms, 4)
    w, v = sc.GNM(coord, sc.InvariantForceField(10.0)).eigen()
    k, _ = orc.compute_kirchhoff(coord, orc.invariant_ff(10.0))
    w_ref = np.linalg.eigvalsh(k)
    check_eigenvalues(w, w_ref, 1)
    check_eigenvectors(k, w, v, tol_res=1e-11, tol_orth=1e-11)


def test_batched_ragged_order(sc):
    """Batch of matrices whose order is not a multiple of anything (n = 1030), device API, both paths compared."""
    import ctypes as C

    import torch

    from springcraft_amd import _hip

    n, batch = 1030, 5
    rs = np.random.RandomState(5)
    mats = np.stack([sym(rs, n) for _ in range(batch)])
    L = _hip.lib()
    ctx = _hip.Context(0)
    out = {}
    for mode in (True, False):
        ctx.set_two_stage(mode)
        a = torch.from_numpy(mats.copy()).cuda()
        w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        v = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        ctx.check(L.sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, batch, C.c_void_p(w.data_ptr()),
                                    C.c_void_p(v.data_ptr())))
        ctx.synchronize()
        out[mode] = (w.cpu().numpy(), v.cpu().numpy())
    for b in range(batch):
        w_ref = np.linalg.eigvalsh(mats[b])
        for mode in (True, False):
            w, v = out[mode][0][b], out[mode][1][b]
            assert np.abs(w - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
            check_eigenvectors(mats[b], w, v, tol_res=1e-11, tol_orth=1e-11)
    ctx.close()


def _lib_dbg():
    import ctypes as C

    from springcraft_amd import _hip

    L = _hip.lib()
    L.sc_dbg_set_chase.restype = C.c_int
    L.sc_dbg_set_chase.argtypes = [C.c_void_p, C.c_int, C.c_int]
    return L


@pytest.mark.parametrize("n,batch", [(1200, 8), (1030, 32), (1030, 1), (520, 3), (259, 9), (322, 2)])
@pytest.mark.parametrize("give_up", [0, 5, 100])
@pytest.mark.parametrize("form", [3, 4, 5])
def test_persistent_chase_and_resume(n, batch, give_up, form):
    """
    (form 3: two sweeps per workgroup through LDS, k_bulge_pair -- the give-up then also exercises the write-back of the
    LDS slots; form 4: one sweep per workgroup, k_bulge_chase; form 5, round 6: the same with the workgroups of a matrix on
    ALL XCDs, band entries handed on by write-through stores -- with eight matrices and more it is form 4; orders with
    partial last blocks and an odd sweep count included.)
    The persistent bulge chase (forced on through the debug entry) as ONE batched solve -- several matrices per XCD at
    batch 32, one XCD without a matrix at batch 3 -- and the event counters of the context: without the test hook the
    chase must finish every sweep itself (no time-out, no take-over by the per-wavefront launches); with the hook
    (every workgroup raises the flag after `give_up` tasks) the take-over must have run, and the eigenpairs are the
    same either way.
    """
    import ctypes as C

    import torch

    from springcraft_amd import _hip

    L = _lib_dbg()
    rs = np.random.RandomState(7 + n + batch)
    mats = np.stack([sym(rs, n) for _ in range(batch)])
    ctx = _hip.Context(0)
    try:
        ctx.set_two_stage(True)
        ctx.check(L.sc_dbg_set_chase(ctx.handle, form, give_up))
        a = torch.from_numpy(mats.copy()).cuda()
        w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        v = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        ctx.check(L.sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, batch, C.c_void_p(w.data_ptr()),
                                    C.c_void_p(v.data_ptr())))
        ctx.synchronize()
        cnt = {k: ctx.counter(k) for k in ("chase_launches", "chase_timeouts", "chase_incomplete", "chase_resumed",
                                           "chase_sweeps", "stepwise_chases", "chase_xcd_min", "chase_xcd_max",
                                           "chase_pair_launches")}
        assert cnt["chase_launches"] == 1 and cnt["stepwise_chases"] == 0, cnt
        assert cnt["chase_pair_launches"] == (1 if form == 3 else 0), cnt
        assert cnt["chase_timeouts"] == 0 and cnt["chase_incomplete"] == 0, cnt
        if give_up == 0:
            assert cnt["chase_resumed"] == 0 and cnt["chase_sweeps"] == batch * (n - 2), cnt
        elif give_up >= 100 and n < 500:
            # (small orders: a workgroup may be through before its hundredth task -- either outcome is legitimate)
            assert cnt["chase_resumed"] in (0, 1), cnt
        else:
            assert cnt["chase_resumed"] == 1 and cnt["chase_sweeps"] < batch * (n - 2), cnt
        # every matrix: residual and orthogonality on the device; eigenvalues of the first and last against LAPACK
        am = torch.from_numpy(mats).cuda()
        eye = torch.eye(n, dtype=torch.float64, device="cuda")
        for b in range(batch):
            r = am[b] @ v[b].T - v[b].T * w[b][None, :]
            assert float(r.abs().max()) <= 1e-10 * float(w[b].abs().max()), (b, cnt)
            assert float((v[b] @ v[b].T - eye).abs().max()) <= 1e-11, (b, cnt)
        for b in sorted({0, batch - 1}):
            w_ref = np.linalg.eigvalsh(mats[b])
            assert np.abs(w[b].cpu().numpy() - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    finally:
        ctx.close()


@pytest.mark.parametrize("n,batch,min_rows", [(700, 1, 128), (1030, 3, 200), (2101, 2, 128), (4700, 1, -1)])
def test_cooperative_panel_qr(n, batch, min_rows):
    """
    Stage 1 with the cooperative panel kernel (k_panel_coop: the workgroups of one launch own 256 rows of a panel each
    and exchange their sums as 16-byte records), forced down to short panels through the debug entry -- 3 to 9 workgroups
    per matrix, several matrices per launch, an odd order -- and by the default rule on one n = 4700 matrix (panels of
    300 rows and more: 19 workgroups -- two rounds of records per poll -- down to 2).  The counters say the kernel ran and no wait timed out; every member:
    residual, orthogonality, eigenvalues against LAPACK.
    """
    import ctypes as C
    import os

    import torch

    from springcraft_amd import _hip

    if min_rows < 0 and (os.environ.get("SPRINGCRAFT_QR_COOP") == "0" or os.environ.get("SPRINGCRAFT_QR_COOP_MIN")):
        pytest.skip("the default rule is overridden (tools/test_matrix.sh)")
    L = _hip.lib()
    L.sc_dbg_set_panel_coop.restype = C.c_int
    L.sc_dbg_set_panel_coop.argtypes = [C.c_void_p, C.c_int]
    rs = np.random.RandomState(11 + n + batch)
    mats = np.stack([sym(rs, n) for _ in range(batch)])
    ctx = _hip.Context(0)
    try:
        ctx.set_two_stage(True)
        ctx.check(L.sc_dbg_set_panel_coop(ctx.handle, min_rows))
        a = torch.from_numpy(mats.copy()).cuda()
        w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        v = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        ctx.check(L.sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, batch, C.c_void_p(w.data_ptr()),
                                    C.c_void_p(v.data_ptr())))
        ctx.synchronize()
        lo = (300 if batch < 4 else 6145) if min_rows < 0 else min_rows
        expected = sum(1 for p in range(n // 64 + 1) if n - (p + 1) * 64 >= max(lo, 65))
        assert expected > 0
        if batch > 1 and int(os.environ.get("SPRINGCRAFT_STAGE1_STREAMS") or 0) > 1:
            expected = 0   # a batch split over streams keeps to ONE panel kernel for all its parts (tools/test_matrix.sh)
        assert ctx.counter("panel_coop_launches") == expected, (ctx.counter("panel_coop_launches"), expected)
        assert ctx.counter("panel_coop_timeouts") == 0
        am = torch.from_numpy(mats).cuda()
        eye = torch.eye(n, dtype=torch.float64, device="cuda")
        for b in range(batch):
            r = am[b] @ v[b].T - v[b].T * w[b][None, :]
            assert float(r.abs().max()) <= 1e-10 * float(w[b].abs().max()), b
            assert float((v[b] @ v[b].T - eye).abs().max()) <= 1e-11, b
        for b in sorted({0, batch - 1}):
            w_ref = np.linalg.eigvalsh(mats[b])
            assert np.abs(w[b].cpu().numpy() - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    finally:
        ctx.close()


def test_chase_by_size_rule_counts(sc):
    """The automatic rule: a latency-bound batch takes the persistent chase, and the counters say it completed."""
    import os

    import torch

    from springcraft_amd.batch import DeviceBatchSolver

    if os.environ.get("SPRINGCRAFT_BULGE_PERSISTENT") is not None:
        pytest.skip("the size rule is overridden (tools/test_matrix.sh)")
    n_atoms, B = 400, 8
    coords = torch.from_numpy(np.stack([synthetic_coord(n_atoms, s) for s in range(B)])).cuda()
    s = DeviceBatchSolver(n_atoms, B, sc.InvariantForceField(13.0))
    s.ctx.set_two_stage(True)
    for _ in range(3):
        s.solve(coords)
    torch.cuda.synchronize()
    assert s.ctx.counter("chase_launches") == 3 and s.ctx.counter("chase_sweeps") == 3 * B * (3 * n_atoms - 2)
    assert s.ctx.counter("chase_timeouts") == 0 and s.ctx.counter("chase_resumed") == 0
    assert s.ctx.counter("chase_xcd_min") >= 1


def test_pair_chase_by_size_rule(sc):
    """
    The automatic rule on a batch that is bound by bytes (batch * n / 128 > 1100, at least 8 matrices): the chase runs in
    the pair form (k_bulge_pair: two sweeps per workgroup through LDS), completes every sweep itself, and the
    eigenpairs of every member are right.
    """
    import os

    import torch

    from springcraft_amd.batch import DeviceBatchSolver

    if os.environ.get("SPRINGCRAFT_BULGE_PERSISTENT") is not None or os.environ.get("SPRINGCRAFT_BULGE_PAIR") is not None:
        pytest.skip("the size rule is overridden (tools/test_matrix.sh)")
    n_atoms, B = 342, 144                      # n = 1026, batch * n / 128 = 1154
    coords_np = np.stack([synthetic_coord(n_atoms, 500 + s) for s in range(B)])
    s = DeviceBatchSolver(n_atoms, B, sc.HinsenForceField(13.0))
    s.ctx.set_two_stage(True)
    w, v = s.solve(torch.from_numpy(coords_np).cuda())
    s.finish()
    assert s.ctx.counter("chase_pair_launches") == 1 and s.ctx.counter("chase_launches") == 1
    assert s.ctx.counter("chase_sweeps") == B * (3 * n_atoms - 2)
    assert s.ctx.counter("chase_timeouts") == 0 and s.ctx.counter("chase_resumed") == 0
    eye = torch.eye(3 * n_atoms, dtype=torch.float64, device="cuda")
    for b in range(B):
        assert float((v[b] @ v[b].T - eye).abs().max()) <= 1e-11, b
    for b in (0, 77, B - 1):
        h, _ = orc.compute_hessian(coords_np[b], orc.hinsen_ff(13.0))
        w_ref = np.linalg.eigvalsh(h)
        wb, vb = w[b].cpu().numpy(), v[b].cpu().numpy()
        assert np.abs(wb - w_ref).max() <= 1e-11 * np.abs(w_ref).max(), b
        assert np.abs(h @ vb.T - vb.T * wb[None, :]).max() <= 1e-10 * np.abs(w_ref).max(), b



@pytest.mark.parametrize("n,batch,min_rows,fail_panel", [(1030, 3, 200, 2), (2101, 2, 128, 0), (4700, 1, -1, 5)])
def test_cooperative_panel_take_over(n, batch, min_rows, fail_panel):
    """
    Round 6: a wait of k_panel_coop that runs into its bound no longer fails the
    solve.  The test hook raises the matrices' abort flags in front of panel `fail_panel`: the cooperative launches from
    there on return at once (nothing stored), the take-over k_panel_serial behind each of them factors the panel from
    memory, the solve ends with LAPACK's eigenvalues and orthonormal vectors, the event is counted, and the context then
    keeps to the chunked launches (no cooperative launch in the second solve).
    """
    import ctypes as C
    import os

    import torch

    from springcraft_amd import _hip

    if os.environ.get("SPRINGCRAFT_QR_COOP") == "0" or os.environ.get("SPRINGCRAFT_QR_COOP_MIN") or \
            (batch > 1 and int(os.environ.get("SPRINGCRAFT_STAGE1_STREAMS") or 0) > 1):
        pytest.skip("the cooperative kernel's rule is overridden (tools/test_matrix.sh)")
    L = _hip.lib()
    for f in (L.sc_dbg_set_panel_coop, L.sc_dbg_set_panel_coop_fail):
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int]
    rs = np.random.RandomState(23 + n + batch)
    mats = np.stack([sym(rs, n) for _ in range(batch)])
    ctx = _hip.Context(0)
    try:
        ctx.set_two_stage(True)
        ctx.check(L.sc_dbg_set_panel_coop(ctx.handle, min_rows))
        ctx.check(L.sc_dbg_set_panel_coop_fail(ctx.handle, fail_panel))
        w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        v = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        for attempt in range(2):
            a = torch.from_numpy(mats.copy()).cuda()
            torch.cuda.synchronize()
            before = ctx.counter("panel_coop_launches")
            ctx.check(L.sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, batch, C.c_void_p(w.data_ptr()),
                                        C.c_void_p(v.data_ptr())))
            ctx.synchronize()              # no error: the solve was finished on the device
            launches = ctx.counter("panel_coop_launches") - before
            if attempt == 0:
                lo = (300 if batch < 4 else 6145) if min_rows < 0 else min_rows
                coop_panels = sum(1 for p in range(n // 64 + 1) if n - (p + 1) * 64 >= max(lo, 65))
                assert launches == coop_panels and coop_panels > fail_panel
                # every panel from the hooked one on, of every matrix
                assert ctx.counter("panel_coop_timeouts") == batch * (coop_panels - fail_panel)
            else:
                assert launches == 0       # the context keeps to the chunked launches after the event
            am = torch.from_numpy(mats).cuda()
            eye = torch.eye(n, dtype=torch.float64, device="cuda")
            for b in range(batch):
                r = am[b] @ v[b].T - v[b].T * w[b][None, :]
                assert float(r.abs().max()) <= 1e-10 * float(w[b].abs().max()), (attempt, b)
                assert float((v[b] @ v[b].T - eye).abs().max()) <= 1e-11, (attempt, b)
                w_ref = np.linalg.eigvalsh(mats[b])
                assert np.abs(w[b].cpu().numpy() - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
            ctx.check(L.sc_dbg_set_panel_coop_fail(ctx.handle, -1)) if attempt == 1 else None
    finally:
        ctx.close()


def test_device_solve_only_enqueues_at_n6000():
    """
    Round 6: a device-pointer solve has no stream synchronisation inside -- the outcome of the
    persistent chase and of the cooperative panel kernel is dealt with by take-over launches on the device and read at
    the next synchronising call.  One n = 6000 matrix (cooperative panels + persistent chase, the path that used to
    synchronise twice): the call returns while most of the solve is still ahead.
    """
    import ctypes as C
    import os
    import time

    import torch

    from springcraft_amd import _hip

    if os.environ.get("SPRINGCRAFT_BULGE_PERSISTENT") == "0":
        pytest.skip("the persistent chase is switched off: 12 000 launches per chase (tools/test_matrix.sh)")
    n = 6000
    rs = np.random.RandomState(3)
    m0 = sym(rs, n)
    L = _hip.lib()
    ctx = _hip.Context(0)
    try:
        ctx.set_two_stage(True)
        w = torch.empty((1, n), dtype=torch.float64, device="cuda")
        v = torch.empty((1, n, n), dtype=torch.float64, device="cuda")
        t_call, t_total = [], []
        for it in range(3):                # (the first solve allocates the workspace and raises LDS limits)
            a = torch.from_numpy(m0.copy()).cuda()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.check(L.sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, 1, C.c_void_p(w.data_ptr()),
                                        C.c_void_p(v.data_ptr())))
            t1 = time.perf_counter()
            ctx.synchronize()
            t2 = time.perf_counter()
            t_call.append(t1 - t0)
            t_total.append(t2 - t0)
        assert ctx.counter("chase_launches") == 3 and ctx.counter("chase_resumed") == 0
        assert min(t_call[1:]) < 0.3 * min(t_total[1:]), (t_call, t_total)
        w_ref = np.linalg.eigvalsh(m0)
        assert np.abs(w[0].cpu().numpy() - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    finally:
        ctx.close()


@pytest.mark.parametrize("n,batch,want_symm3", [(2049, 16, False), (2000, 4, True), (3000, 6, True)])
def test_symm_k_slices_in_a_batched_solve(n, batch, want_symm3):
    """
    The band reduction's X = A22 V with K slices, inside batched solves: 16 x n = 2049 has 528
    64-row tiles -- the regime in which the triangular-operand launches take FIVE slices (an odd order keeps k_symm3 out);
    4 x n = 2000 and 6 x n = 3000 (no multiple of 16) run k_symm3 with 16 resp. 11 slices per tile.  Every member:
    residual and orthogonality; eigenvalues of the first and last against LAPACK.
    """
    import ctypes as C
    import os

    import torch

    from springcraft_amd import _hip

    if os.environ.get("SPRINGCRAFT_SYMM_SPLIT") or os.environ.get("SPRINGCRAFT_SYMM3") == "0":
        pytest.skip("the slice rule / the kernel choice is overridden (tools/test_matrix.sh)")
    L = _hip.lib()
    rs = np.random.RandomState(n + batch)
    mats = np.stack([sym(rs, n) for _ in range(batch)])
    ctx = _hip.Context(0)
    try:
        ctx.set_two_stage(True)
        a = torch.from_numpy(mats.copy()).cuda()
        w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        v = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        ctx.check(L.sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, batch, C.c_void_p(w.data_ptr()),
                                    C.c_void_p(v.data_ptr())))
        ctx.synchronize()
        assert (ctx.counter("symm3_launches") > 0) == want_symm3
        am = torch.from_numpy(mats).cuda()
        eye = torch.eye(n, dtype=torch.float64, device="cuda")
        for b in range(batch):
            r = am[b] @ v[b].T - v[b].T * w[b][None, :]
            assert float(r.abs().max()) <= 1e-10 * float(w[b].abs().max()), b
            assert float((v[b] @ v[b].T - eye).abs().max()) <= 1e-11, b
        for b in sorted({0, batch - 1}):
            w_ref = np.linalg.eigvalsh(mats[b])
            assert np.abs(w[b].cpu().numpy() - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    finally:
        ctx.close()
