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


@pytest.mark.parametrize("give_up", [0, 5, 300])
def test_persistent_chase_and_resume(give_up):
    """
    The persistent bulge chase (forced), alone and giving up after `give_up` tasks per workgroup: the per-wavefront
    launches then finish the chase from the published progress counters.  Own process: the library reads the switches once.
    """
    import os
    import subprocess
    import sys

    code = r'''
import numpy as np
import springcraft_amd as sc
from springcraft_amd import _hip
rs = np.random.RandomState(7)
for n, batch in ((1030, 1), (520, 3)):
    mats = []
    for b in range(batch):
        a = rs.standard_normal((n, n)); mats.append(0.5 * (a + a.T))
    ctx = _hip.context()
    ctx.set_two_stage(True)
    for a in mats:
        w, v = sc.nma.eigh(a)
        w_ref = np.linalg.eigvalsh(a)
        assert np.abs(w - w_ref).max() <= 1e-11 * np.abs(w_ref).max(), np.abs(w - w_ref).max()
        assert np.abs(a @ v.T - v.T * w[None, :]).max() <= 1e-10 * np.abs(w_ref).max()
        assert np.abs(v @ v.T - np.eye(n)).max() <= 1e-11
print("ok")
'''
    env = dict(os.environ, SPRINGCRAFT_BULGE_PERSISTENT="2", SPRINGCRAFT_BULGE_GIVE_UP=str(give_up))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
