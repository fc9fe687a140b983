"""
GPU parity tests for the eigensolve half of the hot path (the HIP replacement of
``np.linalg.eigh`` at the reference's nma.py:61), called through the Python API -> C ABI.

Gates: eigenvalues within 1e-5 relative of the reference's for the
non-trivial modes and |lambda| <= 1e-9 lambda_max for the trivial ones; eigenvectors by residual
||A v - lambda v|| <= 1e-5 ||A||, orthogonality ||V V^T - I||_max <= 1e-8, and |<v, v_ref>| >= 1 - 1e-5
for well separated modes.  The solver actually reaches ~1e-14, which the tests also pin (1e-11).
"""
import numpy as np
import pytest

from oracle import enm_oracle as orc
from tests.util import (check_eigenvalues, check_eigenvectors, generated, load_csv, structures,
                        synthetic_coord)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import springcraft_amd

    return springcraft_amd


def sym(rs, n):
    a = rs.randn(n, n)
    return a + a.T


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 31, 32, 33, 63, 64, 65, 66, 127, 128, 129, 130, 200, 513, 1000])
def test_random_symmetric(sc, n):
    """Every size class: single leaf, odd splits, n = 0/1/2 mod the panel width, several D&C levels."""
    a = sym(np.random.RandomState(n), n)
    w, v = sc.nma.eigh(a)
    w_ref = np.linalg.eigvalsh(a)
    assert np.abs(w - w_ref).max() <= 1e-11 * max(np.abs(w_ref).max(), 1e-300)
    check_eigenvectors(a, w, v, tol_res=1e-11, tol_orth=1e-11)
    w_only = sc.nma.eigh(a, eigenvectors=False)
    assert np.abs(w_only - w_ref).max() <= 1e-11 * max(np.abs(w_ref).max(), 1e-300)


def test_only_lower_triangle_is_read(sc):
    """np.linalg.eigh(UPLO='L') semantics: garbage in the strict upper triangle must not matter."""
    rs = np.random.RandomState(5)
    a = sym(rs, 150)
    junk = a.copy()
    junk[np.triu_indices(150, 1)] = rs.randn(150 * 149 // 2) * 100
    w, v = sc.nma.eigh(junk)
    w_ref, _ = np.linalg.eigh(junk)  # also reads the lower triangle only
    assert np.abs(w - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    check_eigenvectors(a, w, v, tol_res=1e-11, tol_orth=1e-11)
    assert np.array_equal(junk[np.triu_indices(150, 1)], junk[np.triu_indices(150, 1)])  # input untouched


@pytest.mark.parametrize("kind", ["identity", "diag", "tridiagonal", "degenerate4", "rank1", "wilkinson",
                                  "zero", "graded"])
def test_structured(sc, kind):
    """Deflation-heavy spectra: exercise both deflation types, exact zeros and clustered poles."""
    n = 300
    rs = np.random.RandomState(11)
    q, _ = np.linalg.qr(rs.randn(n, n))
    if kind == "identity":
        a = np.eye(n)
    elif kind == "zero":
        a = np.zeros((n, n))
    elif kind == "diag":
        a = np.diag(rs.randn(n))
    elif kind == "tridiagonal":
        a = np.diag(rs.randn(n)) + np.diag(rs.randn(n - 1), 1)
        a = np.triu(a) + np.triu(a, 1).T
    elif kind == "degenerate4":
        a = (q * np.repeat(rs.randn(n // 4), 4)) @ q.T
    elif kind == "rank1":
        a = np.eye(n) + 5 * np.outer(q[:, 0], q[:, 0])
    elif kind == "wilkinson":
        a = np.diag(np.abs(np.arange(n) - n // 2).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    elif kind == "graded":
        a = (q * np.logspace(-12, 3, n)) @ q.T
    a = 0.5 * (a + a.T)
    w, v = sc.nma.eigh(a)
    w_ref = np.linalg.eigvalsh(a)
    scale = max(np.abs(w_ref).max(), 1e-300)
    assert np.abs(w - w_ref).max() <= 1e-11 * scale
    r = np.abs(a @ v.T - v.T * w[None, :]).max()
    assert r <= 1e-11 * max(scale, 1.0)
    assert np.abs(v @ v.T - np.eye(n)).max() <= 1e-11


@pytest.mark.parametrize("glue", [1e-4, 1e-8, 1e-10, 1e-12])
def test_glued_wilkinson_clustered_poles(sc, glue):
    """
    Clustered but NOT deflated poles in the secular equation: copies of the Wilkinson matrix W21+ glued
    by a small off-diagonal entry have eigenvalue clusters of width ~ glue -- the merges of the divide & conquer meet
    poles that are 1e-4 ... 1e-12 apart, too far to deflate, close enough to make a step-size stopping rule dangerous.
    Eigenvalues against LAPACK to a few ulps of the norm, eigenvectors by residual and orthogonality.
    """
    m, copies = 21, 16
    d = np.tile(np.abs(np.arange(m) - m // 2).astype(float), copies)
    e = np.ones(m * copies - 1)
    e[m - 1:: m] = glue
    n = m * copies
    a = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    w, v = sc.nma.eigh(a)
    w_ref = np.linalg.eigvalsh(a)
    norm = np.abs(w_ref).max()
    assert np.abs(w - w_ref).max() <= 5e-14 * norm, np.abs(w - w_ref).max() / norm
    assert np.abs(a @ v.T - v.T * w[None, :]).max() <= 1e-12 * norm
    assert np.abs(v @ v.T - np.eye(n)).max() <= 1e-12
    # a graded spectrum through a random rotation: poles over fifteen orders of magnitude in one merge
    rs = np.random.RandomState(3)
    q, _ = np.linalg.qr(rs.randn(400, 400))
    g = (q * np.logspace(-15, 0, 400)) @ q.T
    g = 0.5 * (g + g.T)
    wg = sc.nma.eigh(g, eigenvectors=False)
    assert np.abs(wg - np.linalg.eigvalsh(g)).max() <= 5e-14


# ---- the reference's eigen goldens ------------------------------------------------------------------

@pytest.mark.parametrize("cutoff", [4, 7])
def test_gnm_eigen_prody_1l2y(sc, cutoff):
    # reference test: tests/test_gnm.py:47-84 (values and vectors, sign fixed by the first component)
    ca = structures()["1l2y_coord"]
    w, v = sc.GNM(ca, sc.InvariantForceField(cutoff)).eigen()
    ref_w = load_csv(f"prody_gnm_{cutoff}_ang_cutoff_evals_1l2y.csv.gz")
    ref_v = load_csv(f"prody_gnm_{cutoff}_ang_cutoff_evecs_1l2y.csv.gz")
    v = v * np.sign(v[:, 0])[:, None]
    ref_v = ref_v * np.sign(ref_v[:, 0])[:, None]
    assert np.allclose(w[1:], ref_w[1:])
    assert w[1:].tolist() == pytest.approx(ref_w[1:].tolist())
    assert v[1:].ravel().tolist() == pytest.approx(ref_v[1:].ravel().tolist(), abs=1e-6)


@pytest.mark.parametrize("name", ["1l2y", "7cal"])
def test_anm_eigenvalues_prody(sc, name):
    """ProDy ANM 13 A eigenvalues; 7cal is the n = 5328 case (fixture of tests/test_anm.py:145-334)."""
    ca = structures()[f"{name}_coord"]
    w, v = sc.ANM(ca, sc.InvariantForceField(13.0)).eigen()
    ref = load_csv(f"prody_anm_13_ang_cutoff_evals_{name}.csv.gz")
    check_eigenvalues(w, np.concatenate([np.zeros(6), ref[6:]]), 6, rtol=1e-5)
    assert v.shape == (len(w), len(w))


@pytest.mark.parametrize("ff_name", ["calpha", "pfanm"])
@pytest.mark.parametrize("name", ["1l2y", "7cal"])
def test_mass_weighted_eigenvalues_bio3d(sc, name, ff_name):
    # reference test: tests/test_anm.py:87-142 (Hinsen and pfENM legs; rtol 5e-3, atol 2e-3)
    s = structures()
    n = len(s[f"{name}_coord"])
    atoms = sc.AtomArray(n)
    atoms.coord = s[f"{name}_coord"]
    atoms.res_name = s[f"{name}_res_name"]
    ff = sc.HinsenForceField() if ff_name == "calpha" else sc.ParameterFreeForceField()
    masses = load_csv(f"bio3d_mass_{name}.csv.gz")
    w, _ = sc.ANM(atoms, ff, masses=masses).eigen()
    ref = load_csv(f"bio3d_anm_{ff_name}_ff_evals_mw_{name}.csv.gz")
    assert np.allclose(w[6:], ref[6:], rtol=5e-3, atol=2e-3)


# ---- benchmark configurations against the reference-generated eigenvalues --------------------------------

def test_config1_gnm(sc):
    g = generated("c1_1l2y_gnm7.npz")
    w, v = sc.GNM(structures()["1l2y_coord"], sc.InvariantForceField(7.0)).eigen()
    check_eigenvalues(w, g["eigenvalues"], 1)
    check_eigenvectors(g["kirchhoff"], w, v)


def test_config2_anm_n512(sc):
    g = generated("c2_n512_inv13.npz")
    coord = synthetic_coord(512, 0, 40.0)
    anm = sc.ANM(coord, sc.InvariantForceField(13.0))
    w, v = anm.eigen()                       # fused: coordinates -> eigenpairs on device
    check_eigenvalues(w, g["eigenvalues"], 6)
    h, _ = orc.compute_hessian(coord, orc.invariant_ff(13.0))
    check_eigenvectors(h, w, v)
    lo, hi = g["eigenvectors_sel_range"]
    overlap = np.abs(np.sum(v[lo:hi] * g["eigenvectors_sel"], axis=1))
    gaps = np.minimum(np.diff(w)[lo - 1:hi - 1], np.diff(w)[lo:hi]) / w[lo:hi]
    assert np.all(overlap[gaps > 1e-3] >= 1 - 1e-5)
    # second route: host matrix (user-assignable, anm.py:120-130) through sc_eigh_f64
    w2, v2 = sc.nma.eigh(anm.hessian)
    assert np.abs(w2 - w).max() <= 1e-10 * w.max()


def test_config3_anm_n2000_hinsen(sc):
    """The headline configuration: N=2000 C-alpha, Hinsen, no cutoff, all 6000 modes."""
    g = generated("c3_n2000_hinsen.npz")
    coord = synthetic_coord(2000, 0)
    w, v = sc.ANM(coord, sc.HinsenForceField()).eigen()
    check_eigenvalues(w, g["nocut_eigenvalues"], 6)
    # size-independent properties at full size: orthonormality and residual on sampled modes
    rs = np.random.RandomState(0)
    sel = np.sort(rs.choice(6000, 64, replace=False))
    gram = v[sel] @ v.T
    assert np.abs(gram - np.eye(6000)[sel]).max() <= 1e-8
    h, _ = orc.compute_hessian(coord, orc.hinsen_ff())
    r = h @ v[sel].T - v[sel].T * w[sel][None, :]
    assert np.linalg.norm(r, axis=0).max() <= 1e-5 * w.max()
    assert abs(w.sum() - np.trace(h)) <= 1e-9 * np.abs(w).sum()       # trace is preserved
    lo, hi = g["nocut_eigenvectors_sel_range"]
    overlap = np.abs(np.sum(v[lo:hi] * g["nocut_eigenvectors_sel"], axis=1))
    gaps = np.minimum(np.diff(w)[lo - 1:hi - 1], np.diff(w)[lo:hi]) / w[lo:hi]
    assert np.all(overlap[gaps > 1e-3] >= 1 - 1e-5)


def test_config4_anm_n1000_batch_members(sc):
    g = generated("c4_n1000_inv13.npz")
    for seed in range(2):
        w, _ = sc.ANM(synthetic_coord(1000, seed, 50.0), sc.InvariantForceField(13.0)).eigen()
        check_eigenvalues(w, g[f"s{seed}_eigenvalues"], 6)


def test_consumers_run_on_device_eigen(sc):
    """The NumPy consumers (nma.py:66-569) keep working on top of the device eigensolver."""
    ca = structures()["1l2y_coord"]
    anm = sc.ANM(ca, sc.InvariantForceField(13.0))
    h_ref, _ = orc.compute_hessian(ca, orc.invariant_ff(13.0))
    w_ref, v_ref = np.linalg.eigh(h_ref)
    f = anm.frequencies()
    assert np.allclose(f[6:], np.sqrt(w_ref[6:]) / (2 * np.pi))
    msf = anm.mean_square_fluctuation()
    cov = np.linalg.pinv(h_ref, hermitian=True, rcond=1e-6)
    assert np.allclose(msf, np.diag(cov).reshape(-1, 3).sum(1))
    d = anm.dcc()
    assert d.shape == (20, 20) and np.allclose(np.diag(d), 1.0)
    assert anm.normal_mode(6, 2.0, 10).shape == (10, 20, 3)
    assert np.allclose(anm.hessian @ anm.covariance @ anm.hessian, anm.hessian)  # tests/test_anm.py:26-37
    with pytest.raises(ValueError):
        sc.nma.eigen(object())                                                # nma.py:58


# ---- partial spectrum (K8; BASELINE config 5) ---------------------------------------------------------------

@pytest.mark.parametrize("n,lo,hi", [(50, 0, 9), (200, 0, 29), (333, 100, 140), (1000, 0, 105), (64, 60, 63), (5, 0, 4)])
def test_partial_spectrum_random(sc, n, lo, hi):
    a = sym(np.random.RandomState(n + lo), n)
    w, v = sc.nma.eigh(a, subset_by_index=(lo, hi))
    w_ref, v_ref = np.linalg.eigh(a)
    m = hi - lo + 1
    assert w.shape == (m,) and v.shape == (m, n)
    scale = np.abs(w_ref).max()
    assert np.abs(w - w_ref[lo:hi + 1]).max() <= 1e-11 * scale
    r = a @ v.T - v.T * w[None, :]
    assert np.abs(r).max() <= 1e-9 * scale
    assert np.abs(v @ v.T - np.eye(m)).max() <= 1e-10
    w_only = sc.nma.eigh(a, eigenvectors=False, subset_by_index=(lo, hi))
    assert np.abs(w_only - w_ref[lo:hi + 1]).max() <= 1e-11 * scale


def test_partial_spectrum_degenerate_cluster(sc):
    """ANM: the six rigid-body modes form a numerically degenerate cluster; compare the invariant subspace."""
    coord = synthetic_coord(300, 9)
    h, _ = orc.compute_hessian(coord, orc.invariant_ff(13.0))
    anm = sc.ANM(coord, sc.InvariantForceField(13.0))
    w, v = anm.eigen(subset_by_index=(0, 25))        # fused device path
    w_ref, v_ref = np.linalg.eigh(h)
    check_eigenvalues(np.concatenate([w, w_ref[26:]]), w_ref, 6)
    assert np.abs(v @ v.T - np.eye(26)).max() <= 1e-9
    p = v[:6].T @ v[:6]
    p_ref = v_ref[:, :6] @ v_ref[:, :6].T
    assert np.abs(p - p_ref).max() <= 1e-8                              # same null space
    r = h @ v.T - v.T * w[None, :]
    assert np.abs(r).max() <= 1e-9 * w_ref.max()
    overlap = np.abs(np.sum(v[6:] * v_ref[:, 6:26].T, axis=1))
    assert np.all(overlap >= 1 - 1e-6)
    # identical matrix through the host-matrix route
    w2, v2 = sc.nma.eigh(anm.hessian, subset_by_index=(0, 25))
    assert np.abs(w2[6:] - w[6:]).max() <= 1e-10 * w_ref.max()


def test_partial_spectrum_exactly_degenerate(sc):
    a = np.eye(120) * 3.0
    w, v = sc.nma.eigh(a, subset_by_index=(0, 19))
    assert np.allclose(w, 3.0) and np.abs(v @ v.T - np.eye(20)).max() <= 1e-10


def test_partial_spectrum_exact_zero_block(sc):
    """
    ADVICE round 5: a matrix with a decoupled block of exactly zero eigenvalues (a tridiagonal matrix with exact zeros
    on its diagonal and sub-diagonal there): the Sturm bisection stops at a bracket of eps |T| instead of driving its
    shifts into the range where the product-form count flushes rows; the zero eigenvalues come back as |w| <= eps |T|
    with orthonormal vectors, the others as LAPACK's.
    """
    n = 96
    rs = np.random.RandomState(4)
    d = rs.uniform(1.0, 3.0, n)
    e = rs.uniform(0.2, 0.5, n - 1)
    d[40:52] = 0.0                                   # twelve exactly zero rows ...
    e[39:52] = 0.0                                   # ... decoupled from both neighbours
    a = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    w_ref = np.linalg.eigvalsh(a)
    k0 = int(np.sum(w_ref < -1e-12))                 # eigenvalues below the zero cluster
    lo, hi = max(0, k0 - 3), min(n - 1, k0 + 12 + 3)
    w, v = sc.nma.eigh(a, subset_by_index=(lo, hi))
    scale = np.abs(w_ref).max()
    assert np.abs(w - w_ref[lo:hi + 1]).max() <= 1e-13 * scale
    assert np.sum(np.abs(w) <= 4e-16 * scale) == 12
    assert np.abs(v @ v.T - np.eye(hi - lo + 1)).max() <= 1e-10
    assert np.abs(a @ v.T - v.T * w[None, :]).max() <= 1e-12 * scale


def test_config5_anm_n8000_lowest_modes(sc):
    """Config 5: N = 8000 C-alpha (24000 x 24000 Hessian), InvariantForceField 13 A, modes 0..105."""
    import os
    from tests.util import GOLDEN

    path = os.path.join(GOLDEN, "generated", "c5_n8000_inv13.npz")
    if not os.path.exists(path):
        pytest.skip("c5 golden not generated")
    g = np.load(path)
    coord = synthetic_coord(8000, 0, 100.0)
    w, v = sc.ANM(coord, sc.InvariantForceField(13.0)).eigen(subset_by_index=(0, 105))
    ref = g["eigenvalues_low106"]
    lam_scale = 100.0   # ||H|| ~ 1e2 for this cutoff; trivial modes are noise at 1e-13 of it
    assert np.abs(w[:6]).max() <= 1e-9 * lam_scale
    rel = np.abs(w[6:] - ref[6:]) / np.abs(ref[6:])
    assert rel.max() <= 1e-5, rel.max()
    assert np.abs(v @ v.T - np.eye(106)).max() <= 1e-8


# ---- F1: covariance = Hermitian pseudo-inverse on device -----------------------------------------------------

@pytest.mark.parametrize("name", ["1l2y", "7cal"])
def test_covariance_moore_penrose(sc, name):
    # reference test: tests/test_anm.py:26-37 (both PDBs; 7cal is 5328 x 5328)
    ca = structures()[f"{name}_coord"]
    anm = sc.ANM(ca, sc.InvariantForceField(13.0))
    h, c = anm.hessian, anm.covariance
    assert np.allclose(h, h @ (c @ h))
    assert np.allclose(c, c @ (h @ c))
    if name == "1l2y":
        assert np.allclose(c, np.linalg.pinv(h, hermitian=True, rcond=1e-6), rtol=1e-8, atol=1e-10)
        anm2 = sc.ANM(ca, sc.InvariantForceField(13.0))
        anm2.covariance = c                                   # hessian is rebuilt by pinv (anm.py:114-117)
        assert np.allclose(anm2.hessian, h, atol=1e-8)


def test_pinvh_matches_numpy(sc):
    rs = np.random.RandomState(3)
    q, _ = np.linalg.qr(rs.randn(200, 200))
    lam = np.concatenate([np.zeros(5), rs.rand(195) + 0.5])
    a = (q * lam) @ q.T
    a = 0.5 * (a + a.T)
    assert np.allclose(sc.nma.pinvh(a), np.linalg.pinv(a, hermitian=True, rcond=1e-6), atol=1e-10)
    gnm = sc.GNM(synthetic_coord(60, 2, 15.0), sc.InvariantForceField(7.0))
    k = gnm.kirchhoff
    assert np.allclose(gnm.covariance, np.linalg.pinv(k, hermitian=True, rcond=1e-6), atol=1e-10)


@pytest.mark.parametrize("scale", [1e200, 1e-200, 3.7e160, 1e-170])
def test_badly_scaled_matrices(sc, scale):
    """
    np.linalg.eigh (LAPACK dsyevd) rescales matrices whose norm is near the over / underflow thresholds; the squared
    norms of the Householder steps would otherwise overflow / vanish.  Same here, on both tridiagonalisation paths.
    """
    from springcraft_amd import _hip

    n = 300
    a = sym(np.random.RandomState(9), n)
    w_ref, _ = np.linalg.eigh(a)
    ctx = _hip.context()
    try:
        for mode in (False, True):
            ctx.set_two_stage(mode)
            w, v = sc.nma.eigh(a * scale)
            assert np.all(np.isfinite(w)) and np.all(np.isfinite(v))
            assert np.abs(w / scale - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
            assert np.abs(a @ v.T - v.T * (w / scale)).max() <= 1e-11 * np.abs(w_ref).max()
            w_only = sc.nma.eigh(a * scale, eigenvectors=False)
            assert np.abs(w_only / scale - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
            ws, _ = sc.nma.eigh(a * scale, subset_by_index=(10, 20))
            assert np.abs(ws / scale - w_ref[10:21]).max() <= 1e-11 * np.abs(w_ref).max()
    finally:
        ctx.set_two_stage(None)


@pytest.mark.parametrize("two_stage", [False, True])
@pytest.mark.parametrize("n", [640, 1000])
def test_exactly_rank_deficient_input(two_stage, n):
    """
    ones(n, n) and relatives: the trailing matrices of the reduction are pure rounding residue that shrinks by a factor
    eps per column, down into the range where squares underflow (round 2: reflectors built from such columns were not
    orthogonal; n = 1000 on the one-stage path returned vectors of norm 0.76).  Gates of SURVEY 8(d).
    """
    import springcraft_amd as sc
    from springcraft_amd import _hip

    ctx = _hip.context()
    ctx.set_two_stage(two_stage)
    try:
        blocks = np.zeros((n, n))
        blocks[: n // 2, : n // 2] = 1.0
        blocks[n // 2:, n // 2:] = 3.0
        for a in (np.ones((n, n)), 1e-120 * np.ones((n, n)), 1e130 * np.ones((n, n)), blocks):
            w, v = sc.nma.eigh(a)
            w_ref = np.linalg.eigvalsh(a)
            scale = np.abs(w_ref).max()
            assert np.abs(w - w_ref).max() <= 1e-12 * scale
            assert np.abs(a @ v.T - v.T * w[None, :]).max() <= 1e-12 * scale
            assert np.abs(v @ v.T - np.eye(n)).max() <= 1e-12
    finally:
        ctx.set_two_stage(None)


@pytest.mark.parametrize("two_stage", [False, True])
def test_non_finite_input_is_rejected(two_stage):
    """NaN / Inf in the matrix: LinAlgError as from np.linalg.eigh (nma.py:61), before any solver kernel sees it."""
    import springcraft_amd as sc
    from springcraft_amd import _hip

    ctx = _hip.context()
    ctx.set_two_stage(two_stage)
    try:
        rs = np.random.RandomState(5)
        a = rs.standard_normal((300, 300))
        a = a + a.T
        for bad in (np.nan, np.inf, -np.inf):
            b = a.copy()
            b[200, 17] = b[17, 200] = bad
            with pytest.raises(np.linalg.LinAlgError):
                sc.nma.eigh(b)
            with pytest.raises(np.linalg.LinAlgError):
                sc.nma.eigh(b, eigenvectors=False)
            with pytest.raises(np.linalg.LinAlgError):
                sc.nma.eigh(b, subset_by_index=(0, 9))
        # the context is still usable and the upper triangle is not read (UPLO = 'L')
        c = a.copy()
        c[17, 200] = np.nan
        w, v = sc.nma.eigh(c)
        assert np.allclose(w, np.linalg.eigvalsh(a), rtol=0, atol=1e-11 * np.abs(w).max())
        # a NaN coordinate with a cut-off force field only isolates its atom (every comparison with NaN is false, as in
        # interaction.py:166): the Hessian stays finite, with three more zero modes
        coord = synthetic_coord(60, 1)
        coord[7, 1] = np.nan
        w = sc.ANM(coord, sc.InvariantForceField(13.0)).eigen()[0]
        assert np.isfinite(w).all() and np.abs(w[:9]).max() <= 1e-9 * w.max() and w[9] > 1e-6 * w.max()
        # without a cut-off the NaN reaches the Hessian
        with pytest.raises(np.linalg.LinAlgError):
            sc.ANM(coord, sc.HinsenForceField()).eigen()
    finally:
        ctx.set_two_stage(None)


def test_many_small_matrices_chunked_gemm_records(sc):
    """
    33 000 matrices of order 40 in one batched solve: the D&C merge launch then carries 66 000 GEMM records, more than
    grid.z holds -- the launcher must split the record list (it used to fail with SC_ERR_INVALID_ARG).
    """
    import ctypes as C

    import torch

    from springcraft_amd import _hip

    n, batch = 40, 33000
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn((batch, n, n), dtype=torch.float64, device="cuda", generator=g)
    a = a + a.transpose(1, 2)
    keep = a.clone()
    w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
    v = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
    ctx = _hip.Context(0)
    try:
        ctx.check(_hip.lib().sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, batch, C.c_void_p(w.data_ptr()),
                                             C.c_void_p(v.data_ptr())))
        ctx.synchronize()
    finally:
        ctx.close()
    r = torch.bmm(keep, v.transpose(1, 2)) - v.transpose(1, 2) * w[:, None, :]
    assert float(r.abs().max()) <= 1e-11 * float(w.abs().max())
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    assert float((torch.bmm(v, v.transpose(1, 2)) - eye).abs().max()) <= 1e-12
    for b in (0, batch // 2, batch - 1):
        w_ref = np.linalg.eigvalsh(keep[b].cpu().numpy())
        assert np.abs(w[b].cpu().numpy() - w_ref).max() <= 1e-12 * np.abs(w_ref).max()


@pytest.mark.parametrize("two_stage", [False, True])
def test_device_entry_points_do_not_synchronise_and_defer_errors(two_stage):
    """
    sc_dev_eigh_f64 only enqueues (its descriptor tables go through the context's pinned staging arena): a matrix with
    a NaN in a batch is solved as the zero matrix, its eigenvalues come back NaN, the other matrices of the batch are
    unaffected, and the error surfaces as LinAlgError (what np.linalg.eigh raises, nma.py:61) at the next
    sc_ctx_synchronize -- once.
    """
    import ctypes as C

    import torch

    from springcraft_amd import _hip

    n, batch = 700, 5
    rs = np.random.RandomState(9)
    mats = np.stack([(lambda a: a + a.T)(rs.standard_normal((n, n))) for _ in range(batch)])
    mats[3, 400, 20] = np.nan          # lower triangle of matrix 3
    L = _hip.lib()
    ctx = _hip.Context(0)
    try:
        ctx.set_two_stage(two_stage)
        # (the context has a stream of its own: inputs are prepared on torch's stream and synchronised first)
        inputs = [torch.from_numpy(mats.copy()).cuda() for _ in range(2)]
        w = torch.empty((batch, n), dtype=torch.float64, device="cuda")
        v = torch.empty((batch, n, n), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for a in inputs:               # two solves back to back without a synchronisation in between
            ctx.check(L.sc_dev_eigh_f64(ctx.handle, C.c_void_p(a.data_ptr()), n, batch, C.c_void_p(w.data_ptr()),
                                        C.c_void_p(v.data_ptr())))
        with pytest.raises(np.linalg.LinAlgError):
            ctx.synchronize()
        ctx.synchronize()              # the flag is cleared by the call that reported it
        wh = w.cpu().numpy()
        assert np.isnan(wh[3]).all()
        for b in (0, 1, 2, 4):
            w_ref = np.linalg.eigvalsh(mats[b])
            assert np.abs(wh[b] - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    finally:
        ctx.close()


def test_result_arrays_on_pooled_page_locked_memory(sc):
    """
    Large results come back on page-locked blocks that are pooled (springcraft_amd/_hip.py:host_array): the block of a
    result that is gone serves the next one of that size, a result (or a view of it) that is still alive keeps its block to
    itself, and the eigenpairs are what LAPACK gives.
    """
    import gc
    import os

    from springcraft_amd import _hip, nma

    n = 1201                                   # (a size no other test leaves blocks of in the pool)
    a = sym(np.random.RandomState(5), n)
    live0 = _hip._pin_live
    w, v = nma.eigh(a)
    # (page-locking may be refused by the box -- the arrays then live on ordinary memory and only the results are checked)
    pooled = os.environ.get("SPRINGCRAFT_PINNED_RESULTS", "1") != "0" and _hip._pin_live - live0 >= 8 * n * n
    assert isinstance(v, np.ndarray) and v.flags.writeable and v.flags.c_contiguous
    w_ref = np.linalg.eigvalsh(a)
    assert np.abs(w - w_ref).max() <= 1e-11 * np.abs(w_ref).max()
    first = v.ctypes.data
    keep = v[5].copy()
    w2, v2 = nma.eigh(a)                       # the first result is alive: another block
    second = v2.ctypes.data
    assert second != first and np.array_equal(v2[5], keep)
    del v, v2
    gc.collect()
    w3, v3 = nma.eigh(a)                       # both blocks are back in the pool: one of them is reused
    assert np.array_equal(v3[5], keep)
    if pooled:
        assert v3.ctypes.data in (first, second)
    view = v3[:10]
    third = v3.ctypes.data
    del v3
    gc.collect()
    w4, v4 = nma.eigh(a)                       # a view keeps its block: this result is on the other one
    w5, v5 = nma.eigh(a)                       # ... and this one on a new one
    assert np.array_equal(view[5], keep) and np.array_equal(v4[5], keep) and np.array_equal(v5[5], keep)
    assert v4.ctypes.data != third and v5.ctypes.data != third and v4.ctypes.data != v5.ctypes.data
    if pooled:
        assert v4.ctypes.data in (first, second)
