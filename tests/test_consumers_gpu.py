"""
GPU parity tests for the mode-subset consumers that run on device-resident eigenpairs
(``csrc/consumers.hip`` through ``sc_modes_*``): frequencies, mean-square fluctuations, B-factors, dynamic
cross-correlations and perturbation response scanning.

The golden vectors are the reference's own fixtures (ProDy / Bio3D results, tests/data of the reference, compared
the way tests/test_anm.py:145-334, :337-358 and tests/test_gnm.py:107-152 do) plus plain-NumPy evaluations of the
reference formulas (nma.py:108-184, :233-359, :476-524) on LAPACK eigenpairs for synthetic structures.
"""
import numpy as np
import pytest

from oracle import enm_oracle as orc
from tests.util import load_csv, structures, synthetic_coord

pytestmark = pytest.mark.gpu

K_B = 1.380649e-23
N_A = 6.02214076e23


@pytest.fixture(scope="module")
def sc():
    import springcraft_amd

    return springcraft_amd


@pytest.fixture(scope="module")
def ca():
    return structures()["1l2y_coord"]


# ---- NumPy restatement of the reference formulas (checker only) --------------------------------------------------
def np_msf(w, v, sel, dim):
    c = (v[sel] ** 2 / w[sel, None]).sum(axis=0)
    return c.reshape(-1, dim).sum(axis=1)


def np_dcc(w, v, sel, dim, norm):
    vs = v[sel]
    c = (vs.T / w[sel]) @ vs
    n = c.shape[0] // dim
    c = c.reshape(n, dim, n, dim).trace(axis1=1, axis2=3)
    if norm:
        d = np.sqrt(np.diag(c))
        c = c / np.outer(d, d)
    return c


def np_prs(h, norm):
    c2 = np.linalg.pinv(h, hermitian=True, rcond=1e-6) ** 2
    idx = np.arange(0, len(h), 3)
    m = np.add.reduceat(np.add.reduceat(c2, idx, axis=0), idx, axis=1)
    if norm:
        m = m / np.diag(m)[:, None]
    return m


# ---- the reference's own fixtures ----------------------------------------------------------------------------------
def test_anm_prody_frequency_fluctuation_dcc(sc, ca):
    """tests/test_anm.py:160-209 (ProDy, ANM 13 A on 1l2y)."""
    anm = sc.ANM(ca, sc.InvariantForceField(13))
    name = "prody_anm_13_ang_cutoff"
    evals = load_csv(f"{name}_evals_1l2y.csv.gz")
    assert np.allclose(anm.frequencies()[6:], np.sqrt(evals[6:]) / (2 * np.pi))
    assert np.allclose(anm.mean_square_fluctuation(tem=None), load_csv(f"{name}_fluctuations_1l2y.csv.gz"))
    assert np.allclose(anm.dcc(), load_csv(f"{name}_dcc_norm_1l2y.csv.gz"))
    assert np.allclose(anm.dcc(norm=False), load_csv(f"{name}_dcc_absolute_1l2y.csv.gz"))
    assert np.allclose(anm.dcc(mode_subset=np.arange(6, 36)), load_csv(f"{name}_dcc_norm_subset_1l2y.csv.gz"))


def test_anm_bio3d_hinsen_mass_weighted(sc, ca):
    """tests/test_anm.py:231-331, Hinsen branch (Bio3D; masses, T = 300 K)."""
    tem, scale = 300, K_B * N_A
    masses = load_csv("bio3d_mass_1l2y.csv.gz")
    ff = sc.HinsenForceField()

    class Atoms:  # a mass array needs an atom container (anm.py:81)
        coord = ca

        @staticmethod
        def array_length():
            return len(ca)

    nomw = sc.ANM(ca, ff)
    fluc_nomw = nomw.mean_square_fluctuation(tem=tem, tem_factors=scale)
    test = sc.ANM(Atoms, ff, masses=masses)
    pre = "bio3d_anm_calpha_ff"
    tol = dict(rtol=5e-03, atol=2e-03)
    assert np.allclose(test.frequencies()[6:], load_csv(f"{pre}_frequencies_mw_1l2y.csv.gz")[6:], **tol)
    fluc = test.mean_square_fluctuation(tem=tem, tem_factors=scale) / (1000 * masses)
    assert np.allclose(fluc, load_csv(f"{pre}_fluctuations_non_mw_1l2y.csv.gz"), **tol)
    sub = test.mean_square_fluctuation(tem=tem, tem_factors=scale, mode_subset=np.arange(11, 33)) / (1000 * masses)
    assert np.allclose(sub, load_csv(f"{pre}_fluctuations_subset_mw_1l2y.csv.gz"), **tol)
    assert np.allclose(test.dcc(), load_csv(f"{pre}_dcc_mw_1l2y.csv.gz"), **tol)
    assert np.allclose(test.dcc(mode_subset=np.arange(6, 36)), load_csv(f"{pre}_dcc_subset_mw_1l2y.csv.gz"), **tol)
    # alternative MSF straight from the covariance diagonal (tests/test_anm.py:308-334)
    alt = nomw.covariance.diagonal().reshape(len(ca), -1).sum(axis=1) * scale * tem
    assert np.allclose(fluc_nomw, alt)


@pytest.mark.parametrize("cutoff", [4, 7, 13])
def test_gnm_prody_fluctuation_dcc(sc, ca, cutoff):
    """
    tests/test_gnm.py:107-152 (ProDy).  The reference holds 13 A files too but only runs cutoffs 4 and 7; its 13 A
    "subset" file does not match the reference's own formula for any mode range (checked with NumPy), so it is
    left out here as well.
    """
    gnm = sc.GNM(ca, sc.InvariantForceField(cutoff))
    name = f"prody_gnm_{cutoff}_ang_cutoff"
    assert np.allclose(gnm.mean_square_fluctuation(), load_csv(f"{name}_fluctuations_1l2y.csv.gz"))
    assert np.allclose(gnm.dcc(), load_csv(f"{name}_dcc_norm_1l2y.csv.gz"))
    if cutoff != 13:
        assert np.allclose(gnm.dcc(mode_subset=np.arange(1, 17)), load_csv(f"{name}_dcc_norm_subset_1l2y.csv.gz"))
    assert np.allclose(gnm.dcc(norm=False), load_csv(f"{name}_dcc_absolute_1l2y.csv.gz"))


def test_prs_prody_1l2y(sc, ca):
    """tests/test_anm.py:337-358."""
    anm = sc.ANM(ca, sc.InvariantForceField(13))
    prs, eff, sens = anm.prs_effector_sensor()
    assert np.allclose(prs, load_csv("prody_anm_13_ang_cutoff_prs_mat_1l2y.csv.gz"))
    assert np.allclose(eff, load_csv("prody_anm_13_ang_cutoff_prs_eff_1l2y.csv.gz"))
    assert np.allclose(sens, load_csv("prody_anm_13_ang_cutoff_prs_sens_1l2y.csv.gz"))


# ---- against the formulas on LAPACK eigenpairs, larger and ragged sizes ---------------------------------------------
@pytest.mark.parametrize("n_atoms,seed", [(50, 0), (171, 1), (400, 2)])
def test_anm_consumers_match_numpy(sc, n_atoms, seed):
    coord = synthetic_coord(n_atoms, seed)
    h, _ = orc.compute_hessian(coord, orc.invariant_ff(13.0))
    w, vt = np.linalg.eigh(h)
    v = vt.T
    anm = sc.ANM(coord, sc.InvariantForceField(13.0))
    full = np.arange(6, 3 * n_atoms)
    sub = np.array([6, 7, 9, 3 * n_atoms - 1, 40, 41])
    for sel, arg in ((full, None), (sub, sub)):
        ref = np_msf(w, v, sel, 3)
        assert np.abs(anm.mean_square_fluctuation(mode_subset=arg) - ref).max() <= 1e-9 * np.abs(ref).max()
        for norm in (True, False):
            ref = np_dcc(w, v, sel, 3, norm)
            got = anm.dcc(mode_subset=arg, norm=norm)
            assert got.shape == (n_atoms, n_atoms)
            assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
    assert np.allclose(anm.bfactor(), 8 * np.pi**2 / 3 * np_msf(w, v, full, 3))
    tem = anm.dcc(tem=300, tem_factors=K_B * N_A)
    assert np.allclose(tem, np_dcc(w, v, full, 3, True) * 300 * K_B * N_A)
    for norm in (True, False):
        ref = np_prs(h, norm)
        got = sc.nma.prs(anm, norm=norm)
        assert np.abs(got - ref).max() <= 1e-8 * np.abs(ref).max()


@pytest.mark.parametrize("n_atoms", [33, 300])
def test_gnm_consumers_match_numpy(sc, n_atoms):
    coord = synthetic_coord(n_atoms, 5)
    k, _ = orc.compute_kirchhoff(coord, orc.invariant_ff(10.0))
    w, vt = np.linalg.eigh(k)
    v = vt.T
    gnm = sc.GNM(coord, sc.InvariantForceField(10.0))
    full = np.arange(1, n_atoms)
    ref = np_msf(w, v, full, 1)
    assert np.abs(gnm.mean_square_fluctuation() - ref).max() <= 1e-9 * np.abs(ref).max()
    ref = np_dcc(w, v, full, 1, True)
    assert np.abs(gnm.dcc() - ref).max() <= 1e-9
    assert np.allclose(gnm.frequencies()[1:], np.sqrt(w[1:]) / (2 * np.pi))


# ---- interface behaviour ------------------------------------------------------------------------------------------------
def test_trivial_modes_rejected_and_bad_index(sc, ca):
    anm = sc.ANM(ca, sc.InvariantForceField(13))
    with pytest.raises(ValueError):
        anm.mean_square_fluctuation(mode_subset=np.arange(5, 20))       # nma.py:155-160
    with pytest.raises(ValueError):
        anm.dcc(mode_subset=[0, 7])
    with pytest.raises(IndexError):
        anm.dcc(mode_subset=[7, 60])                                   # 60 modes: 0..59
    with pytest.raises(ValueError):
        sc.nma.prs(sc.GNM(ca, sc.InvariantForceField(7)))               # nma.py:507-508


def test_eigenpairs_are_shared_until_the_matrix_is_handed_out(sc, ca):
    anm = sc.ANM(ca, sc.InvariantForceField(13))
    from springcraft_amd._model import _modes_cache

    w, _ = anm.eigen()
    modes = _modes_cache.get(anm)
    assert modes is not None
    anm.mean_square_fluctuation()
    anm.dcc()
    assert _modes_cache.get(anm) is modes            # one solve served all three
    h = anm.hessian                                  # "not a copy": the caller may now edit it in place
    assert _modes_cache.get(anm) is None
    h *= 2.0
    w2, _ = anm.eigen()
    assert np.allclose(w2[6:], 2.0 * w[6:])
    assert np.allclose(anm.mean_square_fluctuation(), 0.5 * sc.ANM(ca, sc.InvariantForceField(13)).mean_square_fluctuation())
    # a user-assigned covariance is what prs reduces (anm.py:138-148)
    anm2 = sc.ANM(ca, sc.InvariantForceField(13))
    cov = np.linalg.pinv(np.array(sc.ANM(ca, sc.InvariantForceField(13)).hessian), hermitian=True, rcond=1e-6)
    anm2.covariance = 3.0 * cov
    assert np.allclose(sc.nma.prs(anm2, norm=False), 9.0 * np_prs(np.linalg.pinv(cov, hermitian=True), False),
                       rtol=1e-6)


def test_device_eigenpair_cache_is_bounded(sc):
    """
    Models that stay alive must not pin one (n, n) eigenvector matrix each in HBM (advisor finding, round 1): the
    device-resident eigenpairs live in a byte-bounded LRU; evicted models solve again and give the same numbers.
    """
    from springcraft_amd import _model

    cache = _model._modes_cache
    old = cache.budget
    cache.clear()
    try:
        n_atoms = 60
        one = 8 * ((3 * n_atoms) ** 2 + 3 * n_atoms)
        cache.budget = 3 * one + 100                     # room for three models
        models = [sc.ANM(synthetic_coord(n_atoms, s, 15.0), sc.InvariantForceField(9.0)) for s in range(8)]
        first = [m.eigen()[0] for m in models]
        assert cache.nbytes() <= cache.budget and len(cache._entries) == 3
        msf = [m.mean_square_fluctuation() for m in models]      # evicted ones solve again
        assert cache.nbytes() <= cache.budget
        for m, w, f in zip(models, first, msf):
            w2, v2 = m.eigen()
            assert np.array_equal(w, w2)
            ref = (v2[6:] ** 2 / w2[6:, None]).sum(0).reshape(-1, 3).sum(1)
            assert np.allclose(f, ref)
        models[-1].release_device_cache()
        assert cache.get(models[-1]) is None
        cache.clear()
        cache.budget = one - 1                            # nothing fits: every call solves, nothing is kept
        w3, _ = models[0].eigen()
        assert np.array_equal(w3, first[0]) and cache.nbytes() <= cache.budget
    finally:
        cache.budget = old
        cache.clear()


def test_dcc_all_modes_follows_the_pinv_rule_on_a_nearly_disconnected_network(sc):
    """
    dcc() with all modes is the covariance matrix in the reference (nma.py:324-336), i.e. pinv(M, hermitian=True,
    rcond=1e-6): on two clusters joined by one very weak spring a NON-trivial mode falls below 1e-6 * lambda_max and
    is dropped too.  Checked against NumPy's pinv of the oracle's Kirchhoff / Hessian.
    """
    rs = np.random.RandomState(11)
    a = rs.rand(12, 3) * 6.0
    b = rs.rand(12, 3) * 6.0 + np.array([40.0, 0.0, 0.0])
    coord = np.concatenate([a, b])
    weak = sc.PatchedForceField(sc.InvariantForceField(9.0), contact_pair_on=np.array([[0, 12]]),
                                force_constants=np.array([1e-9]))
    # GNM: one inter-cluster mode of ~1e-10 next to lambda_max ~ 10
    gnm = sc.GNM(coord, weak)
    k = gnm.kirchhoff.copy()
    w = np.linalg.eigvalsh(k)
    assert np.sum(np.abs(w) <= 1e-6 * np.abs(w).max()) == 2          # the trivial mode and the weak one
    cov = np.linalg.pinv(k, hermitian=True, rcond=1e-6)
    d = np.sqrt(np.diag(cov))
    assert np.allclose(sc.GNM(coord, weak).dcc(norm=False), cov, rtol=1e-8, atol=1e-10)
    assert np.allclose(sc.GNM(coord, weak).dcc(), cov / np.outer(d, d), rtol=1e-8, atol=1e-10)
    # ANM: trace of the 3x3 super-elements of pinv(H)
    anm = sc.ANM(coord, weak)
    h = anm.hessian.copy()
    covh = np.linalg.pinv(h, hermitian=True, rcond=1e-6)
    n = len(coord)
    tr = covh.reshape(n, 3, n, 3).swapaxes(1, 2).trace(axis1=2, axis2=3)
    assert np.allclose(sc.ANM(coord, weak).dcc(norm=False), tr, rtol=1e-8, atol=1e-9)
