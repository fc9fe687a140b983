"""
F3 — TabulatedForceField (sENM / dENM / sdENM / eANM parameter sets) evaluated on device, against the
reference's third-party goldens (BioPhysConnectoR, Bio3D; reference tests tests/test_forcefield.py:360-422,
tests/test_anm.py:60-142) and against the host-callback path (force_constant() in Python).
"""
import numpy as np
import pytest

from tests.util import load_csv, structures

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import springcraft_amd

    return springcraft_amd


def atoms_of(sc, name):
    s = structures()
    n = len(s[f"{name}_coord"])
    a = sc.AtomArray(n)
    a.coord = s[f"{name}_coord"]
    a.res_name = s[f"{name}_res_name"]
    a.chain_id = s[f"{name}_chain_id"]
    a.res_id = s[f"{name}_res_id"]
    return a


def fused(ff):
    from springcraft_amd.forcefield import device_plan

    return device_plan(ff)[2]


@pytest.mark.parametrize("ff_name", ["e_anm", "e_anm_mj", "e_anm_ke"])
def test_eanm_hessians_biophysconnector(sc, ff_name):
    # reference test: tests/test_forcefield.py:360-389
    ca = atoms_of(sc, "1l2y")
    ff = getattr(sc.TabulatedForceField, ff_name)(ca)
    assert fused(ff)
    ref_file = {"e_anm": "biophysconnector_anm_eanm_hessian_1l2y.csv.gz",
                "e_anm_mj": "biophysconnector_anm_eanm_mj_hessian_1l2y.csv.gz",
                "e_anm_ke": "biophysconnector_anm_eanm_ke_hessian_1l2y.csv.gz"}[ff_name]
    ref = load_csv(ref_file, skip_header=1)
    h, pairs = sc.compute_hessian(ca.coord, ff)
    if ff_name == "e_anm_ke":
        assert np.allclose(h, ref, atol=1e-4)
    else:
        assert np.allclose(h, ref)
    # host-callback path (exposing the matrix disables the fused path) must agree to rounding
    ff2 = getattr(sc.TabulatedForceField, ff_name)(ca)
    _ = ff2.interaction_matrix
    assert not fused(ff2)
    h2, pairs2 = sc.compute_hessian(ca.coord, ff2)
    assert np.array_equal(pairs, pairs2)
    assert np.abs(h - h2).max() <= 1e-12 * np.abs(h2).max()


def test_sdenm_hessian_bio3d(sc):
    # reference test: tests/test_forcefield.py:392-422 (sdENM leg): distance-binned, type-specific tables
    ca = atoms_of(sc, "1l2y")
    ff = sc.TabulatedForceField.sd_enm(ca)
    assert fused(ff)
    h, _ = sc.compute_hessian(ca.coord, ff)
    assert np.allclose(h, load_csv("bio3d_anm_sdenm_ff_hessian_1l2y.csv.gz"))
    k, _ = sc.compute_kirchhoff(ca.coord, ff)
    ff2 = sc.TabulatedForceField.sd_enm(ca)
    _ = ff2.interaction_matrix
    k2, _ = sc.compute_kirchhoff(ca.coord, ff2)
    assert np.array_equal(k, k2)      # same float32 table entries, exact


@pytest.mark.parametrize("name", ["1l2y", "7cal"])
def test_eanm_eigenvalues_biophysconnector(sc, name):
    # reference test: tests/test_anm.py:60-84 (7cal: n = 5328, four chains -> inter-chain Keskin table)
    ca = atoms_of(sc, name)
    w, _ = sc.ANM(ca, sc.TabulatedForceField.e_anm(ca)).eigen()
    ref = load_csv(f"biophysconnector_anm_eanm_evals_{name}.csv.gz", skip_header=1)
    assert np.allclose(w[6:], ref[6:])


@pytest.mark.parametrize("name", ["1l2y", "7cal"])
def test_sdenm_mass_weighted_eigenvalues_bio3d(sc, name):
    # reference test: tests/test_anm.py:87-142, sdENM leg incl. the PatchedForceField used for multi-chain input
    ca = atoms_of(sc, name)
    ff = sc.TabulatedForceField.sd_enm(ca)
    if len(np.unique(ca.chain_id)) > 1:
        diff = np.diff(ca.res_id)
        after = np.where((diff > 1) | (diff < 0))[0] + 1          # biotite's check_res_id_continuity
        pairs = np.array([after - 1, after]).T
        bonded = 43.52 * 0.0083144621 * 300 * 10
        ff = sc.PatchedForceField(ff, contact_pair_off=pairs, contact_pair_on=pairs,
                                  force_constants=np.full(len(pairs), bonded))
    assert fused(ff)
    masses = load_csv(f"bio3d_mass_{name}.csv.gz")
    w, _ = sc.ANM(ca, ff, masses=masses).eigen()
    ref = load_csv(f"bio3d_anm_sdenm_ff_evals_mw_{name}.csv.gz")
    assert np.allclose(w[6:], ref[6:], rtol=5e-3, atol=2e-3)


def test_tabulated_generic_tables_match_callback(sc):
    """Random symmetric (20,20,k) tables, two chains, gaps in res_id: fused == callback, pair lists identical."""
    rs = np.random.RandomState(4)
    n = 150
    a = sc.AtomArray(n)
    a.coord = (rs.rand(n, 3) * 25).astype(np.float32)
    from springcraft_amd.forcefield import AA_LIST

    a.res_name = np.array(AA_LIST)[rs.randint(0, 20, n)]
    a.chain_id = np.where(np.arange(n) < 90, "A", "B")
    rid = np.arange(n) + 1
    rid[40:] += 3
    rid[90:] = np.arange(n - 90) + 1
    a.res_id = rid
    edges = np.array([4.0, 6.5, 9.0, 12.0])

    def table():
        t = rs.rand(20, 20, 4).astype(np.float32)
        return t + t.transpose(1, 0, 2)

    args = (table(), table(), table(), edges)
    ff = sc.TabulatedForceField(a, *args)
    ff_cb = sc.TabulatedForceField(a, *args)
    _ = ff_cb.interaction_matrix
    assert fused(ff) and not fused(ff_cb)
    k, p = sc.compute_kirchhoff(a.coord, ff)
    k2, p2 = sc.compute_kirchhoff(a.coord, ff_cb)
    assert np.array_equal(p, p2) and np.array_equal(k, k2)
    h, _ = sc.compute_hessian(a.coord, ff)
    h2, _ = sc.compute_hessian(a.coord, ff_cb)
    assert np.abs(h - h2).max() <= 1e-12 * np.abs(h2).max()
