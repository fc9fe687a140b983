"""
GPU parity tests for the assembly half of the hot path (contact scan, pair list, Kirchhoff,
Hessian): the HIP kernels, called through the Python mirror of the reference API (which goes
through the C ABI), against the oracle on the same inputs, against the reference's own golden
vectors, and against vectors generated from the imported reference.

Bars: contacts / pair lists / integer Kirchhoff matrices bit exact; Hessians within 1e-12
relative (Frobenius), the only deviation being the order of the diagonal-block reduction and
a <= 4 ulp difference in Hinsen's d**-6.
"""
import numpy as np
import pytest

from oracle import enm_oracle as orc
from tests.util import generated, load_csv, pair_digest, structures, synthetic_coord

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import springcraft_amd

    return springcraft_amd


def rel_fro(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


# ---- reference golden vectors (ProDy / Bio3D) ----------------------------------------------------

@pytest.mark.parametrize("cutoff", [5, 10, 15])
@pytest.mark.parametrize("use_cell_list", [False, True])
def test_kirchhoff_prody_random(sc, cutoff, use_cell_list):
    # reference test: tests/test_interaction.py:11-40
    coord = load_csv("random_coord_seed_1.csv.gz")
    k, pairs = sc.compute_kirchhoff(coord, sc.InvariantForceField(cutoff), use_cell_list)
    ref = load_csv(f"prody_gnm_{cutoff}_ang_cutoff_kirchhoff_random_coords_seed_1.csv.gz")
    assert np.array_equal(k, ref)
    assert pairs.dtype == np.int64 and pairs.shape[1] == 2


def test_hessian_prody_random(sc):
    # reference test: tests/test_interaction.py:43-68
    coord = load_csv("random_coord_seed_1.csv.gz")
    h, _ = sc.compute_hessian(coord, sc.InvariantForceField(10))
    ref = load_csv("prody_anm_10_ang_cutoff_hessian_random_coords_seed_1.csv.gz")
    assert np.allclose(h, ref, atol=1e-6, rtol=1e-3)  # the reference's tolerance
    assert np.abs(h - ref).max() < 1e-12              # ours


@pytest.mark.parametrize("cutoff", [4, 7, 13])
def test_kirchhoff_prody_1l2y(sc, cutoff):
    # reference test: tests/test_gnm.py:23-44; float32 coordinates as an AtomArray delivers them
    ca = structures()["1l2y_coord"]
    assert ca.dtype == np.float32
    gnm = sc.GNM(ca, sc.InvariantForceField(cutoff))
    ref = load_csv(f"prody_gnm_{cutoff}_ang_cutoff_kirchhoff_1l2y.csv.gz")
    assert np.array_equal(gnm.kirchhoff, ref)


def test_readme_example_c1(sc):
    # config 1: README example, GNM Kirchhoff of 1l2y with InvariantForceField(7.0)
    g = generated("c1_1l2y_gnm7.npz")
    ca = structures()["1l2y_coord"]
    k, pairs = sc.compute_kirchhoff(ca, sc.InvariantForceField(7.0))
    assert np.array_equal(k, g["kirchhoff"])
    assert np.array_equal(pairs, g["pairs"])


def test_hinsen_and_pf_bio3d_1l2y(sc):
    # reference test: tests/test_forcefield.py:392-422
    ca = structures()["1l2y_coord"]
    h, _ = sc.compute_hessian(ca, sc.HinsenForceField())
    assert np.allclose(h, load_csv("bio3d_anm_calpha_ff_hessian_1l2y.csv.gz"), atol=1e-4)
    h, _ = sc.compute_hessian(ca, sc.ParameterFreeForceField())
    assert np.allclose(h, load_csv("bio3d_anm_pfanm_ff_hessian_1l2y.csv.gz"))


# ---- oracle on the benchmark configurations -------------------------------------------------------

CASES = [
    ("c2", 512, 0, 40.0, "inv13"),
    ("c4", 1000, 0, 50.0, "inv13"),
    ("c3cut", 2000, 0, None, "hinsen13"),
    ("c3", 2000, 0, None, "hinsen"),
    ("odd", 777, 5, None, "pf9"),
]


def make_ff(sc_or_orc, tag):
    is_orc = sc_or_orc is orc
    if tag == "inv13":
        return orc.invariant_ff(13.0) if is_orc else sc_or_orc.InvariantForceField(13.0)
    if tag == "hinsen13":
        return orc.hinsen_ff(13.0) if is_orc else sc_or_orc.HinsenForceField(13.0)
    if tag == "hinsen":
        return orc.hinsen_ff() if is_orc else sc_or_orc.HinsenForceField()
    if tag == "pf9":
        return orc.parameter_free_ff(9.0) if is_orc else sc_or_orc.ParameterFreeForceField(9.0)
    raise KeyError(tag)


@pytest.mark.parametrize("name,n,seed,box,fftag", CASES)
def test_against_oracle(sc, name, n, seed, box, fftag):
    coord = synthetic_coord(n, seed, box)
    k_ref, pairs_ref = orc.compute_kirchhoff(coord, make_ff(orc, fftag))
    h_ref, _ = orc.compute_hessian(coord, make_ff(orc, fftag))
    k, pairs = sc.compute_kirchhoff(coord, make_ff(sc, fftag))
    h, pairs_h = sc.compute_hessian(coord, make_ff(sc, fftag))
    assert np.array_equal(pairs, pairs_ref)          # bit exact, np.where order
    assert np.array_equal(pairs_h, pairs_ref)
    assert np.array_equal(k != 0, k_ref != 0)        # same contacts
    if fftag == "inv13":
        assert np.array_equal(k, k_ref)              # integer valued: bit exact
    else:
        assert rel_fro(k, k_ref) < 1e-14
    assert rel_fro(h, h_ref) < 1e-12
    assert np.allclose(h, h.T)                       # reference test: tests/test_interaction.py:71-89
    assert np.abs(h - h.T).max() <= 4e-16 * np.abs(h).max()
    assert np.abs(h - h_ref).max() <= 1e-12 * np.abs(h_ref).max()


def test_generated_digests_c3(sc):
    """Config 3 (N=2000 Hinsen, 3 998 000 directed pairs) against the reference-generated digests."""
    g = generated("c3_n2000_hinsen.npz")
    coord = synthetic_coord(2000, 0)
    h, pairs = sc.compute_hessian(coord, sc.HinsenForceField())
    assert len(pairs) == int(g["nocut_n_pairs"]) == 2000 * 1999
    assert pair_digest(pairs) == str(g["nocut_pairs_sha256"])
    assert np.isclose(np.linalg.norm(h), g["nocut_hess_fro"], rtol=1e-13)
    h4 = h.reshape(2000, 3, 2000, 3)
    idx = np.arange(2000)
    d = h4[idx, :, idx, :]
    assert np.abs(d - g["nocut_hess_diag_blocks"]).max() <= 1e-12 * np.abs(d).max()
    si, sj = g["nocut_hess_sample_i"], g["nocut_hess_sample_j"]
    assert np.allclose(h4[si, :, sj, :], g["nocut_hess_sample_blocks"], rtol=1e-14, atol=0)


# ---- patches --------------------------------------------------------------------------------------

@pytest.mark.parametrize("base", ["inv6", "hinsen8", "hinsen_nocut"])
@pytest.mark.parametrize("ptag", ["shutdown", "off", "on", "all"])
def test_patched_fused(sc, base, ptag):
    """PatchedForceField around a built-in force field: evaluated on device (fused path)."""
    g = generated("patched_n40.npz")
    kw = {}
    if ptag in ("shutdown", "all"):
        kw["contact_shutdown"] = g["shutdown"]
    if ptag in ("off", "all"):
        kw["contact_pair_off"] = g["pair_off"]
    if ptag in ("on", "all"):
        kw["contact_pair_on"] = g["pair_on"]
        kw["force_constants"] = g["force_constants"]
    base_ff = {"inv6": sc.InvariantForceField(6.0), "hinsen8": sc.HinsenForceField(8.0),
               "hinsen_nocut": sc.HinsenForceField()}[base]
    ff = sc.PatchedForceField(base_ff, **kw)
    from springcraft_amd.forcefield import device_plan

    assert device_plan(ff)[2] is True
    k, pairs = sc.compute_kirchhoff(g["coord"], ff)
    h, _ = sc.compute_hessian(g["coord"], ff)
    assert np.array_equal(pairs, g[f"{base}_{ptag}_pairs"])
    k_ref, h_ref = g[f"{base}_{ptag}_kirchhoff"], g[f"{base}_{ptag}_hessian"]
    assert rel_fro(k, k_ref) < 1e-14
    assert rel_fro(h, h_ref) < 1e-13
    if base == "inv6" and ptag != "on" and ptag != "all":
        assert np.array_equal(k, k_ref)


def test_patched_callback_path(sc):
    """A subclass of PatchedForceField is not fused: pairs on device, gamma from Python."""
    g = generated("patched_n40.npz")

    class MyPatched(sc.PatchedForceField):
        pass

    ff = MyPatched(sc.HinsenForceField(8.0), contact_shutdown=g["shutdown"],
                   contact_pair_off=g["pair_off"], contact_pair_on=g["pair_on"],
                   force_constants=g["force_constants"])
    from springcraft_amd.forcefield import device_plan

    assert device_plan(ff)[2] is False
    k, pairs = sc.compute_kirchhoff(g["coord"], ff)
    h, _ = sc.compute_hessian(g["coord"], ff)
    assert np.array_equal(pairs, g["hinsen8_all_pairs"])
    assert np.array_equal(k, g["hinsen8_all_kirchhoff"])     # same arithmetic as the reference: bit exact
    assert np.array_equal(h, g["hinsen8_all_hessian"])


def test_cartesian_index_product(sc):
    # reference test: tests/test_interaction.py:92-116 — custom FF without cutoff -> all i != j
    class CustomFF(sc.ForceField):
        def force_constant(self, atom_i, atom_j, sq_distance):
            return np.ones(len(atom_i))

    n = 60
    coord = synthetic_coord(n, 3)
    _, pairs = sc.compute_hessian(coord, CustomFF())
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ref = np.stack([ii.ravel(), jj.ravel()], axis=1)
    ref = ref[ref[:, 0] != ref[:, 1]]
    assert np.array_equal(pairs, ref)


def test_asymmetric_custom_force_constants(sc):
    """gamma(i,j) != gamma(j,i): diagonal = minus the sum over the FIRST index (interaction.py:52,104)."""
    class Asym(sc.ForceField):
        def force_constant(self, atom_i, atom_j, sq_distance):
            return 1.0 + 0.01 * atom_i + 0.0001 * atom_j

        @property
        def cutoff_distance(self):
            return 9.0

    coord = synthetic_coord(90, 11)
    ff_o = orc.OracleFF(lambda i, j, d2: 1.0 + 0.01 * i + 0.0001 * j, 9.0)
    k, _ = sc.compute_kirchhoff(coord, Asym())
    h, _ = sc.compute_hessian(coord, Asym())
    k_ref, _ = orc.compute_kirchhoff(coord, ff_o)
    h_ref, _ = orc.compute_hessian(coord, ff_o)
    assert np.array_equal(k, k_ref)
    assert np.array_equal(h, h_ref)


def test_parameterfree_forcefield(sc):
    # reference test: tests/test_forcefield.py:337-357
    rs = np.random.RandomState(0)
    coord = rs.rand(5, 3)
    d = coord[:, None, :] - coord[None, :, :]
    with np.errstate(divide="ignore"):
        ref = -1 / (d * d).sum(-1)
    k, _ = sc.compute_kirchhoff(coord, sc.ParameterFreeForceField())
    np.fill_diagonal(ref, 0)
    np.fill_diagonal(k, 0)
    assert np.allclose(k, ref)


# ---- mass weighting, edge cases, errors -------------------------------------------------------------

def test_mass_weights(sc):
    # reference tests: tests/test_anm.py:40-57, tests/test_gnm.py:87-104
    s = structures()
    ca = sc.AtomArray(20)
    ca.coord = s["1l2y_coord"]
    ca.res_name = s["1l2y_res_name"]
    ff = sc.InvariantForceField(7.9)
    ref_anm = sc.ANM(ca, ff)
    same = sc.ANM(ca, ff, masses=np.ones(20))
    diff = sc.ANM(ca, ff, masses=np.arange(1, 21, dtype=float))
    assert np.allclose(same.hessian, ref_anm.hessian)
    assert not np.allclose(diff.hessian, ref_anm.hessian)
    m = np.arange(1, 21, dtype=float)
    h_ref, _ = orc.compute_hessian(ca.coord, orc.invariant_ff(7.9))
    assert np.allclose(diff.hessian, h_ref * orc.mass_weight_matrix(m, 3), rtol=1e-14, atol=0)
    g = sc.GNM(ca, ff, masses=m)
    k_ref, _ = orc.compute_kirchhoff(ca.coord, orc.invariant_ff(7.9))
    assert np.array_equal(g.kirchhoff, k_ref * orc.mass_weight_matrix(m, 1))
    mt = sc.ANM(ca, sc.HinsenForceField(), masses=True)
    assert mt.masses.shape == (20,) and (mt.masses > 50).all()


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 513])
def test_ragged_sizes(sc, n):
    coord = synthetic_coord(n, 42, 12.0)
    k, pairs = sc.compute_kirchhoff(coord, sc.InvariantForceField(6.0))
    h, _ = sc.compute_hessian(coord, sc.HinsenForceField(6.0))
    k_ref, pairs_ref = orc.compute_kirchhoff(coord, orc.invariant_ff(6.0))
    h_ref, _ = orc.compute_hessian(coord, orc.hinsen_ff(6.0))
    assert np.array_equal(k, k_ref)
    assert np.array_equal(pairs, pairs_ref.reshape(-1, 2))
    assert np.abs(h - h_ref).max() <= 1e-12 * max(np.abs(h_ref).max(), 1.0)


def test_empty_and_isolated(sc):
    k, pairs = sc.compute_kirchhoff(np.zeros((0, 3)), sc.InvariantForceField(5.0))
    assert k.shape == (0, 0) and pairs.shape == (0, 2)
    coord = np.array([[0.0, 0, 0], [100.0, 0, 0], [0, 100.0, 0]])
    k, pairs = sc.compute_kirchhoff(coord, sc.InvariantForceField(5.0))
    assert not k.any() and len(pairs) == 0
    h, _ = sc.compute_hessian(coord, sc.InvariantForceField(5.0))
    assert not h.any()


def test_inclusive_cutoff(sc):
    """d^2 == cutoff^2 exactly is a contact (interaction.py:166)."""
    coord = np.array([[0.0, 0, 0], [3.0, 4.0, 0.0], [0.0, 0.0, 5.000000000000001]])
    k, pairs = sc.compute_kirchhoff(coord, sc.InvariantForceField(5.0))
    assert np.array_equal(pairs, [[0, 1], [1, 0]])


def test_errors(sc):
    ff = sc.InvariantForceField(5.0)
    with pytest.raises(ValueError):
        sc.compute_kirchhoff(np.zeros((4, 2)), ff)                       # interaction.py:141-142
    with pytest.raises(ValueError):
        sc.compute_hessian(np.zeros(12), ff)
    with pytest.raises(ValueError):
        sc.InvariantForceField(None)                                      # forcefield.py:277-281
    coord = synthetic_coord(10, 1, 8.0)
    with pytest.raises(ValueError):                                       # interaction.py:210-211
        sc.compute_kirchhoff(coord, sc.PatchedForceField(ff, contact_pair_on=[[2, 2]], force_constants=[1.0]))
    with pytest.raises(IndexError):
        sc.compute_kirchhoff(coord, sc.PatchedForceField(ff, contact_shutdown=[10]))
    with pytest.raises(TypeError):                                        # forcefield.py:170-174
        sc.PatchedForceField(ff, contact_pair_on=[[1, 2]])
    with pytest.raises(TypeError):                                        # anm.py:70-73
        sc.ANM(coord, ff, masses=True)
    anm = sc.ANM(coord, ff)
    with pytest.raises(IndexError):                                       # anm.py:122-127
        anm.hessian = np.zeros((3, 3))
    gnm = sc.GNM(coord, ff)
    with pytest.raises(ValueError):                                       # gnm.py:115-120
        gnm.kirchhoff = np.zeros((3, 3))
    with pytest.raises(IndexError):
        gnm.covariance = np.zeros((3, 3))
