"""
Ragged and decorated batches (``sc_batch_plan_*`` / ``RaggedBatchSolver``): structures of DIFFERENT sizes, with
per-structure force fields (built-in, TabulatedForceField, PatchedForceField) and masses, in ONE batched eigensolve.
The reference models one arbitrary structure per object (anm.py:62-63; forcefield.py:117-261, :369-533; anm.py:89-94);
every structure's result is compared with the oracle / the reference's third-party goldens exactly as the
single-structure tests do.
"""
import numpy as np
import pytest

from oracle import enm_oracle as orc
from tests.test_tabulated_gpu import atoms_of
from tests.util import check_eigenvalues, load_csv, synthetic_coord

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    import springcraft_amd

    return springcraft_amd


def _packed(coords):
    import torch

    return torch.from_numpy(np.concatenate(coords).astype(np.float64)).cuda().contiguous()


def test_sixteen_sizes_one_two_stage_solve(sc):
    """
    16 structures of 16 different N in ONE batched solve on the two-stage path: eigenvalues
    against LAPACK on the oracle's Hessian, eigenvectors by residual / orthonormality against that Hessian, and the
    padding stays out of the way (pad components of the structure's modes are zero, pad eigenvalues sort last).
    """
    import torch

    from springcraft_amd.batch import RaggedBatchSolver

    sizes = [371 + 2 * k for k in range(16)]                  # 3 N = 1113 .. 1203
    coords = [synthetic_coord(n, 100 + k) for k, n in enumerate(sizes)]
    ffs = [sc.InvariantForceField(13.0) if k % 2 == 0 else sc.HinsenForceField(13.0) for k in range(16)]
    s = RaggedBatchSolver(sizes, ffs, dim=3)
    assert s.order == 3 * max(sizes)
    s.ctx.set_two_stage(True)
    s.set_profiling(True)
    w, v = s.solve(_packed(coords))
    torch.cuda.synchronize()
    t = s.last_timings()
    assert t["two_stage"]
    assert s.ctx.counter("chase_launches") + s.ctx.counter("stepwise_chases") == 1     # ONE solve for all 16
    assert s.ctx.counter("chase_timeouts") == 0 and s.ctx.counter("chase_resumed") == 0
    for k, (wk, vk) in enumerate(s.results()):
        m = 3 * sizes[k]
        ff_o = orc.invariant_ff(13.0) if k % 2 == 0 else orc.hinsen_ff(13.0)
        h, _ = orc.compute_hessian(coords[k], ff_o)
        w_ref = np.linalg.eigvalsh(h)
        wk, vk = wk.cpu().numpy(), vk.cpu().numpy()
        check_eigenvalues(wk, w_ref, 6)
        r = h @ vk.T - vk.T * wk[None, :]
        assert np.abs(r).max() <= 1e-10 * w_ref.max(), k
        assert np.abs(vk @ vk.T - np.eye(m)).max() <= 1e-10, k
        # padding: exact zeros in the pad components of the structure's modes, pad eigenvalues above the spectrum
        pad_cols = v[k, :m, m:].abs().max().item() if m < s.order else 0.0
        assert pad_cols <= 1e-13, (k, pad_cols)
        if m < s.order:
            assert float(w[k, m:].min()) > w_ref.max()


def test_ragged_kirchhoff_slots_bit_exact(sc):
    """dim = 1: the leading block of every slot is the oracle's Kirchhoff matrix bit for bit; the rest is the pad."""
    import torch

    from springcraft_amd.batch import RaggedBatchSolver

    sizes = [57, 130, 64, 129, 1]
    coords = [synthetic_coord(n, 7 + k) for k, n in enumerate(sizes)]
    s = RaggedBatchSolver(sizes, sc.InvariantForceField(9.0), dim=1)
    mats = s.assemble(_packed(coords)).cpu().numpy()
    torch.cuda.synchronize()
    for k, n in enumerate(sizes):
        ref, _ = orc.compute_kirchhoff(coords[k], orc.invariant_ff(9.0))
        assert np.array_equal(mats[k, :n, :n], ref), k
        pad = mats[k].copy()
        pad[:n, :n] = 0.0
        d = np.diag(pad)[n:]
        assert np.array_equal(pad, np.diag(np.diag(pad))), k          # zero beside and below, diagonal pad
        if n < s.order:
            bound = np.abs(ref).sum(axis=1).max()
            assert np.all(d > max(bound, 0.0)) and np.all(np.diff(d) > 0), k
    w, _ = s.eigh()
    for k, (wk, _) in enumerate(s.results()):
        ref, _ = orc.compute_kirchhoff(coords[k], orc.invariant_ff(9.0))
        w_ref = np.linalg.eigvalsh(ref)
        assert np.abs(wk.cpu().numpy() - w_ref).max() <= 1e-11 * max(np.abs(w_ref).max(), 1.0), k


def test_eanm_and_mass_weighted_hinsen_7cal_in_one_batch(sc):
    """
    Decorated batch at 7cal size (n = 5328): slot 0 the eANM TabulatedForceField (forcefield.py:702-766) against
    BioPhysConnectoR's eigenvalues, slot 1 the mass-weighted Hinsen ANM (anm.py:89-94) against Bio3D's -- the
    reference's own tests tests/test_anm.py:60-84 and :87-142, here as two members of one batched solve --, slot 2 the
    sdENM PatchedForceField of the multi-chain input with masses against Bio3D, slot 3 1l2y (20 atoms) padded to the
    common order.
    """
    import torch

    from springcraft_amd.batch import RaggedBatchSolver

    ca, small = atoms_of(sc, "7cal"), atoms_of(sc, "1l2y")
    n = ca.array_length()
    masses = load_csv("bio3d_mass_7cal.csv.gz")
    sd = sc.TabulatedForceField.sd_enm(ca)
    diff = np.diff(ca.res_id)
    after = np.where((diff > 1) | (diff < 0))[0] + 1
    pairs = np.array([after - 1, after]).T
    sd = sc.PatchedForceField(sd, contact_pair_off=pairs, contact_pair_on=pairs,
                              force_constants=np.full(len(pairs), 43.52 * 0.0083144621 * 300 * 10))
    ffs = [sc.TabulatedForceField.e_anm(ca), sc.HinsenForceField(), sd, sc.TabulatedForceField.e_anm(small)]
    s = RaggedBatchSolver([n, n, n, 20], ffs, dim=3, masses=[None, masses, masses, None], want_vectors=False)
    coords = [np.asarray(ca.coord, dtype=np.float64)] * 3 + [np.asarray(small.coord, dtype=np.float64)]
    s.solve(_packed(coords))
    torch.cuda.synchronize()
    res = [w.cpu().numpy() for w, _ in s.results()]
    assert np.allclose(res[0][6:], load_csv("biophysconnector_anm_eanm_evals_7cal.csv.gz", skip_header=1)[6:])
    assert np.allclose(res[1][6:], load_csv("bio3d_anm_calpha_ff_evals_mw_7cal.csv.gz")[6:], rtol=5e-3, atol=2e-3)
    assert np.allclose(res[2][6:], load_csv("bio3d_anm_sdenm_ff_evals_mw_7cal.csv.gz")[6:], rtol=5e-3, atol=2e-3)
    assert np.allclose(res[3][6:], load_csv("biophysconnector_anm_eanm_evals_1l2y.csv.gz", skip_header=1)[6:])
    # and each one against the single-structure path of the same library
    w_single, _ = sc.ANM(ca, sc.HinsenForceField(), masses=masses).eigen()
    assert np.abs(res[1] - w_single).max() <= 1e-10 * np.abs(w_single).max()


def test_patched_batch_against_oracle(sc):
    """Per-structure patches (shutdown / pair_off / pair_on with constants) and masses in one batch of mixed sizes."""
    import torch

    from springcraft_amd.batch import RaggedBatchSolver

    sizes = [90, 75, 90]
    coords = [synthetic_coord(n, 40 + k) for k, n in enumerate(sizes)]
    rs = np.random.RandomState(3)
    on = np.array([[3, 80], [10, 11], [50, 5]])
    ffs = [
        sc.PatchedForceField(sc.InvariantForceField(8.0), contact_shutdown=np.array([4, 17]),
                             contact_pair_off=np.array([[0, 1], [20, 22]]), contact_pair_on=on,
                             force_constants=np.array([2.5, 0.5, 7.0])),
        sc.HinsenForceField(10.0),
        sc.PatchedForceField(sc.ParameterFreeForceField(9.0), contact_pair_on=np.array([[1, 70]]),
                             force_constants=np.array([3.0])),
    ]
    masses = [rs.rand(90) + 0.5, None, rs.rand(90) + 0.5]
    s = RaggedBatchSolver(sizes, ffs, dim=3, masses=masses)
    mats = s.assemble(_packed(coords)).cpu().numpy()
    torch.cuda.synchronize()
    for k, n in enumerate(sizes):
        h_single, _ = sc.compute_hessian(coords[k], ffs[k])          # single-structure device path (oracle-checked)
        if masses[k] is not None:
            h_single = h_single * orc.mass_weight_matrix(masses[k], 3)
        m = 3 * n
        assert np.abs(mats[k, :m, :m] - h_single).max() <= 1e-12 * np.abs(h_single).max(), k
    s.eigh()
    for k, (wk, vk) in enumerate(s.results()):
        h_single, _ = sc.compute_hessian(coords[k], ffs[k])
        if masses[k] is not None:
            h_single = h_single * orc.mass_weight_matrix(masses[k], 3)
        w_ref = np.linalg.eigvalsh(h_single)
        assert np.abs(wk.cpu().numpy() - w_ref).max() <= 1e-10 * np.abs(w_ref).max(), k


def test_errors(sc):
    from springcraft_amd.batch import RaggedBatchSolver

    class Custom(sc.ForceField):
        def force_constant(self, atom_i, atom_j, sq_distance):
            return np.ones(len(sq_distance))

        @property
        def cutoff_distance(self):
            return 7.0

    with pytest.raises(ValueError):
        RaggedBatchSolver([10, 12], Custom())                      # host-callback force fields have no batched form
    with pytest.raises(ValueError):
        RaggedBatchSolver([10, 12], [sc.InvariantForceField(7.0)])   # one force field per structure
    with pytest.raises(IndexError):
        RaggedBatchSolver([10, 12], sc.InvariantForceField(7.0), masses=[np.ones(10), np.ones(11)])
    with pytest.raises(ValueError):
        RaggedBatchSolver([10, 12], sc.InvariantForceField(7.0), masses=[np.zeros(10), None])
    with pytest.raises(IndexError):
        RaggedBatchSolver([10], sc.PatchedForceField(sc.InvariantForceField(7.0), contact_shutdown=np.array([10])))
    ca = atoms_of(sc, "1l2y")
    with pytest.raises(ValueError):
        RaggedBatchSolver([21], sc.TabulatedForceField.e_anm(ca))    # force field built for 20 atoms
