"""Shared helpers for the test-suite (fixture loading, eigen-comparison metrics)."""
from os.path import abspath, dirname, join

import numpy as np

GOLDEN = join(dirname(abspath(__file__)), "golden")


def ref_data(name):
    """A data fixture copied from the reference's tests/data (CSV, maybe gzipped)."""
    return join(GOLDEN, "ref_data", name)


def load_csv(name, **kw):
    kw.setdefault("delimiter", ",")
    return np.genfromtxt(ref_data(name), **kw)


def generated(name):
    """A vector file written by oracle/make_golden.py (reference imported in the build container)."""
    return np.load(join(GOLDEN, "generated", name))


def structures():
    return generated("structures.npz")


def synthetic_coord(n, seed, box=None):
    # the reference's own generator (tests/test_interaction.py:80-84)
    if box is None:
        box = 5.0 * n ** (1.0 / 3.0)
    rs = np.random.RandomState(seed)
    return rs.rand(n, 3) * box


def forced_two_stage():
    """None, or the tridiagonalisation path forced through SPRINGCRAFT_TWO_STAGE (tools/test_matrix.sh): False / True."""
    import os

    e = os.environ.get("SPRINGCRAFT_TWO_STAGE")
    return None if e is None or e == "" else e != "0"


def oracle_patched(base_ff, natoms, contact_shutdown=None, contact_pair_off=None, contact_pair_on=None,
                   force_constants=None):
    """
    Oracle force field for ``PatchedForceField(base, ...)``: forcefield.py:183-226 restated on top of an oracle base force
    field (the restatement tests/test_oracle_golden.py::test_generated_patched pins to reference-generated vectors); the
    contact patches themselves are applied by the oracle's ``adjacency`` (interaction.py:193-213).
    """
    from oracle import enm_oracle as orc

    cutoff = base_ff.cutoff_distance

    def gamma(i, j, d2):
        if cutoff is None:
            fc = base_ff.gamma(i, j, d2)
        else:
            fc = np.zeros(len(d2))
            m = d2 <= cutoff**2
            fc[m] = base_ff.gamma(i[m], j[m], d2[m])
        if contact_pair_on is not None:
            pm = np.full((natoms, natoms), -1.0)
            pi, pj = np.asarray(contact_pair_on).T
            pm[pi, pj] = force_constants
            pm[pj, pi] = force_constants
            p = pm[i, j]
            fc = np.where(p == -1, fc, p)
        return fc

    return orc.OracleFF(gamma, cutoff, contact_shutdown=contact_shutdown, contact_pair_off=contact_pair_off,
                        contact_pair_on=contact_pair_on)


def pair_digest(pairs):
    import hashlib

    p = np.ascontiguousarray(np.asarray(pairs).astype(np.int64))
    return hashlib.sha256(p.tobytes()).hexdigest()


def check_eigenvalues(w, w_ref, n_trivial, rtol=1e-5):
    """
    SURVEY.md section 8(d) gates: relative 1e-5 on non-trivial modes, absolute
    1e-9 * lambda_max on the trivial (rigid-body) ones.
    """
    w = np.asarray(w)
    w_ref = np.asarray(w_ref)
    assert w.shape == w_ref.shape
    lam_max = np.abs(w_ref).max()
    assert np.all(np.abs(w[:n_trivial]) <= 1e-9 * lam_max), np.abs(w[:n_trivial]).max() / lam_max
    nt = slice(n_trivial, None)
    rel = np.abs(w[nt] - w_ref[nt]) / np.abs(w_ref[nt])
    assert rel.max() <= rtol, rel.max()
    assert np.all(np.diff(w) >= -1e-12 * lam_max), "eigenvalues not ascending"


def check_eigenvectors(a, w, v, tol_res=1e-5, tol_orth=1e-8):
    """Residual ||A v - lambda v|| <= tol * ||A|| and ||V V^T - I||_max <= tol_orth (rows = modes)."""
    a = np.asarray(a)
    norm_a = np.linalg.norm(a, 2) if a.shape[0] <= 2048 else np.abs(w).max()
    r = a @ v.T - v.T * w[None, :]
    res = np.linalg.norm(r, axis=0).max() / norm_a
    assert res <= tol_res, res
    g = v @ v.T
    orth = np.abs(g - np.eye(len(w))).max()
    assert orth <= tol_orth, orth
    return res, orth
