"""
Normal-mode analysis on top of the device eigensolver.

``eigen`` is the hot-path function (reference: nma.py:29-63): a dense symmetric float64
eigendecomposition, here performed by the hand-written HIP solver (``csrc/eigh*.hip``) instead
of LAPACK ``dsyevd``.  The mode-subset consumers (``frequencies``, ``mean_square_fluctuation``,
``bfactor``, ``dcc``, ``prs``; reference: nma.py:66-359, :476-524) run on the device-resident
eigenpairs (``csrc/consumers.hip``): the (n, n) eigenvector matrix never crosses PCIe for them.
``normal_mode``, ``linear_response`` and ``effector_sensor`` are O(n) / O(n^2) host arithmetic on
results that are already on the host.
"""


import numpy as np

from . import _hip

__all__ = [
    "eigen", "eigh", "pinvh", "frequencies", "mean_square_fluctuation", "bfactor", "dcc",
    "normal_mode", "linear_response", "prs", "effector_sensor",
]

K_B = 1.380649e-23
N_A = 6.02214076e23


def eigh(matrix, eigenvectors=True, subset_by_index=None):
    """
    Device replacement for ``np.linalg.eigh(matrix)`` as used at nma.py:61: ascending
    eigenvalues of a symmetric float64 matrix (lower triangle read) and, as ROWS, the
    corresponding eigenvectors (``eig_vectors[i]`` belongs to ``eig_values[i]``).

    ``subset_by_index=(lo, hi)`` (inclusive, like ``scipy.linalg.eigh``) selects the partial-spectrum
    path: only eigenpairs lo..hi are computed (bisection + inverse iteration), which is what large
    models that only need their slowest modes should use.
    """
    a = np.ascontiguousarray(matrix, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError(f"Expected a square matrix, got shape {a.shape}")
    n = a.shape[0]
    ctx = _hip.context()
    if subset_by_index is not None:
        lo, hi = (int(x) for x in subset_by_index)
        if not (0 <= lo <= hi < n):
            raise ValueError(f"subset_by_index {subset_by_index} out of range for order {n}")
        m = hi - lo + 1
        w = np.empty(m, dtype=np.float64)
        v = np.empty((m, n), dtype=np.float64) if eigenvectors else None
        ctx.check(_hip.lib().sc_eigh_range_f64(ctx.handle, _hip.ptr(a), n, lo, hi, _hip.ptr(w), _hip.ptr(v)))
        return (w, v) if eigenvectors else w
    w = np.empty(n, dtype=np.float64)
    v = _hip.host_array((n, n)) if eigenvectors else None
    ctx.check(_hip.lib().sc_eigh_f64(ctx.handle, _hip.ptr(a), n, _hip.ptr(w), _hip.ptr(v)))
    return (w, v) if eigenvectors else w


def pinvh(matrix, rcond=1e-6):
    """
    Device replacement for ``np.linalg.pinv(matrix, hermitian=True, rcond=rcond)`` as used by the
    ``covariance`` / ``hessian`` / ``kirchhoff`` properties (anm.py:114-117,132-136; gnm.py:107-110,125-131).
    """
    a = np.ascontiguousarray(matrix, dtype=np.float64)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError(f"Expected a square matrix, got shape {a.shape}")
    n = a.shape[0]
    out = _hip.host_array((n, n))
    ctx = _hip.context()
    ctx.check(_hip.lib().sc_pinvh_f64(ctx.handle, _hip.ptr(a), n, float(rcond), _hip.ptr(out)))
    return out


def _model_kind(enm):
    from .anm import ANM
    from .gnm import GNM

    if isinstance(enm, GNM):
        return "gnm", 1
    if isinstance(enm, ANM):
        return "anm", 6
    raise ValueError("Instance of GNM/ANM class expected.")


def eigen(enm, subset_by_index=None):
    """
    Eigenvalues (ascending) and eigenvectors (rows) of the Kirchhoff / Hessian matrix of a
    GNM / ANM (reference: nma.py:29-63).  ``subset_by_index=(lo, hi)`` (extension, inclusive) restricts
    the computation to modes lo..hi.
    """
    _model_kind(enm)
    return enm._eigen_device(subset_by_index)


def frequencies(enm):
    """Frequencies sqrt(lambda)/(2 pi) of all modes; trivial eigenvalues enter as |lambda| (nma.py:66-105)."""
    _, ntriv = _model_kind(enm)
    w = enm._modes_device().values()
    w[:ntriv] = np.abs(w[:ntriv])
    return np.sqrt(w) / (2 * np.pi)


def _mode_selection(enm, mode_subset, n_modes):
    _, ntriv = _model_kind(enm)
    if mode_subset is None:
        return np.arange(ntriv, n_modes)
    subset = np.asarray(mode_subset)
    if np.any(subset < ntriv):
        raise ValueError("Trivial modes are included in the current selection. Please check your input.")
    return subset


def mean_square_fluctuation(enm, mode_subset=None, tem=None, tem_factors=K_B):
    """Per-atom mean square fluctuation sum_k v_k^2 / lambda_k over the selected modes (nma.py:108-184)."""
    _model_kind(enm)
    modes = enm._modes_device()
    contrib = modes.msf(_mode_selection(enm, mode_subset, modes.order))
    if tem is not None:
        contrib = contrib * (tem * tem_factors)
    return contrib


def bfactor(enm, mode_subset=None, tem=None, tem_factors=K_B):
    """Isotropic B-factors 8 pi^2/3 x MSF (nma.py:187-230)."""
    return (8 * np.pi**2) / 3 * mean_square_fluctuation(enm, mode_subset, tem, tem_factors)


def dcc(enm, mode_subset=None, norm=True, tem=None, tem_factors=K_B):
    """Dynamic cross-correlation between nodes over the selected modes (nma.py:233-359)."""
    _model_kind(enm)
    modes = enm._modes_device()
    if mode_subset is None:
        # all modes: the reference takes the covariance matrix here (nma.py:324-336), i.e. pinv(M, hermitian=True,
        # rcond=1e-6): every mode with |lambda| > 1e-6 max|lambda| -- which drops the trivial modes and, on a nearly
        # disconnected network, also non-trivial modes below that threshold
        w = modes.values()
        sel = np.nonzero(np.abs(w) > 1e-6 * np.abs(w).max())[0]
    else:
        sel = _mode_selection(enm, mode_subset, modes.order)
    # sum_k <v_k[a], v_k[b]> / lambda_k, normalised by sqrt(c_aa c_bb) on request
    cov = modes.dcc(sel, norm)
    if tem is not None:  # applied after the normalisation, as the reference does (nma.py:355-357)
        cov = cov * tem * tem_factors
    return cov


def normal_mode(anm, index, amplitude, frames, movement="sine"):
    """Displacement trajectory (frames, n, 3) for one oscillation of mode ``index`` (nma.py:363-419)."""
    from .anm import ANM

    if not isinstance(anm, ANM):
        raise ValueError("Instance of ANM class expected.")
    _, v = eigen(anm)
    mode = v[index].reshape(-1, 3)
    mode = mode * (amplitude / np.sqrt((mode**2).sum(axis=-1)).max())
    phase = np.linspace(0, 1, frames, endpoint=False)
    if movement == "sine":
        scale = np.sin(phase * 2 * np.pi)
    elif movement == "triangle":
        # triangle wave from -1 (phase 0) over +1 (phase 1/2) back to -1 (nma.py:413)
        scale = 2 * np.abs(2 * (phase - np.floor(phase + 0.5))) - 1
    else:
        raise ValueError(f"Movement '{movement}' is unknown")
    return scale[:, None, None] * mode[None, :, :]


def linear_response(anm, force):
    """Linear-response displacement covariance . force, reshaped to (n,3) (nma.py:422-473)."""
    from .anm import ANM

    if not isinstance(anm, ANM):
        raise ValueError("Instance of ANM class expected.")
    force = np.asarray(force)
    n3 = anm.covariance.shape[0]
    if force.ndim == 2:
        if force.shape != (n3 // 3, 3):
            raise ValueError(f"Expected force with shape {(n3 // 3, 3)}, got {force.shape}")
        force = force.ravel()
    elif force.ndim == 1:
        if len(force) != n3:
            raise ValueError(f"Expected force with length {n3}, got {len(force)}")
    else:
        raise ValueError(f"Expected 1D or 2D array, got {force.ndim} dimensions")
    return (anm.covariance @ force).reshape(-1, 3)


def prs(anm, norm=True):
    """Perturbation-response-scanning matrix from the squared covariance (nma.py:476-524)."""
    from .anm import ANM

    if not isinstance(anm, ANM):
        raise ValueError("Instance of ANM class expected.")
    if anm._covariance is not None:
        # a covariance the caller assigned (or already fetched): reduce that very matrix
        c2 = anm._covariance**2
        n = c2.shape[0] // 3
        mat = c2.reshape(n, 3, n, 3).sum(axis=(1, 3))
        if norm:
            mat = mat / np.diag(mat)[:, None]
        return mat
    # covariance = pinv(hessian, rcond=1e-6) (anm.py:114-117) formed and reduced on the device
    return anm._modes_device().prs(1e-6, norm)


def effector_sensor(prs_matrix):
    """Row / column means of the off-diagonal PRS entries (nma.py:527-569)."""
    m = np.array(prs_matrix, dtype=float)
    n = len(m)
    off = m - np.diag(np.diag(m))
    return off.sum(axis=1) / (n - 1), off.sum(axis=0) / (n - 1)
