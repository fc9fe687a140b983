"""
Kirchhoff / Hessian assembly on the MI355X.

Host mirror of the reference's ``springcraft.interaction`` (``compute_kirchhoff``
interaction.py:14-54, ``compute_hessian`` interaction.py:57-111): same signatures, return
values, dtypes and errors.  The work itself — contact scan, ordered pair list, matrix fill,
diagonal reduction — runs in the HIP kernels of ``csrc/assembly.hip`` through the C ABI.
"""

import ctypes as C

import numpy as np

from . import _hip
from .forcefield import device_plan

__all__ = ["compute_kirchhoff", "compute_hessian"]


def compute_kirchhoff(coord, force_field, use_cell_list=True):
    """
    Kirchhoff matrix of the atoms at ``coord`` under ``force_field``.

    Parameters
    ----------
    coord : ndarray, shape=(n,3), dtype=float
    force_field : ForceField, natoms=n
    use_cell_list : bool, optional
        Accepted for interface compatibility (interaction.py:25-31).  The device contact scan is
        an exact float64 all-pairs tile scan either way; results do not depend on this flag.

    Returns
    -------
    kirchhoff : ndarray, shape=(n,n), dtype=float
    pairs : ndarray, shape=(k,2), dtype=int
        Interacting atom pairs, sorted by first then second index, both directions.
    """
    return _assemble(coord, force_field, dim=1)


def compute_hessian(coord, force_field, use_cell_list=True):
    """
    Hessian matrix (3n x 3n, partitioned ``[x1, y1, z1, ... xn, yn, zn]``) of the atoms at
    ``coord`` under ``force_field``; see :func:`compute_kirchhoff` for the arguments.

    Returns
    -------
    hessian : ndarray, shape=(n*3,n*3), dtype=float
    pairs : ndarray, shape=(k,2), dtype=int
    """
    return _assemble(coord, force_field, dim=3)


def _validated_coord(coord, force_field):
    # interaction.py:43,88 (float64) and :141-147 (shape / natoms checks)
    coord = np.asarray(coord).astype(np.float64, copy=False)
    if coord.ndim != 2 or coord.shape[1] != 3:
        raise ValueError(f"Expected coordinates with shape (n,3), got {coord.shape}")
    if force_field.natoms is not None and len(coord) != force_field.natoms:
        raise ValueError(
            f"Got coordinates for {len(coord)} atoms, "
            f"but forcefield was built for {force_field.natoms} atoms"
        )
    return np.ascontiguousarray(coord)


def _normalised_patch(patch, n, keep):
    """numpy-style negative indices -> non-negative; bool masks -> indices; then the C descriptor."""
    if patch is None:
        return None
    shutdown, pair_off, pair_on, fcs, mask_gamma = patch

    def norm(x):
        if x is None:
            return None
        a = np.asarray(x)
        if a.dtype == bool:
            a = np.where(a)[0]
        a = a.astype(np.int64)
        if ((a < -n) | (a >= n)).any():
            bad = a[(a < -n) | (a >= n)].ravel()[0]
            raise IndexError(f"index {bad} is out of bounds for axis 0 with size {n}")
        return np.where(a < 0, a + n, a)

    pair_on_n = norm(pair_on)
    if pair_on_n is not None and len(pair_on_n) and (pair_on_n[:, 0] == pair_on_n[:, 1]).any():
        raise ValueError("Cannot turn on interaction of an atom with itself")  # interaction.py:210-211
    return _hip.make_patch_desc(norm(shutdown), norm(pair_off), pair_on_n, fcs, mask_gamma, keep)


def _pair_list(ctx, coord, ff_desc, patch_desc, want_sq_dist):
    """Ordered (k,2) int64 pair list (+ squared distances) from the device contact scan."""
    L = _hip.lib()
    n = len(coord)
    k = C.c_int64(0)
    pd = C.byref(patch_desc) if patch_desc is not None else None
    ctx.check(L.sc_contacts(ctx.handle, _hip.ptr(coord), n, C.byref(ff_desc), pd, None, C.byref(k)))
    pairs = np.empty((k.value, 2), dtype=np.int64)
    sq = np.empty(k.value, dtype=np.float64) if want_sq_dist else None
    if k.value:
        k2 = C.c_int64(0)
        ctx.check(L.sc_pairs(ctx.handle, _hip.ptr(coord), n, C.byref(ff_desc), pd, k.value,
                             _hip.ptr(pairs), _hip.ptr(sq), C.byref(k2)))
        assert k2.value == k.value
    return pairs, sq


def _assemble(coord, force_field, dim, inv_sqrt_mass=None):
    coord = _validated_coord(coord, force_field)
    n = len(coord)
    ctx = _hip.context()
    L = _hip.lib()
    ff_desc, patch, fused = device_plan(force_field)
    keep = []
    patch_desc = _normalised_patch(patch, n, keep)
    pd = C.byref(patch_desc) if patch_desc is not None else None
    w = None
    if inv_sqrt_mass is not None:
        w = np.ascontiguousarray(inv_sqrt_mass, dtype=np.float64)
    matrix = _hip.host_array((n * dim, n * dim))   # (large ones on pooled page-locked memory: the copy back is 3-5 x faster)

    if fused:
        pairs, _ = _pair_list(ctx, coord, ff_desc, patch_desc, want_sq_dist=False)
        fn = L.sc_kirchhoff_f64 if dim == 1 else L.sc_hessian_f64
        ctx.check(fn(ctx.handle, _hip.ptr(coord), n, C.byref(ff_desc), pd, _hip.ptr(w), _hip.ptr(matrix)))
        return matrix, pairs

    # callback path: device builds pairs + d^2, Python evaluates gamma, device fills the matrix
    pairs, sq_dist = _pair_list(ctx, coord, ff_desc, patch_desc, want_sq_dist=True)
    gamma = force_field.force_constant(pairs[:, 0], pairs[:, 1], sq_dist)
    gamma = np.ascontiguousarray(gamma, dtype=np.float64)  # Tabulated returns float32 (forcefield.py:889)
    if gamma.shape != (len(pairs),):
        raise ValueError(f"force_constant() returned shape {gamma.shape} for {len(pairs)} pairs")
    if dim == 1:
        ctx.check(L.sc_kirchhoff_from_pairs_f64(ctx.handle, n, _hip.ptr(pairs), len(pairs),
                                                _hip.ptr(gamma), _hip.ptr(matrix)))
    else:
        ctx.check(L.sc_hessian_from_pairs_f64(ctx.handle, _hip.ptr(coord), n, _hip.ptr(pairs),
                                              len(pairs), _hip.ptr(gamma), _hip.ptr(matrix)))
    if w is not None:
        w_full = np.repeat(w, dim) if dim > 1 else w
        matrix *= np.outer(w_full, w_full)
    return matrix, pairs
