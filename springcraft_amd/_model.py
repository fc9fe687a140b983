"""
Shared machinery of the :class:`~springcraft_amd.GNM` / :class:`~springcraft_amd.ANM` model
objects: mass handling, lazily cached matrix <-> covariance pair, and the device eigensolve.
(The reference duplicates this logic in gnm.py:58-143 and anm.py:62-148.)
"""

import ctypes as C

import numpy as np

from . import _hip, atoms as _atoms
from .forcefield import device_plan
from .interaction import _assemble, _normalised_patch, _validated_coord

# Average masses (u) of the 20 canonical amino acids as free molecules, i.e. what
# ``biotite.structure.info.mass(res_name, is_residue=True)`` reports from the PDB chemical
# component dictionary; used for ``masses=True`` only when biotite itself is not installed.
_RESIDUE_MASS = {
    "ALA": 89.093, "ARG": 175.209, "ASN": 132.118, "ASP": 133.103, "CYS": 121.158,
    "GLN": 146.144, "GLU": 147.129, "GLY": 75.067, "HIS": 156.162, "ILE": 131.173,
    "LEU": 131.173, "LYS": 147.195, "MET": 149.211, "PHE": 165.189, "PRO": 115.130,
    "SER": 105.093, "THR": 119.119, "TRP": 204.225, "TYR": 181.189, "VAL": 117.146,
}


class _ModesCache:
    """
    Byte-bounded LRU over the device-resident eigenpairs (``_hip.Modes``) of model objects.  A ``Modes`` holds the
    full (n, n) eigenvector matrix in HBM (288 MB for N = 2000 C-alpha); without a bound, a loop that keeps its
    models alive (``for m in models: m.eigen()``) would pin one per model.  The budget is
    ``SPRINGCRAFT_MODES_CACHE_BYTES`` (default 2 GiB); the least recently used entries are released first, an
    entry larger than the whole budget is not kept at all.  The reference caches nothing here (every consumer
    solves again, nma.py:61): dropping an entry only costs the next consumer call of that model a new solve.
    """

    def __init__(self):
        import collections
        import os

        self.budget = int(os.environ.get("SPRINGCRAFT_MODES_CACHE_BYTES", 2 << 30))
        self._entries = collections.OrderedDict()   # id(model) -> (weakref to the model, Modes, bytes)

    @staticmethod
    def _nbytes(modes):
        return 8 * (modes.order * modes.order + modes.order)

    def get(self, model):
        e = self._entries.get(id(model))
        if e is None or e[0]() is not model:
            return None
        self._entries.move_to_end(id(model))
        return e[1]

    def put(self, model, modes):
        import weakref

        self.drop(model)
        size = self._nbytes(modes)
        if size > self.budget:
            return False
        self._entries[id(model)] = (weakref.ref(model), modes, size)
        self._evict()
        return True

    def drop(self, model):
        e = self._entries.pop(id(model), None)
        if e is not None:
            e[1].close()

    def _evict(self):
        total = sum(e[2] for e in self._entries.values())
        while total > self.budget and len(self._entries) > 1:
            _, (_, modes, size) = self._entries.popitem(last=False)
            modes.close()
            total -= size

    def clear(self):
        while self._entries:
            _, (_, modes, _) = self._entries.popitem()
            modes.close()

    def nbytes(self):
        return sum(e[2] for e in self._entries.values())


_modes_cache = _ModesCache()


def residue_mass(res_name):
    try:
        import biotite.structure.info as info  # optional dependency

        return info.mass(res_name, is_residue=True)
    except ImportError:
        return _RESIDUE_MASS[res_name]


class ElasticNetworkModel:
    """Base of GNM (dim=1) and ANM (dim=3)."""

    _dim = 1

    def __init__(self, atoms, force_field, masses=None, use_cell_list=True):
        self._coord = _atoms.coord(atoms)
        self._ff = force_field
        self._use_cell_list = use_cell_list

        # anm.py:67-87 / gnm.py:63-83
        if masses is None or masses is False:
            self._masses = None
        elif masses is True:
            if not _atoms.is_atom_array(atoms):
                raise TypeError("An AtomArray is required to automatically infer masses")
            self._masses = np.array([residue_mass(r) for r in atoms.res_name])
        else:
            # like the reference, a mass array needs an atom container (atoms.array_length())
            if len(masses) != atoms.array_length():
                raise IndexError(f"{len(masses)} masses for {atoms.array_length()} atoms given")
            if np.any(np.asarray(masses) == 0):
                raise ValueError("Masses must not be 0")
            self._masses = np.array(masses, dtype=float)

        self._inv_sqrt_mass = None if self._masses is None else 1 / np.sqrt(self._masses)
        self._matrix = None
        self._covariance = None

    @property
    def masses(self):
        return self._masses

    def release_device_cache(self):
        """Free the device-resident eigenpairs kept for this model (the next consumer call solves again)."""
        _modes_cache.drop(self)

    def __del__(self):
        try:
            _modes_cache.drop(self)
        except Exception:
            pass

    def _size(self):
        return len(self._coord) * self._dim

    # ---- matrix <-> covariance, lazily cached (anm.py:105-148) --------------------------------
    def _get_matrix(self):
        # the matrix object handed out may be edited in place (anm.py:53 "not a copy"), so cached eigenpairs
        # cannot be trusted from here on
        _modes_cache.drop(self)
        if self._matrix is None:
            if self._covariance is None:
                self._matrix, _ = _assemble(self._coord, self._ff, self._dim, self._inv_sqrt_mass)
            else:
                from . import nma

                self._matrix = nma.pinvh(self._covariance, rcond=1e-6)
        return self._matrix

    def _set_matrix(self, value, error=IndexError):
        n = self._size()
        if value.shape != (n, n):
            raise error(f"Expected shape {(n, n)}, got {value.shape}")
        self._matrix = value
        self._covariance = None
        _modes_cache.drop(self)

    def _get_covariance(self):
        _modes_cache.drop(self)
        if self._covariance is None:
            from . import nma

            self._covariance = nma.pinvh(self._get_matrix(), rcond=1e-6)
        return self._covariance

    def _set_covariance(self, value):
        n = self._size()
        if value.shape != (n, n):
            raise IndexError(f"Expected shape {(n, n)}, got {value.shape}")
        self._covariance = value
        self._matrix = None
        _modes_cache.drop(self)

    # ---- eigensolve ---------------------------------------------------------------------------
    def _modes_device(self):
        """
        All eigenpairs as a device-resident :class:`_hip.Modes`.  While neither the matrix nor the covariance
        has been handed out to the caller (so nobody can have edited it) and the force field is evaluated on
        the device, the object is kept in a byte-bounded LRU (:class:`_ModesCache`): ``eigen`` / ``frequencies`` /
        ``mean_square_fluctuation`` / ``dcc`` then share ONE solve, where the reference solves again for each
        (nma.py:61 via :99, :161, :330).  ``release_device_cache()`` drops it explicitly.  Otherwise the current
        host matrix is solved, every time, as the reference does.
        """
        if self._matrix is None and self._covariance is None:
            cached = _modes_cache.get(self)
            if cached is not None:
                return cached
            ff_desc, patch, fused = device_plan(self._ff)
            if fused:
                coord = _validated_coord(self._coord, self._ff)
                keep = []
                patch_desc = _normalised_patch(patch, len(coord), keep)
                ism = None
                if self._inv_sqrt_mass is not None:
                    ism = np.ascontiguousarray(self._inv_sqrt_mass, dtype=np.float64)
                modes = _hip.Modes.from_coord(_hip.context(), coord, self._dim, ff_desc, patch_desc, ism)
                _modes_cache.put(self, modes)   # byte-bounded LRU shared by all models; may evict older entries
                return modes
        matrix = self._get_matrix()
        return _hip.Modes.from_matrix(_hip.context(), matrix, self._dim)

    def _eigen_device(self, subset_by_index=None):
        """
        (eig_values, eig_vectors[rows]) on the device.  When the matrix has not been materialised
        on the host and the force field is evaluated on device, assembly and eigensolve are fused
        (coordinates in, eigenpairs out: the matrix never crosses PCIe); otherwise the host
        matrix (possibly user-assigned, anm.py:120-130) is solved as it is.
        """
        from . import nma

        if subset_by_index is not None:
            lo, hi = (int(x) for x in subset_by_index)
            if self._matrix is None and self._covariance is None and self._dim == 3:
                ff_desc, patch, fused = device_plan(self._ff)
                if fused:
                    coord = _validated_coord(self._coord, self._ff)
                    n = len(coord)
                    if not (0 <= lo <= hi < 3 * n):
                        raise ValueError(f"subset_by_index {subset_by_index} out of range for order {3 * n}")
                    keep = []
                    patch_desc = _normalised_patch(patch, n, keep)
                    pd = C.byref(patch_desc) if patch_desc is not None else None
                    ctx = _hip.context()
                    m = hi - lo + 1
                    w = np.empty(m)
                    v = np.empty((m, 3 * n))
                    ism = None
                    if self._inv_sqrt_mass is not None:
                        ism = np.ascontiguousarray(self._inv_sqrt_mass, dtype=np.float64)
                    ctx.check(_hip.lib().sc_anm_eigen_range_f64(
                        ctx.handle, _hip.ptr(coord), n, C.byref(ff_desc), pd, _hip.ptr(ism), lo, hi,
                        _hip.ptr(w), _hip.ptr(v)))
                    return w, v
            return nma.eigh(self._get_matrix(), subset_by_index=(lo, hi))

        if self._matrix is None and self._covariance is None:
            _, _, fused = device_plan(self._ff)
            if fused:
                return self._modes_device().eigen()
        return nma.eigh(self._get_matrix())
