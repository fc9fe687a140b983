"""
Force fields: spring constants gamma(i, j, d^2) of an elastic network model.

Host-side mirror of the reference's ``springcraft.forcefield`` interface
(forcefield.py:37-114 for the ABC; class-by-class citations below): same class names,
constructor arguments, properties and error behaviour, so that code written against
springcraft runs unchanged.  What is different is *where the constants are evaluated*:
force fields that depend on the distance only (:class:`InvariantForceField`,
:class:`HinsenForceField`, :class:`ParameterFreeForceField`, and a
:class:`PatchedForceField` around one of them) expose a device descriptor and are evaluated
inside the HIP assembly kernels; every other force field (``TabulatedForceField``, user
subclasses) is evaluated by calling its ``force_constant()`` on the device-built pair list.
"""

import abc
import numbers
from os.path import dirname, join, realpath

import numpy as np

from . import _hip
from .atoms import BadStructureError, is_atom_array

__all__ = [
    "ForceField",
    "PatchedForceField",
    "InvariantForceField",
    "HinsenForceField",
    "ParameterFreeForceField",
    "TabulatedForceField",
]

DATA_DIR = join(dirname(realpath(__file__)), "data")

N_AMINO_ACIDS = 20
# Alphabetical by one-letter code: A C D E F G H I K L M N P Q R S T V W Y (forcefield.py:389-394)
AA_LIST = [
    "ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU",
    "MET", "ASN", "PRO", "GLN", "ARG", "SER", "THR", "VAL", "TRP", "TYR",
]
AA_TO_INDEX = {aa: i for i, aa in enumerate(AA_LIST)}


class ForceField(metaclass=abc.ABCMeta):
    """
    Abstract base: subclasses define the force constant of the spring between two atoms
    (reference: forcefield.py:37-114).

    Attributes
    ----------
    cutoff_distance : float or None
        Two atoms interact only if their distance is <= this value; ``None`` = all pairs.
    natoms : int or None
        Number of atoms the force field was built for, ``None`` if it does not depend on them.
    contact_shutdown : ndarray (n,) or None
        Atoms whose contacts are all switched off.
    contact_pair_off, contact_pair_on : ndarray (n,2) or None
        Atom pairs whose contact is switched off / established in any case.
    """

    @abc.abstractmethod
    def force_constant(self, atom_i, atom_j, sq_distance):
        """
        Force constants for the given atom pairs (vectorised over k pairs).
        ``atom_i, atom_j``: int arrays (k,), ``sq_distance``: float array (k,) of *squared*
        distances.  Pairs are already restricted to contacts (forcefield.py:67-94).
        """

    @property
    def cutoff_distance(self):
        return None

    @property
    def contact_shutdown(self):
        return None

    @property
    def contact_pair_off(self):
        return None

    @property
    def contact_pair_on(self):
        return None

    @property
    def natoms(self):
        return None

    # ---- device mapping (not part of the reference interface) ------------------------------
    def _device_kind(self):
        """SC_FF_* code if this exact class is evaluated on device, else None."""
        return None


def _exact_builtin(ff):
    """Only the exact built-in classes are fused; a subclass may override force_constant()."""
    return type(ff) in (InvariantForceField, HinsenForceField, ParameterFreeForceField)


def _check_indices(length, indices):
    # forcefield.py:953-962
    if indices is None or length is None:
        return
    flat = np.asarray(indices).ravel()
    bad = flat[flat >= length]
    if bad.size:
        raise IndexError(f"Index {bad[0]} is out of bounds for a structure of length {length}")


class PatchedForceField(ForceField):
    """
    Wraps another force field and switches selected contacts off / on, optionally with
    individual force constants for the switched-on pairs (reference: forcefield.py:117-261).
    """

    def __init__(self, force_field, contact_shutdown=None, contact_pair_off=None,
                 contact_pair_on=None, force_constants=None):
        self._force_field = force_field
        as_arr = lambda x: None if x is None else np.asarray(x)  # noqa: E731
        self._contact_shutdown = as_arr(contact_shutdown)
        self._contact_pair_off = as_arr(contact_pair_off)
        self._contact_pair_on = as_arr(contact_pair_on)
        self._force_constants = as_arr(force_constants)

        for idx in (self._contact_shutdown, self._contact_pair_off, self._contact_pair_on):
            _check_indices(force_field.natoms, idx)
        if self._contact_pair_on is not None:
            if self._force_constants is None:
                raise TypeError("Individual force constants must be given, if contacts are turned on")
            if len(self._force_constants) != len(self._contact_pair_on):
                raise IndexError(
                    f"{len(self._force_constants)} force constants were given for "
                    f"{len(self._contact_pair_on)} switched on contact_pairs"
                )

    def force_constant(self, atom_i, atom_j, sq_distance):
        # Host evaluation (used on the callback path and by tests); semantics of forcefield.py:183-226
        base = self._force_field
        cutoff = base.cutoff_distance
        if cutoff is None:
            fc = base.force_constant(atom_i, atom_j, sq_distance)
        else:
            inside = sq_distance <= cutoff**2
            fc = np.zeros(len(sq_distance))
            fc[inside] = base.force_constant(atom_i[inside], atom_j[inside], sq_distance[inside])
        if self._contact_pair_on is None:
            return fc
        pi, pj = self._contact_pair_on.T
        size = int(max(pi.max(), pj.max(), np.max(atom_i), np.max(atom_j))) + 1
        override = np.full((size, size), -1.0)
        override[pi, pj] = self._force_constants
        override[pj, pi] = self._force_constants
        picked = override[atom_i, atom_j]
        return np.where(picked == -1, fc, picked)

    @property
    def cutoff_distance(self):
        return self._force_field.cutoff_distance

    def _merged(self, own, inner):
        if inner is None:
            return own
        return np.concatenate([own, inner])

    @property
    def contact_shutdown(self):
        return self._merged(self._contact_shutdown, self._force_field.contact_shutdown)

    @property
    def contact_pair_off(self):
        return self._merged(self._contact_pair_off, self._force_field.contact_pair_off)

    @property
    def contact_pair_on(self):
        return self._merged(self._contact_pair_on, self._force_field.contact_pair_on)

    @property
    def natoms(self):
        return self._force_field.natoms


class InvariantForceField(ForceField):
    """Every contact has the same force constant 1 (reference: forcefield.py:264-289)."""

    def __init__(self, cutoff_distance):
        if cutoff_distance is None:
            raise ValueError("Cutoff distance must be a float")
        self._cutoff_distance = cutoff_distance

    def force_constant(self, atom_i, atom_j, sq_distance):
        return np.ones(len(atom_i))

    @property
    def cutoff_distance(self):
        return self._cutoff_distance

    def _device_kind(self):
        return _hip.SC_FF_INVARIANT


class HinsenForceField(ForceField):
    """
    Hinsen's distance-dependent C-alpha force field (Chem. Phys. 261, 25 (2000)):
    d = max(sqrt(d^2), 2.9 A); gamma = 860 d - 2390 for d < 4 A, else 1.28e6 d^-6
    (reference: forcefield.py:292-330).
    """

    def __init__(self, cutoff_distance=None):
        self._cutoff_distance = cutoff_distance

    def force_constant(self, atom_i, atom_j, sq_distance):
        d = np.maximum(np.sqrt(sq_distance), 2.9)
        near = d * 8.6e2 - 2.39e3
        far = d ** (-6) * 128e4
        return np.where(d < 4.0, near, far)

    @property
    def cutoff_distance(self):
        return self._cutoff_distance

    def _device_kind(self):
        return _hip.SC_FF_HINSEN


class ParameterFreeForceField(ForceField):
    """pfENM (Yang, Song, Jernigan, PNAS 106, 12347 (2009)): gamma = 1/d^2 (forcefield.py:333-366)."""

    def __init__(self, cutoff_distance=None):
        self._cutoff_distance = cutoff_distance

    def force_constant(self, atom_i, atom_j, sq_distance):
        return 1 / sq_distance

    @property
    def cutoff_distance(self):
        return self._cutoff_distance

    def _device_kind(self):
        return _hip.SC_FF_PARAMETER_FREE


def device_plan(force_field):
    """
    Decide how ``force_field`` is evaluated.

    Returns ``(ff_desc, patch_args, fused)``:
      * ``fused`` True: constants are computed inside the HIP kernels from ``ff_desc`` (+patches);
      * ``fused`` False: only the contact scan runs from the descriptor; gamma comes from
        ``force_field.force_constant`` on the host (callback path).
    ``patch_args`` = (shutdown, pair_off, pair_on, on_force_constants, mask_gamma).
    """
    cutoff = force_field.cutoff_distance

    def fusable(ff):
        return _exact_builtin(ff) or (type(ff) is TabulatedForceField and not ff._matrix_exposed)

    def base_desc(ff):
        if type(ff) is TabulatedForceField:
            return ff._tab_desc(_hip.make_ff_desc(_hip.SC_FF_TABULATED, cutoff))
        return _hip.make_ff_desc(ff._device_kind(), cutoff)

    if fusable(force_field):
        return base_desc(force_field), None, True
    if type(force_field) is PatchedForceField and fusable(force_field._force_field):
        patch = (
            force_field._contact_shutdown,
            force_field._contact_pair_off,
            force_field._contact_pair_on,
            force_field._force_constants,
            True,
        )
        return base_desc(force_field._force_field), patch, True
    # callback path: the scan only needs the cutoff and the adjacency patches
    patch = (
        force_field.contact_shutdown,
        force_field.contact_pair_off,
        force_field.contact_pair_on,
        None,
        False,
    )
    if all(p is None for p in patch[:3]):
        patch = None
    return _hip.make_ff_desc(_hip.SC_FF_PARAMETER_FREE, cutoff), patch, False


class TabulatedForceField(ForceField):
    """
    Tabulated force constants by amino-acid type pair, bonded / intra-chain / inter-chain
    relation and distance bin (reference: forcefield.py:369-533).  ``value`` lies in bin ``i``
    when ``value <= cutoff_distance[i]`` and above the previous edge.

    Parameters mirror the reference: ``atoms`` (C-alpha only AtomArray), ``bonded``,
    ``intra_chain``, ``inter_chain`` (scalar, (k,), (20,20) or (20,20,k)), ``cutoff_distance``
    (float, None or increasing bin edges (k,)).
    """

    def __init__(self, atoms, bonded, intra_chain, inter_chain, cutoff_distance):
        if not is_atom_array(atoms):
            raise TypeError(f"Expected 'AtomArray', not {type(atoms).__name__}")
        names = np.asarray(atoms.atom_name)
        elements = np.asarray(atoms.element)
        if not np.all((names == "CA") & (elements == "C")):
            raise BadStructureError("AtomArray does not contain exclusively CA atoms")
        n = atoms.array_length()
        self._natoms = n

        if cutoff_distance is None:
            self._edges, n_bins = None, 1
        elif isinstance(cutoff_distance, numbers.Real):
            self._edges, n_bins = np.array([cutoff_distance]), 1
        else:
            self._edges = np.asarray(cutoff_distance)
            if np.any(np.diff(self._edges) < 0):
                raise ValueError("Distance bin edges are not sorted in increasing order")
            n_bins = len(self._edges)

        self._bonded = _as_table(bonded, n_bins)
        self._intra_chain = _as_table(intra_chain, n_bins)
        self._inter_chain = _as_table(inter_chain, n_bins)

        types = np.array([AA_TO_INDEX[aa] for aa in np.asarray(atoms.res_name)])
        chain = np.asarray(atoms.chain_id)
        res_id = np.asarray(atoms.res_id)

        # non-bonded part by broadcasting: (n, n, bins)
        same_chain = chain[:, None] == chain[None, :]
        ti, tj = types[:, None], types[None, :]
        matrix = np.where(same_chain[:, :, None], self._intra_chain[ti, tj], self._inter_chain[ti, tj])
        # peptide bonds: consecutive residue ids within one chain (forcefield.py:470-473)
        bond = np.where((np.diff(res_id) == 1) & (chain[:-1] == chain[1:]))[0]
        bonded_vals = self._bonded[types[bond], types[bond + 1]]
        matrix[bond, bond + 1] = bonded_vals
        matrix[bond + 1, bond] = bonded_vals
        idx = np.arange(n)
        matrix[idx, idx, :] = 0
        self._interaction_matrix = matrix
        # device descriptor inputs (the tables reproduce the matrix above); handing out the matrix for in-place
        # edits (interaction_matrix property) switches this object to the host-callback path
        _, chain_codes = np.unique(chain, return_inverse=True)
        bonded_next = np.zeros(n, dtype=np.uint8)
        bonded_next[bond] = 1
        self._device_tables = (types.astype(np.int32), chain_codes.astype(np.int32), bonded_next)
        self._matrix_exposed = False

    def force_constant(self, atom_i, atom_j, sq_distance):
        if self._edges is None or len(self._edges) == 1:
            return self._interaction_matrix[atom_i, atom_j, 0]
        bins = np.searchsorted(self._edges**2, sq_distance)
        if (bins >= len(self._edges)).any():
            raise ValueError(
                "Atom interactions above cutoff distance are not allowed in TabulatedForceField"
            )
        return self._interaction_matrix[atom_i, atom_j, bins]

    @property
    def cutoff_distance(self):
        return None if self._edges is None else self._edges[-1]

    @property
    def natoms(self):
        return self._natoms

    @property
    def interaction_matrix(self):
        self._matrix_exposed = True
        return self._interaction_matrix

    def _tab_desc(self, ff_desc):
        types, chain_codes, bonded_next = self._device_tables
        return _hip.make_tab_desc(ff_desc, self._edges if (self._edges is not None and len(self._edges) > 1) else None,
                                  self._bonded, self._intra_chain, self._inter_chain, types, chain_codes,
                                  bonded_next)

    # ---- literature parameter sets (reference: forcefield.py:547-876) -----------------------
    # The published tables ship as CSV data next to this module (springcraft_amd/data).
    @staticmethod
    def s_enm_10(atoms):
        """sENM10 (Dehouck & Mikhailov 2013): type-specific, 10 A cutoff, bonded 10 RT/A^2."""
        fc = _load_table("s_enm_10.csv")
        return TabulatedForceField(atoms, 10.0, fc, fc, 10.0)

    @staticmethod
    def s_enm_13(atoms):
        """sENM13 (Dehouck & Mikhailov 2013): type-specific, 13 A cutoff, bonded 10 RT/A^2."""
        fc = _load_table("s_enm_13.csv")
        return TabulatedForceField(atoms, 10.0, fc, fc, 13.0)

    @staticmethod
    def d_enm(atoms):
        """dENM (Dehouck & Mikhailov 2013): distance-binned only, bonded 46.83 RT/A^2."""
        fc = _load_table("d_enm.csv")
        return TabulatedForceField(atoms, 46.83, fc, fc, _load_table("d_enm_edges.csv"))

    @staticmethod
    def sd_enm(atoms):
        """sdENM (Dehouck & Mikhailov 2013): type- and distance-specific, bonded 43.52 (x RT x 10)."""
        scale = 0.0083144621 * 300 * 10
        fc = _load_table("sd_enm.csv").reshape(-1, 20, 20).T * scale
        return TabulatedForceField(atoms, 43.52 * scale, fc, fc, _load_table("d_enm_edges.csv"))

    @staticmethod
    def e_anm(atoms, nonbonded_mean=False):
        """eANM (Hamacher & McCammon 2006): Miyazawa-Jernigan intra-, Keskin inter-chain, 13 A."""
        return _e_anm_variant(atoms, "miyazawa.csv", "keskin.csv", nonbonded_mean)

    @staticmethod
    def e_anm_mj(atoms, nonbonded_mean=False):
        """eANM with Miyazawa-Jernigan parameters for all non-bonded contacts."""
        return _e_anm_variant(atoms, "miyazawa.csv", "miyazawa.csv", nonbonded_mean)

    @staticmethod
    def e_anm_ke(atoms, nonbonded_mean=False):
        """eANM with Keskin parameters for all non-bonded contacts."""
        return _e_anm_variant(atoms, "keskin.csv", "keskin.csv", nonbonded_mean)


def _e_anm_variant(atoms, intra_name, inter_name, nonbonded_mean):
    intra = _load_table(intra_name)
    inter = _load_table(inter_name)
    if nonbonded_mean:
        intra = np.full((20, 20), np.average(intra))
        inter = np.full((20, 20), np.average(inter))
    return TabulatedForceField(atoms, 82.0, intra, inter, 13.0)


def _as_table(value, n_bins):
    """Normalise a scalar / (k,) / (20,20) / (20,20,k) input to a float32 (20,20,k) table."""
    if np.isnan(value).any():
        raise IndexError("Array contains NaN elements")
    if isinstance(value, numbers.Number):
        return np.full((N_AMINO_ACIDS, N_AMINO_ACIDS, n_bins), value, dtype=np.float32)
    table = np.asarray(value, dtype=np.float32)  # float32 as in the reference (forcefield.py:889-891)
    if table.ndim == 1:
        if len(table) != n_bins:
            raise IndexError(f"Array contains {len(table)} elements for {n_bins} distance bins")
        return np.broadcast_to(table, (N_AMINO_ACIDS, N_AMINO_ACIDS, n_bins)).copy()
    if table.ndim == 2:
        _check_symmetric(table)
        return np.repeat(table[:, :, None], n_bins, axis=2)
    if table.ndim == 3:
        _check_symmetric(table)
        if table.shape[-1] != n_bins:
            raise IndexError(f"Array contains {len(table)} elements for {n_bins} distance bins")
        return table
    raise IndexError(f"Expected array with at most 3 dimensions, {table.ndim} given")


def _check_symmetric(table):
    if table.shape[:2] != (N_AMINO_ACIDS, N_AMINO_ACIDS):
        raise IndexError(
            f"Expected matrix of shape {(N_AMINO_ACIDS, N_AMINO_ACIDS)}, got {table.shape[:2]}"
        )
    if not np.allclose(table, np.swapaxes(table, 0, 1)):
        raise ValueError("Input matrix is not symmetric")


_tables = None


def _load_table(fname):
    """Published parameter tables, packed by tools/make_tables.py into data/enm_tables.npz."""
    global _tables
    if _tables is None:
        with np.load(join(DATA_DIR, "enm_tables.npz")) as z:
            _tables = {k: z[k] for k in z.files}
    return _tables[fname.replace(".csv", "")]
