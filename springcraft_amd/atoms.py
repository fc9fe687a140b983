"""
Input adaptor: the reference takes ``biotite.structure.AtomArray`` or ``ndarray (n,3)``
(anm.py:26-28, ``struc.coord`` at anm.py:63).  biotite is an optional dependency here, so atom
containers are duck-typed: anything with ``.coord`` (n,3), annotation arrays and
``array_length()`` is accepted, including biotite's own class when it is installed.
:class:`AtomArray` below is a minimal stand-alone container with the same attribute names.
"""

import numpy as np

__all__ = ["AtomArray", "coord", "is_atom_array", "BadStructureError", "read_pdb_ca"]


class BadStructureError(Exception):
    """Mirror of ``biotite.structure.BadStructureError`` (raised at forcefield.py:442-444)."""


class AtomArray:
    """Minimal annotation-array container (subset of biotite's ``AtomArray``)."""

    def __init__(self, length):
        self._n = int(length)
        self.coord = np.zeros((self._n, 3), dtype=np.float32)
        self.res_name = np.zeros(self._n, dtype="U3")
        self.chain_id = np.zeros(self._n, dtype="U4")
        self.res_id = np.zeros(self._n, dtype=int)
        self.atom_name = np.full(self._n, "CA", dtype="U6")
        self.element = np.full(self._n, "C", dtype="U2")

    def array_length(self):
        return self._n

    def __len__(self):
        return self._n

    def _take(self, index):
        c = np.asarray(self.coord)[index]
        new = AtomArray(len(c))
        new.coord = c.copy()
        for name in ("res_name", "chain_id", "res_id", "atom_name", "element"):
            setattr(new, name, np.asarray(getattr(self, name))[index].copy())
        return new

    def __getitem__(self, index):
        if isinstance(index, (int, np.integer)):
            raise TypeError("single-atom indexing is not supported; use a slice or mask")
        return self._take(index)

    def copy(self):
        return self._take(slice(None))

    def __add__(self, other):
        new = AtomArray(self._n + len(other))
        new.coord = np.concatenate([self.coord, other.coord])
        for name in ("res_name", "chain_id", "res_id", "atom_name", "element"):
            setattr(new, name, np.concatenate([getattr(self, name), getattr(other, name)]))
        return new


def is_atom_array(obj):
    """True for biotite AtomArrays and for anything quacking like one."""
    return (
        hasattr(obj, "coord")
        and hasattr(obj, "array_length")
        and hasattr(obj, "res_name")
        and not isinstance(obj, np.ndarray)
    )


def coord(item):
    """``biotite.structure.coord``: coordinates of an atom container, or the array itself."""
    if hasattr(item, "coord") and not isinstance(item, np.ndarray):
        return np.asarray(item.coord)
    return np.asarray(item)


def read_pdb_ca(path, model=1):
    """
    Tiny fixed-column PDB reader returning the C-alpha atoms (``atom_name == 'CA'`` and
    ``element == 'C'``, the filter of the reference's tests/test_anm.py:17-18) of one model.
    """
    xyz, res_name, chain, res_id = [], [], [], []
    current = 1
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec == "MODEL ":
                current = int(line[6:].split()[0])
            elif rec == "ENDMDL":
                if current == model:
                    break
                current += 1          # files without explicit MODEL numbering
            elif rec == "ATOM  " and current == model:
                if line[12:16].strip() != "CA" or line[76:78].strip() != "C":
                    continue
                xyz.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
                res_name.append(line[17:20])
                chain.append(line[21])
                res_id.append(int(line[22:26]))
    atoms = AtomArray(len(xyz))
    atoms.coord = np.array(xyz, dtype=np.float32).reshape(-1, 3)
    atoms.res_name = np.array(res_name, dtype="U3")
    atoms.chain_id = np.array(chain, dtype="U4")
    atoms.res_id = np.array(res_id, dtype=int)
    return atoms
