"""Minimal ``biotite.structure`` stand-in (see package docstring)."""
import numpy as np

from . import info  # noqa: F401


class BadStructureError(Exception):
    pass


class AtomArray:
    """Just enough of biotite's AtomArray for springcraft: annotations + coord."""

    def __init__(self, length):
        self._n = length
        self.coord = np.zeros((length, 3), dtype=np.float32)
        self.res_name = np.zeros(length, dtype="U3")
        self.chain_id = np.zeros(length, dtype="U4")
        self.res_id = np.zeros(length, dtype=int)
        self.atom_name = np.zeros(length, dtype="U6")
        self.element = np.zeros(length, dtype="U2")

    def array_length(self):
        return self._n

    def __len__(self):
        return self._n


def coord(item):
    if isinstance(item, AtomArray):
        return item.coord
    return np.asarray(item)


def displacement(a, b):
    return np.asarray(b) - np.asarray(a)


def index_displacement(c, pairs):
    c = coord(c)
    return c[pairs[:, 1]] - c[pairs[:, 0]]


def distance(a, b):
    d = displacement(a, b)
    return np.sqrt((d * d).sum(axis=-1))


class CellList:
    """Brute-force replacement: same predicate as biotite's (d^2 <= r^2, inclusive)."""

    def __init__(self, atom_array, cell_size, periodic=False, box=None, selection=None):
        self._coord = coord(atom_array).astype(np.float64)

    def create_adjacency_matrix(self, threshold_distance):
        d = self._coord[:, None, :] - self._coord[None, :, :]
        sq = np.sum(d * d, axis=-1)
        return sq <= threshold_distance**2
