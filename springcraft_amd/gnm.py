"""
:class:`GNM` — Gaussian Network Model (host mirror of the reference's gnm.py:20-303).
"""

from . import nma
from ._model import ElasticNetworkModel

__all__ = ["GNM"]

K_B = nma.K_B
N_A = nma.N_A


class GNM(ElasticNetworkModel):
    """
    Gaussian Network Model.

    Parameters
    ----------
    atoms : AtomArray, shape=(n,) or ndarray, shape=(n,3), dtype=float
    force_field : ForceField, natoms=n
    masses : bool or ndarray, shape=(n,), dtype=float, optional
    use_cell_list : bool, optional
        Interface compatibility only.

    Attributes
    ----------
    kirchhoff : ndarray, shape=(n,n), dtype=float.  Not a copy.
    covariance : ndarray, shape=(n,n), dtype=float  (pseudo-inverse of the Kirchhoff matrix).  Not a copy.
    masses : None or ndarray, shape=(n,), dtype=float
    """

    _dim = 1

    @property
    def kirchhoff(self):
        return self._get_matrix()

    @kirchhoff.setter
    def kirchhoff(self, value):
        # the reference raises ValueError here and IndexError in every other setter (gnm.py:115-120)
        self._set_matrix(value, ValueError)

    @property
    def covariance(self):
        return self._get_covariance()

    @covariance.setter
    def covariance(self, value):
        self._set_covariance(value)

    def eigen(self, subset_by_index=None):
        """Eigenvalues (ascending, (n,)) and eigenvectors (rows, (n,n)) of the Kirchhoff matrix (gnm.py:145-158)."""
        return nma.eigen(self, subset_by_index)

    def frequencies(self):
        """Mode frequencies in arbitrary units (gnm.py:160-176)."""
        return nma.frequencies(self)

    def mean_square_fluctuation(self, mode_subset=None, tem=None, tem_factors=K_B):
        """Per-atom mean square fluctuation (gnm.py:178-212)."""
        return nma.mean_square_fluctuation(self, mode_subset, tem, tem_factors)

    def bfactor(self, mode_subset=None, tem=None, tem_factors=K_B):
        """Isotropic B-factors (gnm.py:214-244)."""
        return nma.bfactor(self, mode_subset, tem, tem_factors)

    def dcc(self, mode_subset=None, norm=True, tem=None, tem_factors=K_B):
        """Dynamic cross-correlation (n,n) (gnm.py:246-303)."""
        return nma.dcc(self, mode_subset, norm, tem, tem_factors)
