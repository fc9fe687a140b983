"""
:class:`ANM` — Anisotropic Network Model (host mirror of the reference's anm.py:20-445).
"""

from . import nma
from ._model import ElasticNetworkModel

__all__ = ["ANM"]

K_B = nma.K_B
N_A = nma.N_A


class ANM(ElasticNetworkModel):
    """
    Anisotropic Network Model.

    Parameters
    ----------
    atoms : AtomArray, shape=(n,) or ndarray, shape=(n,3), dtype=float
        The atoms (usually C-alpha only) or their coordinates.
    force_field : ForceField, natoms=n
    masses : bool or ndarray, shape=(n,), dtype=float, optional
        Mass-weight the Hessian with 1/sqrt(m_i m_j); ``True`` infers residue masses from
        ``atoms.res_name`` (needs an AtomArray).
    use_cell_list : bool, optional
        Interface compatibility only; the device contact scan does not need it.

    Attributes
    ----------
    hessian : ndarray, shape=(n*3,n*3), dtype=float
        Partitioned ``[x1, y1, z1, ... xn, yn, zn]``.  Not a copy.
    covariance : ndarray, shape=(n*3,n*3), dtype=float
        Pseudo-inverse of the Hessian.  Not a copy.
    masses : None or ndarray, shape=(n,), dtype=float
    """

    _dim = 3

    @property
    def hessian(self):
        return self._get_matrix()

    @hessian.setter
    def hessian(self, value):
        self._set_matrix(value, IndexError)

    @property
    def covariance(self):
        return self._get_covariance()

    @covariance.setter
    def covariance(self, value):
        self._set_covariance(value)

    def eigen(self, subset_by_index=None):
        """
        Eigenvalues (ascending, shape (3n,)) and eigenvectors (rows, shape (3n,3n)) of the
        Hessian; the first six belong to rigid-body motions (anm.py:150-167).
        """
        return nma.eigen(self, subset_by_index)

    def normal_mode(self, index, amplitude, frames, movement="sine"):
        """Displacements (frames, n, 3) animating mode ``index`` (anm.py:169-207)."""
        return nma.normal_mode(self, index, amplitude, frames, movement)

    def linear_response(self, force):
        """Displacement (n,3) induced by ``force`` via linear response theory (anm.py:209-238)."""
        return nma.linear_response(self, force)

    def frequencies(self):
        """Mode frequencies in arbitrary units, ascending (anm.py:240-256)."""
        return nma.frequencies(self)

    def mean_square_fluctuation(self, mode_subset=None, tem=None, tem_factors=K_B):
        """Per-atom mean square fluctuation (anm.py:258-289)."""
        return nma.mean_square_fluctuation(self, mode_subset, tem, tem_factors)

    def bfactor(self, mode_subset=None, tem=None, tem_factors=K_B):
        """Isotropic B-factors from the MSF (anm.py:291-321)."""
        return nma.bfactor(self, mode_subset, tem, tem_factors)

    def dcc(self, mode_subset=None, norm=True, tem=None, tem_factors=K_B):
        """Dynamic cross-correlation (n,n) between nodes (anm.py:323-382)."""
        return nma.dcc(self, mode_subset, norm, tem, tem_factors)

    def prs_effector_sensor(self, norm=True):
        """PRS matrix plus effector / sensor profiles (anm.py:384-445)."""
        prs_mat = nma.prs(self, norm)
        eff, sens = nma.effector_sensor(prs_mat)
        return prs_mat, eff, sens
