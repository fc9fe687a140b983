"""
The data flow of k_bulge_pair (twostage.hip: two consecutive sweeps of the bulge chase per workgroup, the second team taking
its blocks from the first team's LDS slots, shifted by one row and one column) as a NumPy model, checked on the CPU
against the plain task-by-task chase -- including the give-up at any step with the write-back of the live slots and the
take-over from the published counts.  The kernel's index arithmetic was derived from this model
(tools/models/bulge_pair_model.py); the GPU tests (tests/test_two_stage_gpu.py) check the kernel itself.
"""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("bulge_pair_model", os.path.join(ROOT, "tools", "models", "bulge_pair_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _band(m, n, seed):
    rs = np.random.RandomState(seed)
    ab = np.zeros((m.LDAB, n))
    for j in range(n):
        w = min(m.KB, n - 1 - j)
        ab[0:w + 1, j] = rs.standard_normal(w + 1)
    return ab


def test_pair_chase_equals_the_task_by_task_chase():
    m = _model()
    # orders with one, two and several positions per sweep, odd and even sweep counts, full and partial last blocks
    for n in (70, 130, 131, 193, 258):
        ab = _band(m, n, n)
        ref = m.reference_chase(ab, n)
        got = m.pair_chase(ab, n)
        assert np.array_equal(ref, got), n                       # the whole band storage, stale and bulge entries included
        assert np.abs(got[2:m.KB + 1]).max() == 0.0, n           # tridiagonal: nothing left below the first sub-diagonal
        # and it is the same matrix: eigenvalues of the tridiagonal result against the dense band matrix
        a = np.zeros((n, n))
        for j in range(n):
            for d in range(min(m.KB, n - 1 - j) + 1):
                a[j + d, j] = a[j, j + d] = ab[d, j]
        t = np.diag(got[0]) + np.diag(got[1, :n - 1], 1) + np.diag(got[1, :n - 1], -1)
        w_ref = np.linalg.eigvalsh(a)
        assert np.abs(np.linalg.eigvalsh(t) - w_ref).max() <= 1e-12 * np.abs(w_ref).max(), n


def test_give_up_and_take_over_at_every_step():
    m = _model()
    n = 200
    ab = _band(m, n, 5)
    ref = m.reference_chase(ab, n)
    for sA in (0, 70, 136, 196):
        for step in range(0, m.chase_len(n, sA) + 2):
            refl = {}
            part, done = m.pair_chase(ab, n, abort=(sA, step), refl=refl)
            fin = m.finish_with(part, n, done, refl)
            assert np.abs(fin[:2] - ref[:2]).max() < 1e-11 and np.abs(fin[2:m.KB + 1]).max() < 1e-11, (sA, step)


def test_loader_wave_protocol_early_and_late_delivery():
    # round 6: team A's blocks of the common steps arrive by (emulated) LDS-DMA into the slot team B has just emptied; the
    # kernel's offset formulas, the landing layouts and the phases of every slot access -- delivered at once or at the
    # last moment the handshake allows
    m = _model()
    for n in (321, 450):
        ab = _band(m, n, n)
        ref = m.reference_chase(ab, n)
        for late in (False, True):
            assert np.array_equal(ref, m.pair_chase_loader(ab, n, late)), (n, late)
