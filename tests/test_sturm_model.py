"""
CPU check of the division-free Sturm count that k_sturm_range uses since round 5 (tools/models/sturm_product_model.py is its
NumPy specification; the GPU tests check the kernel): the count equals the number of eigenvalues below the shift -- against
LAPACK's eigenvalues and against the ratio recurrence of dstebz -- on random matrices, on graded ones that need the
re-normalisation, and on the corner cases the zero rule exists for (shift equal to a diagonal entry, decoupled blocks, zero
rows, repeated eigenvalues).
"""
import numpy as np
import pytest

from tools.models.sturm_product_model import count_product, count_ratio


def eig_count(d, e, x):
    t = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    return int((np.linalg.eigvalsh(t) < x).sum())


@pytest.mark.parametrize("seed", range(6))
def test_random_tridiagonal(seed):
    rs = np.random.RandomState(seed)
    n = int(rs.randint(5, 200))
    d = rs.randn(n) * 10.0 ** rs.randint(-3, 4)
    e = rs.randn(n - 1) * 10.0 ** rs.randint(-3, 4)
    w = np.linalg.eigvalsh(np.diag(d) + np.diag(e, 1) + np.diag(e, -1))
    gaps = np.diff(w)
    for k in rs.choice(n - 1, size=min(20, n - 1), replace=False):
        if gaps[k] <= 1e-9 * max(1.0, np.abs(w).max()):
            continue
        x = 0.5 * (w[k] + w[k + 1])
        assert count_product(d, e, x) == k + 1 == count_ratio(d, e, x)
    lo, hi = w[0] - 1.0 - abs(w[0]), w[-1] + 1.0 + abs(w[-1])
    assert count_product(d, e, lo) == 0 and count_product(d, e, hi) == n


def test_graded_matrix_needs_the_renormalisation():
    # entries over 60 orders of magnitude (1e-45 .. 1e15 of the unit): without the power-of-two rescaling every 4 rows the
    # sequence underflows on the small rows; shifts down to 1e-40 are still counted exactly
    n = 400
    d = 10.0 ** np.linspace(-45, 15, n)
    e = np.sqrt(d[:-1] * d[1:]) * 0.3
    for x in (1e-40, 1e-20, 1e-3, 1.0, 1e9, 9e14):
        assert count_product(d, e, x) == count_ratio(d, e, x)
    n2 = 3000                                    # long and well scaled: growth 3^4 between two rescalings at most
    rs = np.random.RandomState(3)
    d2, e2 = rs.rand(n2), rs.rand(n2 - 1)
    for x in (-0.5, 0.3, 1.1, 2.9):
        assert count_product(d2, e2, x) == count_ratio(d2, e2, x)


def test_the_limit_of_the_product_form_is_far_below_eps():
    # 200 orders of magnitude: the ratio form still counts the eigenvalues around 1e-190 of the norm, the product form
    # does not (at 1e-65 of the norm it is off by one, further down by more) -- but it is exact for every shift above
    # ~1e-60 of the norm, 44 orders below eps |T|
    n = 400
    d = 10.0 ** np.linspace(-100, 100, n)
    e = np.sqrt(d[:-1] * d[1:]) * 0.3
    for x in (1e42, 1e60, 1e99):
        assert count_product(d, e, x) == count_ratio(d, e, x)
    assert abs(count_product(d, e, 1e35) - count_ratio(d, e, 1e35)) <= 2


def test_zero_rule_and_decoupled_blocks():
    # shift exactly on a diagonal entry of a decoupled 1 x 1 block (p becomes exactly 0, and b = 0 follows)
    d = np.array([2.0, 5.0, 1.0, 3.0, 3.0, 7.0])
    e = np.array([1.0, 0.0, 0.0, 2.0, 0.0])       # blocks {0,1}, {2}, {3,4}, {5}
    for x in (1.0, 7.0, 5.0, 3.0, 2.0):
        c = count_product(d, e, x)
        assert c == count_ratio(d, e, x)
        # an eigenvalue that equals the shift may be counted on either side; all others must be counted exactly
        t = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        w = np.linalg.eigvalsh(t)
        below, at = int((w < x - 1e-12).sum()), int((np.abs(w - x) <= 1e-12).sum())
        assert below <= c <= below + at
    # zero rows (an atom without contacts), repeated zero eigenvalues, shift 0 and just around it
    d = np.array([0.0, 0.0, 4.0, 0.0, 1.0, 1.0])
    e = np.array([0.0, 0.0, 0.0, 0.0, 1.0])
    assert count_product(d, e, 1e-9) == count_ratio(d, e, 1e-9) == eig_count(d, e, 1e-9)
    assert count_product(d, e, -1e-9) == count_ratio(d, e, -1e-9) == 0
    c0 = count_product(d, e, 0.0)
    assert 0 <= c0 <= 4                           # four eigenvalues ARE 0: either side is right
    # everything decoupled: the count is the number of diagonal entries below the shift
    rs = np.random.RandomState(1)
    d = rs.randn(50)
    e = np.zeros(49)
    for x in (-0.3, 0.0, 0.7):
        assert count_product(d, e, x) == int((d < x).sum())


def test_monotone_in_the_shift():
    rs = np.random.RandomState(9)
    n = 120
    d, e = rs.randn(n), rs.randn(n - 1)
    xs = np.sort(rs.uniform(-6, 6, size=200))
    counts = [count_product(d, e, x) for x in xs]
    assert all(b >= a for a, b in zip(counts, counts[1:]))
