"""
GPU unit tests of the f64 MFMA GEMM launch paths (csrc/gemm_f64.hip) against NumPy: every block tile of k_gemm2, the
three operand layouts, split-K, the lower-triangle launch grid of the band reduction's SYR2K (entries above the diagonal
must stay untouched), the XCD pairing of launches with 2-4 row tiles, ragged sizes, K = 0.  The triangular-operand and gather launches are covered through the solvers that use them
(tests/test_two_stage_gpu.py, tests/test_eigh_gpu.py).  Tolerance: K * 4 ulp of the largest partial product sum.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gemm():
    from springcraft_amd import _hip

    L = _hip.lib()
    ctx = _hip.context()
    fn = L.sc_dbg_gemm_host
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_double, C.c_double, C.c_int]

    def run(m, n, k, mode, tile, split=1, alpha=1.0, beta=0.0, lower_grid=0, seed=0):
        rs = np.random.RandomState(seed)
        # mode 0: A m x k col-major, B k x n col-major; 1: B stored n x k (C = A B^T, lower only); 2: A stored k x m
        A = rs.uniform(-1, 1, (m, k))
        B = rs.uniform(-1, 1, (k, n))
        C0 = rs.uniform(-1, 1, (m, n))
        a = np.asfortranarray(A) if mode != 2 else np.ascontiguousarray(A)       # (i, kk) at kk*m+i  |  i*k+kk
        b = np.asfortranarray(B) if mode != 1 else np.ascontiguousarray(B)       # (kk, j) at j*k+kk  |  kk*n+j
        c = np.asfortranarray(C0.copy())
        rc = fn(ctx.handle, a.ctypes.data, b.ctypes.data, c.ctypes.data, m, n, k, mode, tile, split, alpha, beta, lower_grid)
        assert rc == 0, ctx.last_error() if hasattr(ctx, "last_error") else rc
        ref = alpha * (A @ B) + (beta * C0 if beta != 0.0 else 0.0)
        if mode == 1:
            ii, jj = np.indices((m, n))
            ref = np.where(ii >= jj, ref, C0)      # lower_only: entries above the diagonal keep their old value
        tol = 4 * np.finfo(float).eps * max(k, 1) * (abs(alpha) + abs(beta) + 1)
        err = np.abs(c - ref).max()
        assert err <= tol, (m, n, k, mode, tile, split, err, tol)
        return c

    return run


@pytest.mark.parametrize("tile", [10, 11, 12, 13])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_layouts_and_tiles(gemm, tile, mode):
    if mode == 1:
        gemm(517, 517, 77, 1, tile, beta=1.0, alpha=-1.0)
    else:
        gemm(517, 333, 130, mode, tile, beta=1.0, alpha=-1.0)
        gemm(130, 517, 333, mode, tile)


@pytest.mark.parametrize("tile", [10, 11, 12, 13])
@pytest.mark.parametrize("m", [64, 127, 128, 129, 700, 1030])
def test_lower_triangle_grid(gemm, tile, m):
    """SYR2K launch: only tiles on / below the diagonal are started; result equals the rectangular launch."""
    c1 = gemm(m, m, 128, 1, tile, alpha=-1.0, beta=1.0, lower_grid=1, seed=m)
    c0 = gemm(m, m, 128, 1, tile, alpha=-1.0, beta=1.0, lower_grid=0, seed=m)
    assert np.array_equal(c0, c1)


@pytest.mark.parametrize("m", [129, 256, 300, 512])
def test_few_row_tiles_pairing(gemm, m):
    """2-4 row tiles: the XCD-paired 1-D launch (the V^T Z products of the back-transformation: layout k-contiguous)."""
    gemm(m, 1500, 640, 2, 10)
    gemm(m, 1500, 640, 0, 12, beta=1.0)
    gemm(m, 70, 33, 2, 13)


@pytest.mark.parametrize("split", [2, 3, 8])
def test_split_k(gemm, split):
    gemm(64, 192, 1999, 2, 10, split=split)
    gemm(257, 1001, 777, 2, 12, split=split)
    gemm(300, 200, 500, 0, 13, split=split)


def test_degenerate_sizes(gemm):
    for tile in (10, 11, 12, 13):
        gemm(300, 200, 0, 0, tile, beta=1.0)          # K = 0: C <- beta C
        gemm(300, 200, 0, 0, tile, beta=0.0)          # K = 0: C <- 0
        gemm(1, 1, 1, 0, tile)
        gemm(5, 3, 17, 2, tile, beta=1.0, alpha=2.5)
        gemm(130, 1, 16, 0, tile)
        gemm(1, 130, 15, 0, tile)
