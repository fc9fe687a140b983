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


# ---- k_gemm3 (csrc/gemm3.hip): the role-split persistent kernel of the short-K updates ----------------------------------
@pytest.fixture(scope="module")
def gemm3():
    import os

    from springcraft_amd import _hip

    if os.environ.get("SPRINGCRAFT_GEMM3") == "0":
        pytest.skip("k_gemm3 is switched off (tools/test_matrix.sh)")

    L = _hip.lib()
    ctx = _hip.context()
    fn = L.sc_dbg_gemm3_host
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_double]
    fn2 = L.sc_dbg_gemm_host
    fn2.restype = C.c_int
    fn2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_double, C.c_double, C.c_int]

    def run(count, m, n, k, layout, lower=0, beta=1.0, seed=0, against_gemm2=True):
        rs = np.random.RandomState(seed)
        A = rs.uniform(-1, 1, (count, m, k))
        B = rs.uniform(-1, 1, (count, k, n))
        C0 = rs.uniform(-1, 1, (count, m, n))
        # per matrix column-major: A (i, kk) at kk*m + i; B (kk, j) at j*k + kk (layout 0) or kk*n + j (layout 2)
        # (layout 1: A stored k-contiguous, per matrix (k x m) column-major = A itself in C order)
        a = np.ascontiguousarray(np.transpose(A, (0, 2, 1))) if layout != 1 else np.ascontiguousarray(A)
        b = np.ascontiguousarray(np.transpose(B, (0, 2, 1))) if layout != 2 else np.ascontiguousarray(B)
        c = np.ascontiguousarray(np.transpose(C0, (0, 2, 1)))
        rc = fn(ctx.handle, a.ctypes.data, b.ctypes.data, c.ctypes.data, count, m, n, k, layout, lower, beta)
        assert rc == 0, rc
        got = np.transpose(c, (0, 2, 1))
        ref = A @ B + (beta * C0 if beta != 0.0 else 0.0)
        if lower:
            ii, jj = np.indices((m, n))
            # entries above the diagonal keep their old value (lower = 2: but for the first super-diagonal entry of every
            # even row, which the band reduction keeps for k_symm3)
            ref = np.where((ii | 1) >= jj if lower == 2 else ii >= jj, ref, C0)
        tol = 4 * np.finfo(float).eps * k * 3
        err = np.abs(got - ref).max()
        assert err <= tol, (count, m, n, k, layout, lower, beta, err, tol)
        if against_gemm2:
            # bit for bit what k_gemm2 computes (same MFMA instruction, k-steps of a dot product in the same order)
            for z in (0, count - 1):
                c2 = np.asfortranarray(C0[z].copy())
                a2 = np.asfortranarray(A[z]) if layout != 1 else np.ascontiguousarray(A[z])
                b2 = np.asfortranarray(B[z]) if layout != 2 else np.ascontiguousarray(B[z])
                rc = fn2(ctx.handle, a2.ctypes.data, b2.ctypes.data, c2.ctypes.data, m, n, k, {0: 0, 1: 2, 2: 1}[layout], 12, 1,
                         1.0, beta, 0)
                assert rc == 0
                ref2 = c2 if not lower else np.where(np.indices((m, n))[0] >= np.indices((m, n))[1], c2, C0[z])
                if lower == 2:                      # (sc_dbg_gemm_host's NT mode keeps the plain lower triangle)
                    ii, jj = np.indices((m, n))
                    assert np.array_equal(got[z][ii >= jj], ref2[ii >= jj])
                elif layout != 2 or lower:          # (sc_dbg_gemm_host's NT mode is lower_only)
                    assert np.array_equal(got[z], ref2), (m, n, k, layout, lower, np.abs(got[z] - ref2).max())
        return got

    return run


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("m,n,k", [(128, 64, 128), (256, 128, 256), (130, 66, 144), (2, 2, 128), (1000, 770, 160),
                                   (48, 6, 512), (384, 1, 128) if False else (384, 2, 128), (1026, 1026, 256)])
@pytest.mark.parametrize("beta", [1.0, 0.0])
def test_gemm3_shapes(gemm3, layout, m, n, k, beta):
    """Full and ragged tiles in both dimensions, both operand layouts, C += A B and C = A B."""
    gemm3(2, m, n, k, layout, beta=beta, seed=m + n + k)


@pytest.mark.parametrize("m", [64, 126, 128, 130, 700, 1030, 2000])
@pytest.mark.parametrize("k", [128, 256])
@pytest.mark.parametrize("lower", [1, 2])
def test_gemm3_lower_only(gemm3, m, k, lower):
    """The trailing update of the band reduction: lower triangle only, entries above the diagonal untouched (lower = 2: plus
    the first super-diagonal entry of every even row)."""
    gemm3(3, m, m, k, 2, lower=lower, seed=m)


def test_gemm3_many_tiles_per_workgroup(gemm3):
    """More tiles than workgroups (every CU walks several tiles: swap, prefetch of the next C, stores of the previous)."""
    gemm3(5, 2048, 1280, 128, 0, seed=1)
    gemm3(5, 2048, 1280, 128, 2, seed=2)
    gemm3(6, 256, 3000, 1744, 1, beta=0.0, seed=5)      # W = V^T Z of the back-transformation: two row tiles, long K
    gemm3(9, 1408, 1408, 256, 2, lower=1, seed=3)
    gemm3(4, 3000, 3000, 256, 0, beta=0.0, seed=4, against_gemm2=False)


def test_gemm3_declines_what_it_does_not_take(gemm3):
    """Odd m, K not a multiple of 16, K < 128: the launcher says no (callers then use k_gemm2)."""
    from springcraft_amd import _hip

    assert gemm3 is not None

    L = _hip.lib()
    ctx = _hip.context()
    z = np.zeros(1 << 16)
    for m, n, k, layout in [(129, 64, 128, 0), (128, 64, 120, 0), (128, 64, 64, 0), (128, 65, 128, 2)]:
        rc = L.sc_dbg_gemm3_host(ctx.handle, z.ctypes.data, z.ctypes.data, z.ctypes.data, 1, m, n, k, layout, 0, C.c_double(1.0))
        assert rc != 0, (m, n, k, layout)


# ---- k_symm3 (csrc/symm3.hip): X = sym(A) V with only (row | 1) >= col of A read
@pytest.fixture(scope="module")
def symm3():
    import os

    from springcraft_amd import _hip

    if os.environ.get("SPRINGCRAFT_SYMM3") == "0":
        pytest.skip("k_symm3 is switched off (tools/test_matrix.sh)")
    L = _hip.lib()
    ctx = _hip.context()
    fn = L.sc_dbg_symm3_host
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]

    def run(count, m, split=1, seed=0):
        rs = np.random.RandomState(seed + m + count)
        out = []
        A = rs.uniform(-1, 1, (count, m, m))
        S = np.tril(A) + np.tril(A, -1).transpose(0, 2, 1)          # the symmetric matrices the lower triangles stand for
        V = rs.uniform(-1, 1, (count, m, 64))
        # what the kernel is handed: the lower triangle, the first super-diagonal entry of every even row equal to its
        # mirror image (what the band reduction keeps), NaN everywhere else above the diagonal -- nothing of it may be read
        stored = np.where(np.tril(np.ones((m, m), dtype=bool)), S, np.nan)
        ev = np.arange(0, m - 1, 2)
        stored[:, ev, ev + 1] = S[:, ev, ev + 1]
        a = np.ascontiguousarray(stored.transpose(0, 2, 1))         # column-major per matrix
        v = np.ascontiguousarray(V.transpose(0, 2, 1))
        x = np.empty_like(v)
        rc = fn(ctx.handle, a.ctypes.data, v.ctypes.data, x.ctypes.data, count, m, split)
        assert rc == 0, rc
        X = x.transpose(0, 2, 1)
        ref = S @ V
        tol = 4 * np.finfo(float).eps * m
        err = np.abs(X - ref).max()
        assert np.isfinite(X).all() and err <= tol * max(1.0, np.abs(ref).max()), (count, m, split, err)
        return out

    return run


@pytest.mark.parametrize("count,m", [(1, 256), (3, 384), (2, 1024), (1, 2000), (5, 784), (40, 512), (2, 1000), (1, 2102),
                                     (3, 258)])
def test_symm3_whole_k(symm3, count, m):
    """One launch, every tile all m columns: full tiles, a partial last tile (m = 2000, 784), orders that are no multiple
    of 16 (a last K step that reaches beyond the matrix), one to many tiles per workgroup."""
    symm3(count, m)


@pytest.mark.parametrize("count,m,split", [(1, 1024, 3), (1, 2000, 5), (2, 768, 2), (1, 4096, 9), (1, 2936, 3)])
def test_symm3_k_slices(symm3, count, m, split):
    """K slices: a slice may start behind a tile's diagonal block or end in front of it."""
    symm3(count, m, split)
