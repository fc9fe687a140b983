// Two-stage tridiagonalisation for large matrices (replaces the one-stage panel algorithm of tridiag.hip, whose
// matrix-vector products are bound by HBM bandwidth: 4/3 n^3 bytes per matrix).
//
//   stage 1  sy2sb   dense -> band of half-width kB = 64.  Per 64-column panel: Householder QR of the block below
//                    the band (k_panel_qr, one launch per column, the panel streamed through LDS), then the
//                    two-sided update  A22 <- Q^T A22 Q = A22 - V W^T - W V^T  in which all O(n^2 b) work is f64-MFMA
//                    GEMM: X = A22 V (two triangular-operand GEMMs on the lower-stored A22), V^T [X|V] (Gram, split-K),
//                    W = [X|V] [T; -S/2] and the SYR2K  [V|W] [W|V]^T.
//   stage 2  sb2st   band -> tridiagonal by Householder bulge chasing (Lang's algorithm: sweep s annihilates column s
//                    below the first sub-diagonal and chases the bulge down the band, one reflector of length <= 64
//                    per block).  Task (s, k) only conflicts with (s+1, k-1) and later, so launch t runs every task
//                    with 2 s + k = t, one workgroup each (k_bulge_step): 2 n + n/64 launches, no spin-waits.
//   back-transformation  Z <- Q1 Q2 Z.  Q2: the reflectors of 64 consecutive sweeps at the same chase position form
//                    a "diamond" (127 x 64 parallelogram) = one compact-WY block I - V T V^T; diamonds (S, k) are
//                    applied in wavefronts 3 (Smax - S) + k = const as two grouped GEMMs per wavefront (bt2).
//                    Q1: the stage-1 reflectors through the block back-transformation of backtransform.hip.
//
// Role in the reference: part of np.linalg.eigh (LAPACK dsyevd) at nma.py:61; LAPACK itself uses the one-stage dsytrd.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "eigh_internal.h"

namespace {

constexpr int kB = 64;            // band half-width = panel width = reflector length of stage 2
constexpr int kG = 64;            // sweeps per diamond
constexpr int kLdab = 2 * kB;     // rows of the band storage: AB(i, j) = ab[(i - j) + j * kLdab]
constexpr int kDiaLd = 128;       // leading dimension of a diamond (kB + kG - 1 = 127 rows used)
constexpr int kDiaSize = kDiaLd * kG;
constexpr int kQrRows = 128;      // rows of the panel one k_panel_qr workgroup owns
constexpr int kSmallSplit = 8;    // split-K of the V^T [X1|X2|V] product

struct HH {
  double beta, tau, scale;
};

// LAPACK dlarfg scalars: x = (alpha, tail), xn2 = ||tail||^2;  H x = beta e1,  v = (1, scale * tail)
__device__ __forceinline__ HH householder(double alpha, double xn2) {
  HH h;
  if (xn2 == 0.0) {
    h.beta = alpha; h.tau = 0.0; h.scale = 0.0;
    return h;
  }
  h.beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
  h.tau = (h.beta - alpha) / h.beta;
  h.scale = 1.0 / (alpha - h.beta);
  return h;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ================================================================================================================
// Stage 1: panel QR.  Panel = A[r0 : n, j0 : j0 + kB] (m x kB, column-major, ld n).  Launch j (0 .. nr):
//   (a) j >= 1: finish reflector j-1 from the partial results of launch j-1 (tail Gram row, pivot row) and apply it
//       to columns j .. kB-1; column j-1 becomes (R entries above, beta at the pivot, v below); v also goes, with its
//       explicit 1 and zeros above, into the three panel buffers [V|W], [W|V], [X1|X2|V];
//   (b) j < nr: tail Gram row of column j:  g[c] = sum_{r > j} P[r, j] P[r, c]  (per 128-row chunk; chunk 0 also saves
//       the pivot row P[j, j..]) for launch j+1.
// Grid (chunks, batch), 256 threads; the chunk's columns j-1 .. kB-1 live in LDS for the duration of the launch.
__global__ __launch_bounds__(256) void k_panel_qr(double* __restrict__ a_all, long long stride_a,
                                                  double* __restrict__ tri_all, TriLayout TL,
                                                  double* __restrict__ sb_all, SbLayout SL, int j0, int j, int nr,
                                                  int c_end) {
  // c_end: columns j .. c_end-1 are updated (kB: the whole rest of the panel; blocked panels: the rest of the
  // 8-column inner block, the other columns get the inner block's reflectors at once from k_pqr_blk_a / _b)
  constexpr int LD = kQrRows + 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* P = sm;                      // [kB][LD]   P[c * LD + r]
  double* wv = sm + kB * LD;           // [kB]  w_c of the reflector being applied
  double* vv = wv + kB;                // [kQrRows]  v_r
  double* red = vv + kQrRows;          // [4][kB]
  __shared__ double s_scale, s_beta, s_tau;

  const int n = TL.n;
  const int r0 = j0 + kB, m = n - r0;
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  double* sb = sb_all + (size_t)blockIdx.y * SL.slab;
  const int chunk = blockIdx.x, nchunks = gridDim.x;
  const int row_base = chunk * kQrRows;   // local (panel) row of this chunk's first row
  const int tid = threadIdx.x;
  const int prev = j - 1;
  const int c_lo = j > 0 ? j - 1 : 0;
  // launch j reads what launch j-1 left and writes for launch j+1 while other workgroups may still be reading:
  // two copies, alternating
  const int nchunk_cap = (n + kQrRows - 1) / kQrRows + 1;
  const double* part_in = sb + SL.qrpart + (size_t)((j + 1) & 1) * nchunk_cap * kB;
  double* part_out = sb + SL.qrpart + (size_t)(j & 1) * nchunk_cap * kB;
  const double* piv_in = sb + SL.qrpiv + (size_t)((j + 1) & 1) * (kB + 8);
  double* piv_out = sb + SL.qrpiv + (size_t)(j & 1) * (kB + 8);

  // (a0) reflector scalars and w
  if (j >= 1) {
    {
      // tail Gram row of column prev: sum of the chunks' partial rows, four threads per column, loads four deep
      const int cq = tid & 63, qq = tid >> 6;
      double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
      int ch = qq;
      for (; ch + 12 < nchunks; ch += 16) {
        g0 += part_in[(size_t)ch * kB + cq];
        g1 += part_in[(size_t)(ch + 4) * kB + cq];
        g2 += part_in[(size_t)(ch + 8) * kB + cq];
        g3 += part_in[(size_t)(ch + 12) * kB + cq];
      }
      for (; ch < nchunks; ch += 4) g0 += part_in[(size_t)ch * kB + cq];
      red[qq * kB + cq] = (g0 + g1) + (g2 + g3);
    }
    __syncthreads();
    if (tid < kB) red[tid] = (red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]);
    __syncthreads();
    if (tid == 0) {
      const double alpha = piv_in[prev];
      const HH h = householder(alpha, red[prev]);
      s_scale = h.scale; s_beta = h.beta; s_tau = h.tau;
      if (chunk == 0) tri[TL.tau + j0 + prev] = h.tau;
    }
    __syncthreads();
    if (tid < kB && tid >= j) {
      const double prow = piv_in[tid];
      wv[tid] = s_tau * (prow + s_scale * red[tid]);
    }
  }

  // (1) chunk -> LDS: lanes along the rows (contiguous in memory)
  // (loads are unconditional, with clamped indices, and issued eight at a time: a load that is merged with a zero
  //  under a predicate makes hipcc wait for it before issuing the next one)
  {
    const int r = tid & (kQrRows - 1), half = tid >> 7;
    const int rl = row_base + r;
    const double* src = A + (size_t)j0 * n + r0 + std::min(rl, m - 1);
    for (int c = c_lo + half; c < c_end; c += 16) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[(size_t)std::min(c + 2 * u, kB - 1) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + 2 * u < c_end) P[(c + 2 * u) * LD + r] = rl < m ? t[u] : 0.0;
    }
  }
  __syncthreads();

  const int c = tid & 63, q = tid >> 6;   // column / row quarter (32 rows) of this thread in the compute phases
  if (j >= 1) {
    // (a1) v
    if (tid < kQrRows) {
      const int rl = row_base + tid;
      double v = 0.0;
      if (rl < m) v = rl > prev ? s_scale * P[prev * LD + tid] : (rl == prev ? 1.0 : 0.0);
      vv[tid] = v;
    }
    __syncthreads();
    // (a2) P[:, c] -= v w_c
    if (c >= j && c < c_end) {
      const double w = wv[c];
#pragma unroll 8
      for (int r = q * 32; r < q * 32 + 32; ++r) P[c * LD + r] -= vv[r] * w;
    }
    // column prev: v below the pivot, beta at it (rows above keep their R entries); and the panel buffers
    if (tid < kQrRows) {
      const int rl = row_base + tid;
      if (rl < m) {
        const double v = vv[tid];
        if (rl > prev) P[prev * LD + tid] = v;
        else if (rl == prev) P[prev * LD + tid] = s_beta;
        const size_t row = (size_t)r0 + rl;
        sb[SL.vw + (size_t)prev * n + row] = v;
        sb[SL.wv + (size_t)(kB + prev) * n + row] = v;
        sb[SL.xv + (size_t)(2 * kB + prev) * n + row] = v;
      }
    }
    __syncthreads();
  }

  if (j < nr) {
    // (b) tail Gram row of column j over this chunk: rows with local index > j
    double acc = 0.0;
    if (c >= j && c < c_end) {
#pragma unroll 8
      for (int r = q * 32; r < q * 32 + 32; ++r) {
        const int rl = row_base + r;
        if (rl > j && rl < m) acc += P[j * LD + r] * P[c * LD + r];
      }
    }
    red[q * kB + c] = acc;
    __syncthreads();
    if (tid < kB) {
      part_out[(size_t)chunk * kB + tid] = (red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]);
      if (chunk == 0 && tid >= j && tid < c_end) piv_out[tid] = P[tid * LD + j];   // pivot row (alpha at [j])
    }
  } else {
    // last launch of the panel: columns without a reflector (short last panel) are zero in the V buffers
    for (int cc = nr; cc < kB; ++cc)
      for (int r = tid; r < kQrRows; r += 256) {
        const int rl = row_base + r;
        if (rl < m) {
          const size_t row = (size_t)r0 + rl;
          sb[SL.vw + (size_t)cc * n + row] = 0.0;
          sb[SL.wv + (size_t)(kB + cc) * n + row] = 0.0;
          sb[SL.xv + (size_t)(2 * kB + cc) * n + row] = 0.0;
        }
      }
  }

  // (2) LDS -> chunk
  if (j >= 1) {
    const int r = tid & (kQrRows - 1), half = tid >> 7;
    const int rl = row_base + r;
    if (rl < m)
      for (int cc = c_lo + half; cc < c_end; cc += 2) A[(size_t)(j0 + cc) * n + r0 + rl] = P[cc * LD + r];
  }
}

// ---- blocked panels: the 8 reflectors of an inner block [c0, c0+8) applied to the columns to their right at once ----
// k_pqr_blk_a: finishes reflector c0+7 (the inner block's last) and forms, per 128-row chunk, the partial products
//   M[i][c] = v_{c0+i} . P[:, c]  for c = c0 .. kB-1  (the first 8 columns are the Gram matrix of the block's reflectors).
// k_pqr_blk_b: sums them, builds the 8 x 8 T factor, W = T^T M, updates P[:, c] -= V W for c >= c0+8 and leaves the
//   tail Gram row / pivot row of column c0+8 for the next inner block's first column launch.
constexpr int kIb = 8;

// explicit form of the inner block's reflectors in the LDS copy of chunk 0 (memory keeps R above the pivots)
__device__ __forceinline__ void blk_explicit_v(double* P, int LD, int c0, int ncols, int row_base) {
  if (row_base != 0) return;   // pivot rows are local rows c0 .. c0+7 of the first chunk
  for (int idx = threadIdx.x; idx < ncols * kB; idx += 256) {
    const int i = idx / kB, r = idx % kB;   // rows 0..63 suffice (pivots < 64)
    const int piv = c0 + i;
    if (r < piv) P[(c0 + i) * LD + r] = 0.0;
    else if (r == piv) P[(c0 + i) * LD + r] = 1.0;
  }
}

__global__ __launch_bounds__(256) void k_pqr_blk_a(double* __restrict__ a_all, long long stride_a,
                                                   double* __restrict__ tri_all, TriLayout TL,
                                                   double* __restrict__ sb_all, SbLayout SL, int j0, int c0) {
  constexpr int LD = kQrRows + 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* P = sm;                      // [kB][LD]
  double* vv = sm + kB * LD;           // [kQrRows]
  double* red = vv + kQrRows;          // [4][2][kB]
  __shared__ double s_scale, s_beta, s_tau;
  const int n = TL.n;
  const int r0 = j0 + kB, m = n - r0;
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  double* sb = sb_all + (size_t)blockIdx.y * SL.slab;
  const int chunk = blockIdx.x, nchunks = gridDim.x;
  const int row_base = chunk * kQrRows;
  const int tid = threadIdx.x;
  const int prev = c0 + kIb - 1;
  const int nchunk_cap = (n + kQrRows - 1) / kQrRows + 1;
  const double* part_in = sb + SL.qrpart + (size_t)(prev & 1) * nchunk_cap * kB;
  const double* piv_in = sb + SL.qrpiv + (size_t)(prev & 1) * (kB + 8);

  // scalars of reflector prev
  {
    double g = 0.0;
    for (int ch = tid; ch < nchunks; ch += 256) g += part_in[(size_t)ch * kB + prev];
    g = wave_sum(g);
    if ((tid & 63) == 0) red[tid >> 6] = g;
    __syncthreads();
    if (tid == 0) {
      const HH h = householder(piv_in[prev], (red[0] + red[1]) + (red[2] + red[3]));
      s_scale = h.scale; s_beta = h.beta; s_tau = h.tau;
      if (chunk == 0) tri[TL.tau + j0 + prev] = h.tau;
    }
  }
  // chunk (columns c0 .. kB-1) -> LDS
  {
    const int r = tid & (kQrRows - 1), half = tid >> 7;
    const int rl = row_base + r;
    const double* src = A + (size_t)j0 * n + r0 + std::min(rl, m - 1);
    for (int c = c0 + half; c < kB; c += 16) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[(size_t)std::min(c + 2 * u, kB - 1) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + 2 * u < kB) P[(c + 2 * u) * LD + r] = rl < m ? t[u] : 0.0;
    }
  }
  __syncthreads();
  // v of reflector prev: to memory (v below the pivot, beta at it), to the panel buffers, explicit form in LDS
  if (tid < kQrRows) {
    const int rl = row_base + tid;
    double v = 0.0;
    if (rl < m) v = rl > prev ? s_scale * P[prev * LD + tid] : (rl == prev ? 1.0 : 0.0);
    if (rl < m) {
      if (rl > prev) A[(size_t)(j0 + prev) * n + r0 + rl] = v;
      else if (rl == prev) A[(size_t)(j0 + prev) * n + r0 + rl] = s_beta;
      const size_t row = (size_t)r0 + rl;
      sb[SL.vw + (size_t)prev * n + row] = v;
      sb[SL.wv + (size_t)(kB + prev) * n + row] = v;
      sb[SL.xv + (size_t)(2 * kB + prev) * n + row] = v;
    }
    P[prev * LD + tid] = v;
  }
  __syncthreads();
  blk_explicit_v(P, LD, c0, kIb - 1, row_base);
  __syncthreads();
  // M partial: thread (c, q): 8 dot products over its 32 rows
  const int c = tid & 63, q = tid >> 6;
  double acc[kIb];
#pragma unroll
  for (int i = 0; i < kIb; ++i) acc[i] = 0.0;
  if (c >= c0) {
#pragma unroll 4
    for (int r = q * 32; r < q * 32 + 32; ++r) {
      const double x = P[c * LD + r];
#pragma unroll
      for (int i = 0; i < kIb; ++i) acc[i] += P[(c0 + i) * LD + r] * x;
    }
  }
  double* p8 = sb + SL.qrpart8 + (size_t)chunk * kIb * kB;
#pragma unroll
  for (int pass = 0; pass < kIb / 2; ++pass) {
    red[(q * 2 + 0) * kB + c] = acc[2 * pass];
    red[(q * 2 + 1) * kB + c] = acc[2 * pass + 1];
    __syncthreads();
    if (tid < 2 * kB) {
      const int h = tid >> 6, cc = tid & 63;
      p8[(size_t)(2 * pass + h) * kB + cc] =
          (red[(0 * 2 + h) * kB + cc] + red[(1 * 2 + h) * kB + cc]) + (red[(2 * 2 + h) * kB + cc] + red[(3 * 2 + h) * kB + cc]);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_pqr_blk_b(double* __restrict__ a_all, long long stride_a,
                                                   const double* __restrict__ tri_all, TriLayout TL,
                                                   double* __restrict__ sb_all, SbLayout SL, int j0, int c0) {
  constexpr int LD = kQrRows + 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* P = sm;                      // [kB][LD]
  double* Ms = sm + kB * LD;           // [kIb][kB]  M, later W
  double* red = Ms + kIb * kB;         // [4][kB]
  __shared__ double T[kIb][kIb];
  const int n = TL.n;
  const int r0 = j0 + kB, m = n - r0;
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  const double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  double* sb = sb_all + (size_t)blockIdx.y * SL.slab;
  const int chunk = blockIdx.x, nchunks = gridDim.x;
  const int row_base = chunk * kQrRows;
  const int tid = threadIdx.x;
  const int jn = c0 + kIb;             // first column to the right = next pivot column
  const int nchunk_cap = (n + kQrRows - 1) / kQrRows + 1;
  double* part_out = sb + SL.qrpart + (size_t)(jn & 1) * nchunk_cap * kB;
  double* piv_out = sb + SL.qrpiv + (size_t)(jn & 1) * (kB + 8);

  // chunk (columns c0 .. kB-1) -> LDS (issued first: the loads fly while the partial products are summed)
  {
    const int r = tid & (kQrRows - 1), half = tid >> 7;
    const int rl = row_base + r;
    const double* src = A + (size_t)j0 * n + r0 + std::min(rl, m - 1);
    for (int c = c0 + half; c < kB; c += 16) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[(size_t)std::min(c + 2 * u, kB - 1) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + 2 * u < kB) P[(c + 2 * u) * LD + r] = rl < m ? t[u] : 0.0;
    }
  }
  // M = sum of the chunks' partial products
  for (int idx = tid; idx < kIb * kB; idx += 256) {
    const double* p8 = sb + SL.qrpart8 + idx;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
    int ch = 0;
    for (; ch + 3 < nchunks; ch += 4) {
      g0 += p8[(size_t)ch * kIb * kB];
      g1 += p8[(size_t)(ch + 1) * kIb * kB];
      g2 += p8[(size_t)(ch + 2) * kIb * kB];
      g3 += p8[(size_t)(ch + 3) * kIb * kB];
    }
    for (; ch < nchunks; ++ch) g0 += p8[(size_t)ch * kIb * kB];
    Ms[idx] = (g0 + g1) + (g2 + g3);
  }
  __syncthreads();
  // T (larft, forward columnwise) of the inner block: G[l][q] = Ms[l][c0 + q]
  if (tid == 0) {
    for (int qq = 0; qq < kIb; ++qq) {
      const double tau = tri[TL.tau + j0 + c0 + qq];
      for (int a = 0; a < qq; ++a) {
        double s2 = 0.0;
        for (int l = a; l < qq; ++l) s2 += T[a][l] * Ms[l * kB + c0 + qq];
        T[a][qq] = -tau * s2;
      }
      T[qq][qq] = tau;
      for (int a = qq + 1; a < kIb; ++a) T[a][qq] = 0.0;
    }
  }
  __syncthreads();
  // W = T^T M (in place, column by column: each thread owns column c)
  if (tid < kB && tid >= jn) {
    double mcol[kIb], wcol[kIb];
#pragma unroll
    for (int l = 0; l < kIb; ++l) mcol[l] = Ms[l * kB + tid];
#pragma unroll
    for (int i = 0; i < kIb; ++i) {
      double s2 = 0.0;
#pragma unroll
      for (int l = 0; l <= i; ++l) s2 += T[l][i] * mcol[l];
      wcol[i] = s2;
    }
#pragma unroll
    for (int i = 0; i < kIb; ++i) Ms[i * kB + tid] = wcol[i];
  }
  __syncthreads();
  blk_explicit_v(P, LD, c0, kIb, row_base);
  __syncthreads();
  // P[:, c] -= V W[:, c]
  const int c = tid & 63, q = tid >> 6;
  if (c >= jn) {
    double wcol[kIb];
#pragma unroll
    for (int i = 0; i < kIb; ++i) wcol[i] = Ms[i * kB + c];
#pragma unroll 4
    for (int r = q * 32; r < q * 32 + 32; ++r) {
      double s2 = 0.0;
#pragma unroll
      for (int i = 0; i < kIb; ++i) s2 += P[(c0 + i) * LD + r] * wcol[i];
      P[c * LD + r] -= s2;
    }
  }
  __syncthreads();
  // tail Gram row and pivot row of column jn over the next inner block
  {
    double acc = 0.0;
    if (c >= jn && c < jn + kIb) {
#pragma unroll 8
      for (int r = q * 32; r < q * 32 + 32; ++r) {
        const int rl = row_base + r;
        if (rl > jn && rl < m) acc += P[jn * LD + r] * P[c * LD + r];
      }
    }
    red[q * kB + c] = acc;
    __syncthreads();
    if (tid < kB) {
      part_out[(size_t)chunk * kB + tid] = (red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]);
      if (chunk == 0 && tid >= jn && tid < jn + kIb) piv_out[tid] = P[tid * LD + jn];
    }
  }
  // LDS -> chunk (columns to the right of the inner block)
  {
    const int r = tid & (kQrRows - 1), half = tid >> 7;
    const int rl = row_base + r;
    if (rl < m)
      for (int cc = jn + half; cc < kB; cc += 2) A[(size_t)(j0 + cc) * n + r0 + rl] = P[cc * LD + r];
  }
}

// One workgroup per matrix: T (larft, forward columnwise) from tau and G = V^T V;  S = T^T (V^T X) T;
// C = [T; T; -S/2]  (3 kB x kB, column-major), the right-hand factor of  W = [X1 | X2 | V] C.
__global__ __launch_bounds__(256) void k_sb_small(const double* __restrict__ tri_all, TriLayout TL,
                                                  double* __restrict__ sb_all, SbLayout SL, int j0) {
  constexpr int LD = kB + 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* G = sm;                 // [kB][LD]  G[i * LD + j]
  double* M1 = G + kB * LD;
  double* T = M1 + kB * LD;
  double* U = T + kB * LD;
  const double* tri = tri_all + (size_t)blockIdx.x * TL.slab;
  double* sb = sb_all + (size_t)blockIdx.x * SL.slab;
  const int tid = threadIdx.x;
  // the split-K product is (kB x 3 kB), column-major ld kB: columns [X1 | X2 | V]
  const double* prod = sb + SL.small;
  const size_t slice = (size_t)kB * 3 * kB;
  for (int idx = tid; idx < kB * kB; idx += 256) {
    const int i = idx & 63, jj = idx >> 6;
    double g = 0.0, x = 0.0;
    for (int s = 0; s < kSmallSplit; ++s) {
      const double* ps = prod + s * slice;
      x += ps[i + (size_t)jj * kB] + ps[i + (size_t)(kB + jj) * kB];
      g += ps[i + (size_t)(2 * kB + jj) * kB];
    }
    G[i * LD + jj] = g;
    M1[i * LD + jj] = x;
    T[i * LD + jj] = 0.0;
  }
  __syncthreads();
  for (int qq = 0; qq < kB; ++qq) {
    const double tau = tri[TL.tau + j0 + qq];
    double s = 0.0;
    if (tid < qq)
      for (int l = tid; l < qq; ++l) s += T[tid * LD + l] * G[l * LD + qq];
    __syncthreads();
    if (tid < qq) T[tid * LD + qq] = -tau * s;
    if (tid == qq) T[qq * LD + qq] = tau;
    __syncthreads();
  }
  // U = M1 T ; S = T^T U
  for (int idx = tid; idx < kB * kB; idx += 256) {
    const int i = idx >> 6, jj = idx & 63;
    double s = 0.0;
    for (int l = 0; l <= jj; ++l) s += M1[i * LD + l] * T[l * LD + jj];
    U[i * LD + jj] = s;
  }
  __syncthreads();
  double* cm = sb + SL.cmat;   // ld 3 kB
  for (int idx = tid; idx < kB * kB; idx += 256) {
    const int i = idx & 63, jj = idx >> 6;
    double s = 0.0;
    for (int l = 0; l <= i; ++l) s += T[l * LD + i] * U[l * LD + jj];
    const double t = T[i * LD + jj];
    cm[i + (size_t)jj * 3 * kB] = t;
    cm[kB + i + (size_t)jj * 3 * kB] = t;
    cm[2 * kB + i + (size_t)jj * 3 * kB] = -0.5 * s;
  }
}

// ================================================================================================================
// Band storage and stage 2
__global__ __launch_bounds__(256) void k_band_extract(const double* __restrict__ a_all, long long stride_a,
                                                      double* __restrict__ sb_all, SbLayout SL) {
  const int n = SL.n;
  const double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* ab = sb_all + (size_t)blockIdx.y * SL.slab + SL.ab;
  const size_t total = (size_t)kLdab * n;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int d = (int)(idx % kLdab), jj = (int)(idx / kLdab);
    ab[idx] = (d <= kB && jj + d < n) ? A[(size_t)jj * n + jj + d] : 0.0;
  }
}

__global__ __launch_bounds__(256) void k_band_to_tri(const double* __restrict__ sb_all, SbLayout SL,
                                                     double* __restrict__ tri_all, TriLayout TL) {
  const int n = SL.n;
  const double* ab = sb_all + (size_t)blockIdx.y * SL.slab + SL.ab;
  double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    tri[TL.d + i] = ab[(size_t)i * kLdab];
    tri[TL.e + i] = i < n - 1 ? ab[(size_t)i * kLdab + 1] : 0.0;
  }
}

// After the band has been copied out: make the stage-1 reflectors a plain sub-matrix view of A for the block
// back-transformation (unit entry at row c + kB, zeros between the diagonal block and the unit entry).
__global__ __launch_bounds__(256) void k_sb_clean(double* __restrict__ a_all, long long stride_a, int n) {
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  const int c = blockIdx.x;   // column
  for (int r = c + 1 + threadIdx.x; r <= std::min(c + kB, n - 1); r += 256) A[(size_t)c * n + r] = r == c + kB ? 1.0 : 0.0;
}

// Number of chase positions of sweep s: blocks of kB rows from row s+1 to n-1
__host__ __device__ inline int chase_len(int n, int s) { return (n - 1 - s + kB - 1) / kB; }

// Launch t of the bulge chase: workgroup x handles task (s, k) with k = (t & 1) + 2x, s = (t - k) / 2.
__global__ __launch_bounds__(256) void k_bulge_step(double* __restrict__ sb_all, SbLayout SL,
                                                    int t) {
  constexpr int LD = kB + 1;
  __shared__ double E[kB * LD];
  double* D = E;   // the diagonal block is processed after E has gone back to memory: same buffer
  __shared__ double vp[kB], vn[kB], u[kB], red[4 * kB];
  __shared__ double s_tau, s_beta;

  const int n = SL.n;
  const int k = (t & 1) + 2 * (int)blockIdx.x;
  const int s = (t - k) / 2;
  if (s < 0 || s > n - 3 || k >= chase_len(n, s)) return;
  double* sb = sb_all + (size_t)blockIdx.y * SL.slab;
  double* ab = sb + SL.ab;
  const int S = s / kG, cc = s - S * kG;
  // diamonds of group S start at sum_{S' < S} chase_len(n, 64 S') = S K0 - S (S - 1) / 2  (K0 = chase_len(n, 0))
  const size_t dia = (size_t)S * chase_len(n, 0) - (size_t)S * (S - 1) / 2 + k;
  // diamonds are stored row-major (sweep index contiguous): element (row r, sweep c) at [c + r * kG]
  double* vd = sb + SL.vd + dia * kDiaSize + (size_t)cc * kG + cc;   // this reflector's first entry (row cc, column cc)
  const int tid = threadIdx.x;
  const int i = tid & 63, q = tid >> 6;

  const int r0 = s + 1 + k * kB;             // first row of the reflector being generated
  const int L = std::min(kB, n - r0);        // its length (>= 1)

  // the diagonal block D = AB(r0 .. r0+L-1, r0 .. r0+L-1) (lower stored) is not touched before its own update:
  // fetch it right away (unconditional loads with clamped indices, masked when they go to LDS)
  double d16[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int jc = std::min(q * 16 + u, L - 1);
    const int ic = std::min(std::max(i, jc), L - 1);
    d16[u] = ab[(size_t)(ic - jc) + (size_t)(r0 + jc) * kLdab];
  }

  if (k > 0) {
    const int c0 = r0 - kB;
    const double* vdp = sb + SL.vd + (dia - 1) * kDiaSize + (size_t)cc * kG + cc;
    const double tau_p = sb[SL.tau2 + (dia - 1) * kG + cc];
    if (tid < kB) vp[tid] = vdp[(size_t)tid * kG];
    // E(i, j) = AB(r0 + i, c0 + j), rows i < L
    {
      double t16[16];
      const int ic = std::min(i, L - 1);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int jj = q * 16 + u;
        t16[u] = ab[(size_t)(kB + ic - jj) + (size_t)(c0 + jj) * kLdab];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) E[i * LD + q * 16 + u] = i < L ? t16[u] : 0.0;
    }
    __syncthreads();
    // u = E vp
    {
      double a = 0.0;
#pragma unroll
      for (int jj = q * 16; jj < q * 16 + 16; ++jj) a += E[i * LD + jj] * vp[jj];
      red[q * kB + i] = a;
    }
    __syncthreads();
    if (tid < kB) u[tid] = tau_p * ((red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]));
    __syncthreads();
#pragma unroll
    for (int jj = q * 16; jj < q * 16 + 16; ++jj) E[i * LD + jj] -= u[i] * vp[jj];
    __syncthreads();
    // new reflector from the first column of E
    if (tid < 64) {
      const double x = E[tid * LD];
      const double t2 = wave_sum((tid >= 1 && tid < L) ? x * x : 0.0);
      const HH h = householder(E[0], t2);
      vn[tid] = tid == 0 ? 1.0 : (tid < L ? x * h.scale : 0.0);
      if (tid == 0) { s_tau = h.tau; s_beta = h.beta; }
    }
    __syncthreads();
    // z_j = sum_i E[i, j] v_i  (j >= 1);  thread (j = i, rows q*16..)
    {
      double a = 0.0;
#pragma unroll
      for (int ii = q * 16; ii < q * 16 + 16; ++ii) a += E[ii * LD + i] * vn[ii];
      red[q * kB + i] = a;
    }
    __syncthreads();
    if (tid < kB) u[tid] = s_tau * ((red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]));
    __syncthreads();
    // E <- H E, first column = beta e1; write back
    for (int jj = q * 16; jj < q * 16 + 16; ++jj) {
      double e = E[i * LD + jj] - vn[i] * u[jj];
      if (jj == 0) e = i == 0 ? s_beta : 0.0;
      if (i < L) ab[(size_t)(kB + i - jj) + (size_t)(c0 + jj) * kLdab] = e;
    }
    __syncthreads();   // D reuses E's buffer
  } else {
    // sweep start: x = AB(s+1 .. s+L, s)
    if (tid < 64) {
      const double xr = ab[(size_t)(1 + std::min(tid, L - 1)) + (size_t)s * kLdab];
      const double x = tid < L ? xr : 0.0;
      const double t2 = wave_sum(tid >= 1 ? x * x : 0.0);
      const double alpha = __shfl(x, 0);
      const HH h = householder(alpha, t2);
      vn[tid] = tid == 0 ? 1.0 : x * h.scale;
      if (tid == 0) { s_tau = h.tau; s_beta = h.beta; }
      if (tid < L) ab[(size_t)(1 + tid) + (size_t)s * kLdab] = tid == 0 ? h.beta : 0.0;
    }
    __syncthreads();
  }

  // two-sided update of the diagonal block D = AB(r0 .. r0+L-1, r0 .. r0+L-1) (lower stored)
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int jj = q * 16 + u;
    if (i >= jj) {
      const double x = (i < L) ? d16[u] : 0.0;
      D[i * LD + jj] = x;
      D[jj * LD + i] = x;
    }
  }
  __syncthreads();
  {
    double a = 0.0;
#pragma unroll
    for (int jj = q * 16; jj < q * 16 + 16; ++jj) a += D[i * LD + jj] * vn[jj];
    red[q * kB + i] = a;
  }
  __syncthreads();
  if (tid < 64) {
    const double p = s_tau * ((red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]));
    const double dot = wave_sum(p * vn[tid]);
    const double alpha2 = -0.5 * s_tau * dot;
    u[tid] = p + alpha2 * vn[tid];   // w
  }
  __syncthreads();
  for (int jj = q * 16; jj < q * 16 + 16; ++jj)
    if (i >= jj && i < L)
      ab[(size_t)(i - jj) + (size_t)(r0 + jj) * kLdab] = D[i * LD + jj] - vn[i] * u[jj] - u[i] * vn[jj];

  // the reflector goes into its diamond
  if (tid < L) vd[(size_t)tid * kG] = vn[tid];
  if (tid == 0) sb[SL.tau2 + dia * kG + cc] = s_tau;
}

// ================================================================================================================
// Diamonds: T factor and V T.  One workgroup per (diamond, matrix).  The diamond is held compactly in LDS
// (Vc[c][i] = V[c + i, c], the 64 entries of reflector c), G and T share one buffer (G is only read by the first
// wave's recurrence, which then writes T over it).
__global__ __launch_bounds__(256) void k_dia_tfactor(double* __restrict__ sb_all, SbLayout SL, int dia0) {
  constexpr int LD = kG + 1;
  __shared__ double Vc[kG * LD];     // Vc[c * LD + i]
  __shared__ double GT[kG * LD];     // G[a * LD + b] (a < b), overwritten column by column with T
  double* sb = sb_all + (size_t)blockIdx.y * SL.slab;
  const size_t dia = (size_t)dia0 + blockIdx.x;
  const double* vd = sb + SL.vd + dia * kDiaSize;
  const double* tau = sb + SL.tau2 + dia * kG;
  const int tid = threadIdx.x;
  {
    // row r of the diamond holds sweeps c in [r - 63, r]; lanes along c (contiguous in memory)
    const int c = tid & 63, q = tid >> 6;
    for (int i = q; i < kB; i += 4) Vc[c * LD + i] = vd[(size_t)(c + i) * kG + c];
  }
  __syncthreads();
  // G[a, b] = v_a . v_b for a < b: rows b .. a + 63, i.e. entries i = b - a .. 63 of v_a against 0 .. of v_b
  {
    const int a = tid & 63, q = tid >> 6;
    for (int b = q * 16; b < q * 16 + 16; ++b) {
      double s = 0.0;
      if (a < b) {
        const int sh = b - a;
        for (int i = sh; i < kB; ++i) s += Vc[a * LD + i] * Vc[b * LD + i - sh];
      }
      GT[a * LD + b] = s;
    }
  }
  __syncthreads();
  // T (larft, forward columnwise): T[0:q, q] = -tau_q T[0:q, 0:q] G[0:q, q], T[q, q] = tau_q.  Lane a of the first wave
  // keeps row a of T in registers (T[a, l] = 0 for l < a), G comes as LDS broadcasts: 2016 fully unrolled FMAs per lane
  // and no synchronisation inside the recurrence.
  if (tid < kG) {
    double trow[kG];
#pragma unroll
    for (int qq = 0; qq < kG; ++qq) {
      double s = 0.0;
#pragma unroll
      for (int l = 0; l < qq; ++l) s += trow[l] * GT[l * LD + qq];
      const double tq = tau[qq];
      trow[qq] = tid < qq ? -tq * s : (tid == qq ? tq : 0.0);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): every lane has finished reading G before T overwrites it
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int qq = 0; qq < kG; ++qq) GT[tid * LD + qq] = trow[qq];
  }
  __syncthreads();
  // VT[r, c] = sum_l V[r, l] T[l, c]   (128 x 64, column-major ld 128); V[r, l] = Vc[l][r - l] for 0 <= r - l < 64
  double* vt = sb + SL.vt2 + dia * kDiaSize;
  {
    const int r = tid & 127, half = tid >> 7;
    for (int c = half * 32; c < half * 32 + 32; ++c) {
      double s = 0.0;
      const int l0 = std::max(0, r - (kB - 1)), l1 = std::min(c, r);
      for (int l = l0; l <= l1; ++l) s += Vc[l * LD + r - l] * GT[l * LD + c];
      vt[(size_t)c * kDiaLd + r] = s;
    }
  }
}

// ================================================================================================================
// Z <- Q2 Z, fused.  The columns of Z are independent, so one workgroup takes kNc = 32 columns through ALL diamonds
// (sweep groups last to first, chase positions first to last) with no synchronisation between workgroups:
//   W1^T (32 x 64) = Z^T (32 x 128) VD (128 x 64)           A operand: LDS copy of the Z window, B: VD from L2
//   Z^T (32 x 128) -= W1^T (32 x 64) VT^T (64 x 128)          A operand: W1 via LDS, B: VT from L2
// The 128-row window of Z stays in registers as MFMA accumulators (wave w owns rows 16w.. of each 64-row half); going
// from one chase position to the next, the lower half becomes the upper half (register rename), 64 finished rows are
// stored and 64 new rows are loaded: Z is read once and written once per sweep group.
constexpr int kNc = 32;
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k_bt2_fused(const double* __restrict__ sb_all, SbLayout SL,
                                                      const int* __restrict__ dia_off, double* __restrict__ z_all,
                                                      long long stride_z, int ncols, int batch, int xcd_map) {
  constexpr int LDH = kB + 2;      // Z window copy: [col][row in half], 66: conflict-free A-operand reads
  constexpr int LDW = kNc + 16;    // W1 copy: [sweep][col]
  __shared__ double Zs[2][kNc * LDH];
  __shared__ double W1s[kG * LDW];
  const int n = SL.n;
  // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), and every workgroup
  // streams all diamonds of its matrix through that L2.  With xcd_map the 1-D grid is decoded so that all column
  // chunks of a matrix run on the same XCD (matrix b on XCD b mod 8): a diamond is then fetched over the fabric
  // once per XCD-round instead of once per XCD and drifts less out of the L2.
  int mat, chunk;
  if (xcd_map) {
    const int nchunk = (ncols + kNc - 1) / kNc;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    mat = xcd + 8 * (slot / nchunk);
    chunk = slot % nchunk;
    if (mat >= batch) return;
  } else {
    mat = blockIdx.y;
    chunk = blockIdx.x;
  }
  const double* sb = sb_all + (size_t)mat * SL.slab;
  double* Z = z_all + (size_t)mat * stride_z;
  const int j0 = chunk * kNc;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;

  // accumulators: zt[h][ni][r] <-> Z(row = win + 64 h + 16 w + fr, col = j0 + 16 ni + fk + 4 r)
  d4 zt[2][2];
  double b1[20], b2[2][16];

  auto load_half = [&](int h, int row0) {   // rows row0 + 16 w + fr
    const int row = row0 + 16 * w + fr;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = j0 + 16 * ni + fk + 4 * r;
        // unconditional load from a clamped address, masked afterwards
        const double v = Z[(size_t)std::min(col, ncols - 1) * n + std::min(row, n - 1)];
        zt[h][ni][r] = (row < n && col < ncols) ? v : 0.0;
      }
  };
  auto store_half = [&](int h, int row0) {
    const int row = row0 + 16 * w + fr;
    if (row >= n) return;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = j0 + 16 * ni + fk + 4 * r;
        if (col < ncols) Z[(size_t)col * n + row] = zt[h][ni][r];
      }
  };
  auto copy_half_to_lds = [&](int h, int phys) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) Zs[phys][(16 * ni + fk + 4 * r) * LDH + 16 * w + fr] = zt[h][ni][r];
  };
  // The diamond is a parallelogram: sweep c is non-zero in rows c .. c + 63 only.  Wave w (sweeps 16 w .. 16 w + 15)
  // therefore only needs rows 16 w .. 16 w + 79 of VD in the first product (20 of the 32 k-steps), and in the second
  // product the rows 64 + 16 w .. of VT are zero for sweeps < 16 w (VT[r, c] = sum_l V[r, l] T[l, c], l >= r - 63).
  auto fetch_b1 = [&](size_t dia) {   // VD[row 16 w + 4 kk + fk][sweep 16 w + fr], kk < 20
    const double* vd = sb + SL.vd + dia * kDiaSize + 16 * w + fr + (size_t)(16 * w + fk) * kG;
#pragma unroll
    for (int kk = 0; kk < 20; ++kk) b1[kk] = vd[(size_t)kk * 4 * kG];
  };
  auto fetch_b2 = [&](size_t dia) {   // VT[row 64 h + 16 w + fr][sweep 4 kk + fk]
    const double* vt = sb + SL.vt2 + dia * kDiaSize + 16 * w + fr + (size_t)fk * kDiaLd;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) b2[h][kk] = vt[(size_t)kk * 4 * kDiaLd + 64 * h];
  };

  for (int S = SL.ngroups - 1; S >= 0; --S) {
    const int d0 = dia_off[S], nk = dia_off[S + 1] - d0;
    int win = S * kG + 1;          // first row of the window
    int par = 0;                   // physical LDS buffer of the window's first half
    load_half(0, win);
    load_half(1, win + 64);
    fetch_b1((size_t)d0);
    fetch_b2((size_t)d0);
    copy_half_to_lds(0, 0);
    copy_half_to_lds(1, 1);
    for (int k = 0; k < nk; ++k) {
      __syncthreads();   // window copy complete
      // ---- W1^T = Z^T VD
      d4 c1[2] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
#pragma unroll
      for (int kk = 0; kk < 20; ++kk) {
        const int kr = 4 * w + kk;                 // k-step (4 rows each) within the 128-row window
        const double* zs = Zs[(kr >> 4) ^ par];
        const int rr = (kr & 15) * 4 + fk;
        const double a0 = zs[fr * LDH + rr], a1 = zs[(16 + fr) * LDH + rr];
        c1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1[kk], c1[0], 0, 0, 0);
        c1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[kk], c1[1], 0, 0, 0);
      }
      if (k + 1 < nk) fetch_b1((size_t)d0 + k + 1);   // consumed: refill for the next diamond
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) W1s[(16 * w + fr) * LDW + 16 * ni + fk + 4 * r] = -c1[ni][r];
      __syncthreads();   // W1 complete; nobody reads the window copy any more
      // the 64 rows that enter the window next are not touched by this diamond: fetch them now, behind the MFMAs
      d4 zn[2] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
      if (k + 1 < nk) {
        const int row = win + 128 + 16 * w + fr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = j0 + 16 * ni + fk + 4 * r;
            const double v = Z[(size_t)std::min(col, ncols - 1) * n + std::min(row, n - 1)];
            zn[ni][r] = (row < n && col < ncols) ? v : 0.0;
          }
      }
      // ---- Z^T -= W1^T VT^T
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const double a0 = W1s[(4 * kk + fk) * LDW + fr], a1 = W1s[(4 * kk + fk) * LDW + 16 + fr];
        zt[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b2[0][kk], zt[0][0], 0, 0, 0);
        zt[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b2[0][kk], zt[0][1], 0, 0, 0);
        if (kk >= 4 * w) {   // wave-uniform
          zt[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b2[1][kk], zt[1][0], 0, 0, 0);
          zt[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b2[1][kk], zt[1][1], 0, 0, 0);
        }
      }
      if (k + 1 < nk) fetch_b2((size_t)d0 + k + 1);
      // ---- slide: the first half is finished
      store_half(0, win);
      if (k + 1 < nk) {
        zt[0][0] = zt[1][0];
        zt[0][1] = zt[1][1];
        copy_half_to_lds(0, par ^ 1);       // the old second half (updated) stays where it is, as the new first half
        win += 64;
        zt[1][0] = zn[0];
        zt[1][1] = zn[1];
        copy_half_to_lds(1, par);           // new rows go where the finished half was
        par ^= 1;
      } else {
        store_half(1, win + 64);
      }
    }
    __syncthreads();   // the next group rewrites both window buffers
  }
}

}  // namespace

// ================================================================================================================
// Layout
size_t sb_slab_doubles(int n, int ncols, SbLayout* out) {
  SbLayout L{};
  L.n = n;
  long long off = 0;
  auto take = [&](long long cnt) { long long o = off; off += (cnt + 31) / 32 * 32; return o; };
  const int nchunk = (n + kQrRows - 1) / kQrRows + 1;
  L.vw = take((long long)n * 2 * kB);
  L.wv = take((long long)n * 2 * kB);
  L.xv = take((long long)n * 3 * kB);
  L.qrpart = take((long long)2 * nchunk * kB);
  L.qrpiv = take(2 * (kB + 8));
  L.qrpart8 = take((long long)nchunk * 8 * kB);
  L.small = take((long long)kSmallSplit * kB * 3 * kB);
  L.cmat = take(3 * kB * kB);
  L.ab = take((long long)kLdab * n);
  // diamonds
  const int nsweep = std::max(n - 2, 0);
  L.ngroups = (nsweep + kG - 1) / kG;
  long long ndia = 0;
  for (int S = 0; S < L.ngroups; ++S) {
    const int len = (n - 1 - S * kG + kB - 1) / kB;
    ndia += len;
  }
  L.ndia = ndia;
  L.vd = take(ndia * kDiaSize);
  L.vt2 = take(ndia * kDiaSize);
  L.tau2 = take(ndia * kG);
  (void)ncols;
  L.slab = off;
  if (out) *out = L;
  return (size_t)off;
}

int sb_desc_count(int n, int batch) {
  const int npanels = n / kB + 1;
  return npanels * 6 * batch;
}


// Diamond offsets per sweep group (shared by all matrices of the batch): host copy, (ngroups + 1) ints.
static std::vector<int> dia_offsets(int n) {
  const int nsweep = std::max(n - 2, 0);
  const int ng = (nsweep + kG - 1) / kG;
  std::vector<int> off((size_t)ng + 1, 0);
  for (int S = 0; S < ng; ++S) off[(size_t)S + 1] = off[(size_t)S] + (n - 1 - S * kG + kB - 1) / kB;
  return off;
}

// ================================================================================================================
// Stage 1 + 2 driver.  d_a: lower triangle valid (after mirror_lower_batched).  On exit: tri slab holds d, e and the
// stage-1 tau; A holds the stage-1 reflectors (cleaned for the back-transformation); the sb slab holds the diamonds.
int sytrd_2stage_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, double* d_tri_ws,
                         const TriLayout& TL, double* d_sb_ws, const SbLayout& SL, int* d_dia_off,
                         GemmDesc* d_descs, float* ms_stage1, float* ms_stage2, double* d_band_copy) {
  hipStream_t st = ctx->stream;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  const bool prof = ctx->profiling && ms_stage1 && ms_stage2;
  if (prof) {
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    SC_HIP(ctx, hipEventRecord(ev[0], st));
  }

  // ---- descriptors of stage 1: per panel [symm x 2 | vtx | w x 2 | syr2k] x batch
  int npanels = 0;
  while (n - (npanels + 1) * kB >= 2) ++npanels;
  std::vector<GemmDesc> h((size_t)npanels * 6 * batch);
  for (int p = 0; p < npanels; ++p) {
    const int j0 = p * kB, r0 = j0 + kB, m = n - r0;
    for (int b = 0; b < batch; ++b) {
      double* A = d_a + (size_t)b * stride_a;
      double* sb = d_sb_ws + (size_t)b * SL.slab;
      double* a22 = A + (size_t)r0 * n + r0;
      GemmDesc* g = &h[((size_t)p * batch + b) * 6];
      // X1 = L V
      GemmDesc X1{};
      X1.a = a22; X1.sa_i = 1; X1.sa_k = n; X1.a_tri = 1;
      X1.b = sb + SL.xv + (size_t)2 * kB * n + r0; X1.sb_k = 1; X1.sb_j = n;
      X1.c = sb + SL.xv + r0; X1.ldc = n;
      X1.m = m; X1.n = kB; X1.k = m; X1.alpha = 1.0; X1.beta = 0.0;
      g[0] = X1;
      // X2 = strict(L)^T V
      GemmDesc X2 = X1;
      X2.sa_i = n; X2.sa_k = 1; X2.a_tri = 2;
      X2.c = sb + SL.xv + (size_t)kB * n + r0;
      g[1] = X2;
      // V^T [X1 | X2 | V], split-K slices
      GemmDesc P{};
      P.a = sb + SL.xv + (size_t)2 * kB * n + r0; P.sa_i = n; P.sa_k = 1;
      P.b = sb + SL.xv + r0; P.sb_k = 1; P.sb_j = n;
      P.c = sb + SL.small; P.ldc = kB;
      P.m = kB; P.n = 3 * kB; P.k = m; P.alpha = 1.0; P.beta = 0.0;
      P.split_stride = (long long)kB * 3 * kB;
      g[2] = P;
      // W = [X1 | X2 | V] C  -> [V|W] second half and [W|V] first half
      GemmDesc W{};
      W.a = sb + SL.xv + r0; W.sa_i = 1; W.sa_k = n;
      W.b = sb + SL.cmat; W.sb_k = 1; W.sb_j = 3 * kB;
      W.c = sb + SL.vw + (size_t)kB * n + r0; W.ldc = n;
      W.m = m; W.n = kB; W.k = 3 * kB; W.alpha = 1.0; W.beta = 0.0;
      g[3] = W;
      W.c = sb + SL.wv + r0;
      g[4] = W;
      // A22 -= [V|W] [W|V]^T, lower triangle
      GemmDesc R{};
      R.a = sb + SL.vw + r0; R.sa_i = 1; R.sa_k = n;
      R.b = sb + SL.wv + r0; R.sb_k = n; R.sb_j = 1;
      R.c = a22; R.ldc = n;
      R.m = m; R.n = m; R.k = 2 * kB; R.alpha = -1.0; R.beta = 1.0;
      R.lower_only = 1;
      g[5] = R;
    }
  }
  // regroup so that each launch's records are contiguous: [panel][kind][batch]
  std::vector<GemmDesc> hs(h.size());
  for (int p = 0; p < npanels; ++p)
    for (int b = 0; b < batch; ++b) {
      const GemmDesc* g = &h[((size_t)p * batch + b) * 6];
      GemmDesc* o = &hs[(size_t)p * 6 * batch];
      o[0 * batch + b] = g[0];              // symm: 2 * batch records: [X1 x batch | X2 x batch]
      o[1 * batch + b] = g[1];
      o[2 * batch + b] = g[2];
      o[3 * batch + b] = g[3];              // w: 2 * batch records
      o[4 * batch + b] = g[4];
      o[5 * batch + b] = g[5];
    }
  if (!hs.empty())
    SC_HIP(ctx, hipMemcpyAsync(d_descs, hs.data(), hs.size() * sizeof(GemmDesc), hipMemcpyHostToDevice, st));
  const std::vector<int> doff = dia_offsets(n);
  SC_HIP(ctx, hipMemcpyAsync(d_dia_off, doff.data(), doff.size() * sizeof(int), hipMemcpyHostToDevice, st));

  // tau of columns without a reflector must read 0
  for (int b = 0; b < batch; ++b)
    SC_HIP(ctx, hipMemsetAsync(d_tri_ws + (size_t)b * TL.slab + TL.tau, 0, sizeof(double) * n, st));

  const size_t lds_qr = sizeof(double) * ((size_t)kB * (kQrRows + 1) + kB + kQrRows + 4 * kB);
  const size_t lds_small = sizeof(double) * 4 * kB * (kB + 1);
  const size_t lds_blk_a = sizeof(double) * ((size_t)kB * (kQrRows + 1) + kQrRows + 8 * kB);
  const size_t lds_blk_b = sizeof(double) * ((size_t)kB * (kQrRows + 1) + kIb * kB + 4 * kB);
  static const bool blocked_qr = getenv("SPRINGCRAFT_QR_UNBLOCKED") == nullptr;
  for (int p = 0; p < npanels; ++p) {
    const int j0 = p * kB, r0 = j0 + kB, m = n - r0;
    const int nr = std::min(kB, m - 1);
    const int nchunks = (m + kQrRows - 1) / kQrRows;
    const dim3 qgrid((unsigned)nchunks, (unsigned)batch);
    if (nr == kB && blocked_qr) {
      // blocked panel: inner blocks of 8 columns, their reflectors applied to the rest of the panel at once
      hipLaunchKernelGGL(k_panel_qr, qgrid, dim3(256), lds_qr, st, d_a, stride_a, d_tri_ws, TL, d_sb_ws, SL, j0, 0, nr,
                         kIb);
      for (int c0 = 0; c0 < kB; c0 += kIb) {
        for (int j = c0 + 1; j < c0 + kIb; ++j)
          hipLaunchKernelGGL(k_panel_qr, qgrid, dim3(256), lds_qr, st, d_a, stride_a, d_tri_ws, TL, d_sb_ws, SL, j0, j,
                             nr, c0 + kIb);
        hipLaunchKernelGGL(k_pqr_blk_a, qgrid, dim3(256), lds_blk_a, st, d_a, stride_a, d_tri_ws, TL, d_sb_ws, SL, j0, c0);
        if (c0 + kIb < kB)
          hipLaunchKernelGGL(k_pqr_blk_b, qgrid, dim3(256), lds_blk_b, st, d_a, stride_a, d_tri_ws, TL, d_sb_ws, SL, j0,
                             c0);
      }
    } else {
      for (int j = 0; j <= nr; ++j)
        hipLaunchKernelGGL(k_panel_qr, qgrid, dim3(256), lds_qr, st, d_a, stride_a, d_tri_ws, TL, d_sb_ws, SL, j0, j, nr,
                           kB);
    }
    const GemmDesc* g = d_descs + (size_t)p * 6 * batch;
    SC_TRY(launch_gemm_f64(ctx, g, batch, m, kB, kGemmTile, 1, false, true, kGemmAmBk));           // X1 = L V
    SC_TRY(launch_gemm_f64(ctx, g + batch, batch, m, kB, kGemmTile, 1, false, true, kGemmAkBk));   // X2 = strict(L)^T V
    SC_TRY(launch_gemm_f64(ctx, g + 2 * batch, batch, kB, 3 * kB, kGemmTile, kSmallSplit, false, false, kGemmAkBk));
    hipLaunchKernelGGL(k_sb_small, dim3((unsigned)batch), dim3(256), lds_small, st, d_tri_ws, TL, d_sb_ws, SL, j0);
    SC_TRY(launch_gemm_f64(ctx, g + 3 * batch, 2 * batch, m, kB, kGemmTile, 1, false, false, kGemmAmBk));
    SC_TRY(launch_gemm_f64(ctx, g + 5 * batch, batch, m, m, kGemmTile, 1, false, false, kGemmAmBn));
  }
  SC_HIP(ctx, hipGetLastError());
  if (prof) SC_HIP(ctx, hipEventRecord(ev[1], st));

  // ---- stage 2
  hipLaunchKernelGGL(k_band_extract, dim3(256, (unsigned)batch), dim3(256), 0, st, d_a, stride_a, d_sb_ws, SL);
  if (d_band_copy)   // debugging aid: the band as stage 1 left it (first matrix)
    SC_HIP(ctx, hipMemcpyAsync(d_band_copy, d_sb_ws + SL.ab, sizeof(double) * kLdab * n, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_sb_clean, dim3((unsigned)n, (unsigned)batch), dim3(256), 0, st, d_a, stride_a, n);
  for (int b = 0; b < batch; ++b) {
    double* sb = d_sb_ws + (size_t)b * SL.slab;
    SC_HIP(ctx, hipMemsetAsync(sb + SL.vd, 0, sizeof(double) * (size_t)SL.ndia * kDiaSize, st));
    SC_HIP(ctx, hipMemsetAsync(sb + SL.tau2, 0, sizeof(double) * (size_t)SL.ndia * kG, st));
  }
  if (n >= 3) {
    const int t_max = 2 * (n - 3) + chase_len(n, n - 3) - 1;
    const int gx = chase_len(n, 0) / 2 + 1;
    for (int t = 0; t <= t_max; ++t)
      hipLaunchKernelGGL(k_bulge_step, dim3((unsigned)gx, (unsigned)batch), dim3(256), 0, st, d_sb_ws, SL, t);
  }
  hipLaunchKernelGGL(k_band_to_tri, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, st, d_sb_ws, SL,
                     d_tri_ws, TL);
  SC_HIP(ctx, hipGetLastError());
  if (prof) {
    SC_HIP(ctx, hipEventRecord(ev[2], st));
    SC_HIP(ctx, hipEventSynchronize(ev[2]));
    SC_HIP(ctx, hipEventElapsedTime(ms_stage1, ev[0], ev[1]));
    SC_HIP(ctx, hipEventElapsedTime(ms_stage2, ev[1], ev[2]));
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  SC_HIP(ctx, hipStreamSynchronize(st));   // host descriptor vectors must outlive their uploads
  return SC_OK;
}

// ================================================================================================================
// T factors and V T of all diamonds, on `st` (they only depend on the bulge chase, not on Z).
int bt2_prepare(sc_ctx* ctx, int n, int batch, double* d_sb_ws, const SbLayout& SL, hipStream_t st) {
  if (n < 3 || SL.ndia == 0) return SC_OK;
  for (long long d0 = 0; d0 < SL.ndia; d0 += 32768) {
    const unsigned cnt = (unsigned)std::min<long long>(32768, SL.ndia - d0);
    hipLaunchKernelGGL(k_dia_tfactor, dim3(cnt, (unsigned)batch), dim3(256), 0, st, d_sb_ws, SL, (int)d0);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// Z <- Q2 Z (after bt2_prepare).  d_z: (batch) ncols columns of length n (ld n).
int bt2_batched(sc_ctx* ctx, int n, int batch, double* d_sb_ws, const SbLayout& SL, const int* d_dia_off, double* d_z,
                long long stride_z, int ncols, float* ms_fused) {
  hipStream_t st = ctx->stream;
  if (n < 3 || SL.ndia == 0 || ncols <= 0) return SC_OK;
  hipEvent_t ev[2] = {nullptr, nullptr};
  const bool prof = ctx->profiling && ms_fused;
  if (prof) {
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    SC_HIP(ctx, hipEventRecord(ev[0], st));
  }
  const int nchunk = (ncols + kNc - 1) / kNc;
  if (batch >= 8) {
    const int per_xcd = (batch + 7) / 8;   // matrices per XCD
    hipLaunchKernelGGL(k_bt2_fused, dim3((unsigned)(8 * per_xcd * nchunk)), dim3(256), 0, st, d_sb_ws, SL, d_dia_off,
                       d_z, stride_z, ncols, batch, 1);
  } else {
    hipLaunchKernelGGL(k_bt2_fused, dim3((unsigned)nchunk, (unsigned)batch), dim3(256), 0, st, d_sb_ws, SL, d_dia_off,
                       d_z, stride_z, ncols, batch, 0);
  }
  SC_HIP(ctx, hipGetLastError());
  if (prof) {
    SC_HIP(ctx, hipEventRecord(ev[1], st));
    SC_HIP(ctx, hipEventSynchronize(ev[1]));
    SC_HIP(ctx, hipEventElapsedTime(ms_fused, ev[0], ev[1]));
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  return SC_OK;
}

int sb_band_width() { return kB; }

// ---- debugging entry point (not part of the public C ABI): band after stage 1 (128 x n, AB(i,j) at [(i-j) + 128 j])
// and the tridiagonal after stage 2 of ONE host matrix (NumPy layout, lower triangle read).
extern "C" int sc_dbg_two_stage(sc_ctx* ctx, const double* a, int n, double* band_out, double* d_out, double* e_out) {
  if (!ctx || !a || n < 4 * kB) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  TriLayout TL{};
  TL.n = n; TL.nb = kB;
  TL.d = 0; TL.e = n; TL.tau = 2 * (long long)n; TL.slab = 3 * (long long)n + 64;
  SbLayout SL;
  const size_t sbd = sb_slab_doubles(n, 0, &SL);
  double *d_a = nullptr, *d_tri = nullptr, *d_sb = nullptr, *d_band = nullptr;
  int* d_off = nullptr;
  GemmDesc* d_desc = nullptr;
  SC_HIP(ctx, hipMalloc(&d_a, sizeof(double) * (size_t)n * n));
  SC_HIP(ctx, hipMalloc(&d_tri, sizeof(double) * TL.slab));
  SC_HIP(ctx, hipMalloc(&d_sb, sizeof(double) * sbd));
  SC_HIP(ctx, hipMalloc(&d_band, sizeof(double) * kLdab * n));
  SC_HIP(ctx, hipMalloc(&d_off, sizeof(int) * (n / 64 + 8)));
  SC_HIP(ctx, hipMalloc(&d_desc, sizeof(GemmDesc) * sb_desc_count(n, 1)));
  SC_HIP(ctx, hipMemcpyAsync(d_a, a, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, ctx->stream));
  int rc = mirror_lower_batched(ctx, d_a, (long long)n * n, n, 1);
  if (rc == SC_OK)
    rc = sytrd_2stage_batched(ctx, d_a, (long long)n * n, n, 1, d_tri, TL, d_sb, SL, d_off, d_desc, nullptr, nullptr,
                              d_band);
  if (rc == SC_OK) {
    (void)hipMemcpy(band_out, d_band, sizeof(double) * kLdab * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(d_out, d_tri + TL.d, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(e_out, d_tri + TL.e, sizeof(double) * n, hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_a); (void)hipFree(d_tri); (void)hipFree(d_sb); (void)hipFree(d_band); (void)hipFree(d_off);
  (void)hipFree(d_desc);
  return rc;
}
