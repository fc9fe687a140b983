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
#include <type_traits>
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
constexpr int kDescKinds = 9;     // GEMM records per panel and matrix (stage 1)
#ifndef SC_QR_IB
#define SC_QR_IB 8
#endif
constexpr int kIb = SC_QR_IB;     // inner block of the blocked panel QR (columns whose reflectors are applied to the rest at once)
constexpr int kEarly = (kIb + 2) / 2;   // column loads per thread that cover a launch of an inner block (kIb + 1 columns)

typedef int v4i __attribute__((ext_vector_type(4)));   // a 16-byte record (early hand-off of the chase, the cooperative panel)

struct HH {
  double beta, tau, scale;
};

// LAPACK dlarfg scalars: x = (alpha, tail), xn2 = ||tail||^2;  H x = beta e1,  v = (1, scale * tail)
__device__ __forceinline__ HH householder(double alpha, double xn2) {
  HH h;
  // No reflection either when the column (pivot included) is below 1e-140: its squares are in the underflow range, where
  // a norm is not a norm any more (LAPACK's dlarfg rescales there) and the reflector would come out non-orthogonal.  The
  // input is scaled to [1e-100, 1e100] (matrix_scale_factor), so such a column is < 1e-40 of the matrix: the caller
  // stores zeros for its tail, a backward error far below rounding.  Seen with exactly rank-deficient input such as
  // ones(n, n), whose trailing matrices shrink by a factor eps per column.
  if (xn2 == 0.0 || alpha * alpha + xn2 < 1e-280) {
    h.beta = alpha; h.tau = 0.0; h.scale = 0.0;
    return h;
  }
  h.beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
  h.tau = (h.beta - alpha) / h.beta;
  h.scale = 1.0 / (alpha - h.beta);
  return h;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. waits for
// the acknowledgement of every global store in flight (about 2 us under load) -- wasted when no thread of the workgroup
// reads global data that another one wrote in the same kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---- XCDs of the device, counted by a probe launch (once per context): every workgroup ORs the bit of the XCC_ID it
// runs on into a mask.  8 x CUs workgroups of one wave: the dispatcher deals consecutive workgroups round-robin over
// the XCDs, so every XCD that exists is seen.  (num_cus / 32 holds for MI350 / MI355 and their partitions only.)
__global__ void k_xcd_probe(unsigned* mask) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) atomicOr(mask, 1u << (id & 15u));
}
int probe_xcd_count(sc_ctx* ctx, hipStream_t st) {
  SC_TRY(sc_reserve_dc_aux(ctx, 256));
  unsigned* d_mask = reinterpret_cast<unsigned*>(ctx->dc_aux);
  SC_HIP(ctx, hipMemsetAsync(d_mask, 0, sizeof(unsigned), st));
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  hipLaunchKernelGGL(k_xcd_probe, dim3((unsigned)(8 * cus)), dim3(64), 0, st, d_mask);
  SC_HIP(ctx, hipGetLastError());
  unsigned h_mask = 0;
  SC_HIP(ctx, hipMemcpyAsync(&h_mask, d_mask, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  SC_HIP(ctx, hipStreamSynchronize(st));
  int count = 0, top = 0;
  for (int b = 0; b < 16; ++b)
    if (h_mask >> b & 1u) { ++count; top = b + 1; }
  // the kernels bind matrix b to XCD b mod nxcd by XCC_ID value: ids must be 0 .. count - 1 (they are on every part seen)
  ctx->nxcd = (count >= 1 && count == top) ? std::min(count, 8) : 1;
  return SC_OK;
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
  // only the columns this launch touches, c_lo .. c_end-1, are held (the image is indexed relative to c_lo): with the
  // blocked panels that is at most 9 columns = 9 KB instead of 66 KB, i.e. 8 workgroups per CU instead of 2
  const int c_lo = j > 0 ? j - 1 : 0;
  const int ncl = c_end - c_lo;
  double* P = sm;                      // [ncl][LD]   column c at P[(c - c_lo) * LD + r]
  double* wv = sm + ncl * LD;          // [kB]  w_c of the reflector being applied
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
  // launch j reads what launch j-1 left and writes for launch j+1 while other workgroups may still be reading:
  // two copies, alternating
  const int nchunk_cap = (n + kQrRows - 1) / kQrRows + 1;
  const double* part_in = sb + SL.qrpart + (size_t)((j + 1) & 1) * nchunk_cap * kB;
  double* part_out = sb + SL.qrpart + (size_t)(j & 1) * nchunk_cap * kB;
  const double* piv_in = sb + SL.qrpiv + (size_t)((j + 1) & 1) * (kB + 8);
  double* piv_out = sb + SL.qrpiv + (size_t)(j & 1) * (kB + 8);

  // (1a) chunk -> registers, requested BEFORE the partial results of the previous launch are read and reduced (two
  // dependent memory round trips become one; a launch of a blocked panel holds at most 9 columns = one pass)
  const int ld_r = tid & (kQrRows - 1), ld_half = tid >> 7;
  const int ld_rl = row_base + ld_r;
  const double* ld_src = A + (size_t)j0 * n + r0 + std::min(ld_rl, m - 1);
  const bool early = ncl <= 2 * kEarly;
  double t_early[kEarly];
  if (early) {
#pragma unroll
    for (int u = 0; u < kEarly; ++u) t_early[u] = ld_src[(size_t)std::min(c_lo + ld_half + 2 * u, kB - 1) * n];
  }

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
  if (early) {
#pragma unroll
    for (int u = 0; u < kEarly; ++u) {
      const int c = c_lo + ld_half + 2 * u;
      if (c < c_end) P[(c - c_lo) * LD + ld_r] = ld_rl < m ? t_early[u] : 0.0;
    }
  } else {
    const int r = ld_r, half = ld_half, rl = ld_rl;
    const double* src = ld_src;
    for (int c = c_lo + half; c < c_end; c += 16) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = src[(size_t)std::min(c + 2 * u, kB - 1) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + 2 * u < c_end) P[((c + 2 * u) - c_lo) * LD + r] = rl < m ? t[u] : 0.0;
    }
  }
  __syncthreads();

  const int c = tid & 63, q = tid >> 6;   // column / row quarter (32 rows) of this thread in the compute phases
  if (j >= 1) {
    // (a1) v
    if (tid < kQrRows) {
      const int rl = row_base + tid;
      double v = 0.0;
      if (rl < m) v = rl > prev ? s_scale * P[(prev - c_lo) * LD + tid] : (rl == prev ? 1.0 : 0.0);
      vv[tid] = v;
    }
    __syncthreads();
    // (a2) P[:, c] -= v w_c
    if (c >= j && c < c_end) {
      const double w = wv[c];
#pragma unroll 8
      for (int r = q * 32; r < q * 32 + 32; ++r) P[(c - c_lo) * LD + r] -= vv[r] * w;
    }
    // column prev: v below the pivot, beta at it (rows above keep their R entries); and the panel buffers
    if (tid < kQrRows) {
      const int rl = row_base + tid;
      if (rl < m) {
        const double v = vv[tid];
        if (rl > prev) P[(prev - c_lo) * LD + tid] = v;
        else if (rl == prev) P[(prev - c_lo) * LD + tid] = s_beta;
        const size_t row = (size_t)r0 + rl;
        sb[SL.vw + (size_t)prev * n + row] = v;
        sb[SL.wv + (size_t)(kB + prev) * n + row] = -v;   // (the [W|V] panel holds -[W|V]: see the trailing update's record)
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
        if (rl > j && rl < m) acc += P[(j - c_lo) * LD + r] * P[(c - c_lo) * LD + r];
      }
    }
    red[q * kB + c] = acc;
    __syncthreads();
    if (tid < kB) {
      part_out[(size_t)chunk * kB + tid] = (red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]);
      if (chunk == 0 && tid >= j && tid < c_end) piv_out[tid] = P[(tid - c_lo) * LD + j];   // pivot row (alpha at [j])
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
      for (int cc = c_lo + half; cc < c_end; cc += 2) A[(size_t)(j0 + cc) * n + r0 + rl] = P[(cc - c_lo) * LD + r];
  }
}

// ---- blocked panels: the 8 reflectors of an inner block [c0, c0+8) applied to the columns to their right at once ----
// k_pqr_blk_a: finishes reflector c0+7 (the inner block's last) and forms, per 128-row chunk, the partial products
//   M[i][c] = v_{c0+i} . P[:, c]  for c = c0 .. kB-1  (the first 8 columns are the Gram matrix of the block's reflectors).
// k_pqr_blk_b: sums them, builds the 8 x 8 T factor, W = T^T M, updates P[:, c] -= V W for c >= c0+8 and leaves the
//   tail Gram row / pivot row of column c0+8 for the next inner block's first column launch.

// explicit form of the inner block's reflectors in the LDS copy of chunk 0 (memory keeps R above the pivots)
__device__ __forceinline__ void blk_explicit_v(double* P, int LD, int c0, int ncols, int row_base) {
  if (row_base != 0) return;   // pivot rows are local rows c0 .. c0+7 of the first chunk
  for (int idx = threadIdx.x; idx < ncols * kB; idx += 256) {
    const int i = idx / kB, r = idx % kB;   // rows 0..63 suffice (pivots < 64)
    const int piv = c0 + i;
    if (r < piv) P[i * LD + r] = 0.0;          // (the LDS image starts at column c0)
    else if (r == piv) P[i * LD + r] = 1.0;
  }
}

__global__ __launch_bounds__(256) void k_pqr_blk_a(double* __restrict__ a_all, long long stride_a,
                                                   double* __restrict__ tri_all, TriLayout TL,
                                                   double* __restrict__ sb_all, SbLayout SL, int j0, int c0) {
  constexpr int LD = kQrRows + 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* P = sm;                      // [kB - c0][LD]: columns c0 .. kB-1, indexed relative to c0
  double* vv = sm + (kB - c0) * LD;    // [kQrRows]
  double* red = vv + kQrRows;          // [4][2][kB]
  __shared__ double s_scale, s_beta;
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
      s_scale = h.scale; s_beta = h.beta;
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
        if (c + 2 * u < kB) P[((c + 2 * u) - c0) * LD + r] = rl < m ? t[u] : 0.0;
    }
  }
  __syncthreads();
  // v of reflector prev: to memory (v below the pivot, beta at it), to the panel buffers, explicit form in LDS
  if (tid < kQrRows) {
    const int rl = row_base + tid;
    double v = 0.0;
    if (rl < m) v = rl > prev ? s_scale * P[(prev - c0) * LD + tid] : (rl == prev ? 1.0 : 0.0);
    if (rl < m) {
      if (rl > prev) A[(size_t)(j0 + prev) * n + r0 + rl] = v;
      else if (rl == prev) A[(size_t)(j0 + prev) * n + r0 + rl] = s_beta;
      const size_t row = (size_t)r0 + rl;
      sb[SL.vw + (size_t)prev * n + row] = v;
      sb[SL.wv + (size_t)(kB + prev) * n + row] = -v;   // (the [W|V] panel holds -[W|V]: see the trailing update's record)
      sb[SL.xv + (size_t)(2 * kB + prev) * n + row] = v;
    }
    P[(prev - c0) * LD + tid] = v;
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
      const double x = P[(c - c0) * LD + r];
#pragma unroll
      for (int i = 0; i < kIb; ++i) acc[i] += P[((c0 + i) - c0) * LD + r] * x;
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
  double* P = sm;                      // [kB - c0][LD]: columns c0 .. kB-1, indexed relative to c0
  double* Ms = sm + (kB - c0) * LD;    // [kIb][kB]  M, later W
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
        if (c + 2 * u < kB) P[((c + 2 * u) - c0) * LD + r] = rl < m ? t[u] : 0.0;
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
      for (int i = 0; i < kIb; ++i) s2 += P[((c0 + i) - c0) * LD + r] * wcol[i];
      P[(c - c0) * LD + r] -= s2;
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
        if (rl > jn && rl < m) acc += P[(jn - c0) * LD + r] * P[(c - c0) * LD + r];
      }
    }
    red[q * kB + c] = acc;
    __syncthreads();
    if (tid < kB) {
      part_out[(size_t)chunk * kB + tid] = (red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]);
      if (chunk == 0 && tid >= jn && tid < jn + kIb) piv_out[tid] = P[(tid - c0) * LD + jn];
    }
  }
  // LDS -> chunk (columns to the right of the inner block)
  {
    const int r = tid & (kQrRows - 1), half = tid >> 7;
    const int rl = row_base + r;
    if (rl < m)
      for (int cc = jn + half; cc < kB; cc += 2) A[(size_t)(j0 + cc) * n + r0 + rl] = P[(cc - c0) * LD + r];
  }
}

// ---- the whole panel QR in ONE workgroup per matrix (panels of at most 1024 RU rows) -----------------------------------
// The launches above exist because a column's reflector needs sums over all rows, i.e. over all 128-row chunks: 80
// dependent launches per panel, 10 - 22 us each, the longest item of a latency-bound solve (C4: 4700 launches per
// step).  Here 1024 threads own the rows of the panel (thread t: rows t, t + 1024, ...), the 8 columns of an inner block
// live in registers, and every sum over the rows is a wave reduction + 16 partials in LDS + two workgroup barriers
// (~1 us instead of a launch).  Same arithmetic as the launches above (tail Gram row + pivot row per column, the inner
// block's reflectors applied to the rest of the panel as one block update), other reduction trees.
// One workgroup streams its panel at the rate of one CU: the path is for batches (the matrices run side by side) and
// panels of at most 4096 rows; larger panels and single large matrices keep the chunked launches.
constexpr int kWgThreads = 1024, kWgWaves = kWgThreads / 64;

// Sums of eight values over the 64 lanes with 10 exchanges instead of 48: three halving steps (a lane keeps half of its
// values and receives the partner's sums of those), then three plain steps.  The exchanges inside a row of 16 lanes (eight of the ten) are DPP
// moves on the vector ALU; the two across rows go through the LDS crossbar, which all 16 waves of the workgroup share.  Lane l < 8 returns the total of v[wave_reduce8_index(l)].
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ int wave_reduce8_index(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }
__device__ __forceinline__ double wave_reduce8(const double (&v)[8]) {
  const int lane = threadIdx.x & 63;
  double k4[4], k2[2];
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double send = b0 ? v[k] : v[k + 4];
    k4[k] = (b0 ? v[k + 4] : v[k]) + dpp_quad<0xB1>(send);      // quad_perm [1, 0, 3, 2]: lane ^ 1
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double send = b1 ? k4[k] : k4[k + 2];
    k2[k] = (b1 ? k4[k + 2] : k4[k]) + dpp_quad<0x4E>(send);    // quad_perm [2, 3, 0, 1]: lane ^ 2
  }
  const double send = b2 ? k2[0] : k2[1];
  const double from_lo = dpp_quad<0x114>(send), from_hi = dpp_quad<0x104>(send);   // row_shr:4 (lane - 4), row_shl:4 (lane + 4)
  double r = (b2 ? k2[1] : k2[0]) + (b2 ? from_lo : from_hi);                      // lane ^ 4
  r += dpp_quad<0x128>(r);                                                         // row_ror:8 = lane ^ 8 inside a row of 16
  r += __shfl_xor(r, 16);
  r += __shfl_xor(r, 32);
  return r;
}

template <int RU, int CU, int NT = 1024>
__global__ __launch_bounds__(NT) void k_panel_wg(double* __restrict__ a_all, long long stride_a,
                                                   double* __restrict__ tri_all, TriLayout TL,
                                                   double* __restrict__ sb_all, SbLayout SL, int j0) {
  constexpr int kWgThreads = NT, kWgWaves = NT / 64;   // (shadow the file-level constants: NT threads per panel)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* red = sm;                               // [2][kWgWaves][8] wave partials of the per-column sums (by column parity)
  double* piv = red + 2 * kWgWaves * 8;           // [2][8]          pivot row of the inner block
  double* tauL = piv + 16;                        // [8]
  double* Ms = tauL + 8;                          // [8][kB]         M = V^T P, then W = T^T M (column index = panel column)
  double* part = Ms + 8 * kB;                         // [kWgWaves][kB][8] wave partials of M
  const int n = TL.n;
  const int r0 = j0 + kB, m = n - r0;
  double* A = a_all + (size_t)blockIdx.x * stride_a;
  double* tri = tri_all + (size_t)blockIdx.x * TL.slab;
  double* sb = sb_all + (size_t)blockIdx.x * SL.slab;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double* P = A + (size_t)j0 * n + r0;            // P(r, c) = P[c * n + r]
  int rl[RU], rc[RU];                             // my rows, and the same clamped into the panel for the loads
  bool ok[RU];
#pragma unroll
  for (int u = 0; u < RU; ++u) { rl[u] = tid + kWgThreads * u; ok[u] = rl[u] < m; rc[u] = std::min(rl[u], m - 1); }
  // accesses = a column's (uniform, scalar) base + the row's 32-bit byte offset, re-materialised at the access so that
  // the compiler does not hoist RU x 64 per-column vector addresses out of the loops (they would not fit the registers)
  typedef char __attribute__((address_space(1)))* gbp;
  typedef double __attribute__((address_space(1)))* gdp;
  typedef const double __attribute__((address_space(1)))* gdp_c;
  auto off = [&](int row) -> unsigned {
    unsigned e = 8u * (unsigned)row;
    asm volatile("" : "+v"(e));
    return e;
  };
  auto ld = [&](const double* colbase, int row) -> double { return *(gdp_c)((gbp)colbase + off(row)); };
  auto st = [&](double* colbase, int row, double val) { *(gdp)((gbp)colbase + off(row)) = val; };

  for (int c0 = 0; c0 < kB; c0 += 8) {
    // ---- the inner block's columns -> registers
    double x[RU][8];
#pragma unroll
    for (int u = 0; u < RU; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[u][i] = ld(P + (size_t)(c0 + i) * n, rc[u]);
#pragma unroll
    for (int u = 0; u < RU; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[u][i] = ok[u] ? x[u][i] : 0.0;
    // ---- its 8 reflectors
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int j = c0 + jj;                      // pivot = local row j, owned by thread j (u = 0)
      const int pb = jj & 1;
      double g[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        double acc = 0.0;
        if (c >= jj) {
#pragma unroll
          for (int u = 0; u < RU; ++u) acc += rl[u] > j ? x[u][jj] * x[u][c] : 0.0;
        }
        g[c] = acc;
      }
      const double gs = wave_reduce8(g);
      if (lane < 8) red[(pb * kWgWaves + wv) * 8 + wave_reduce8_index(lane)] = gs;
      if (tid == j) {
#pragma unroll
        for (int c = jj; c < 8; ++c) piv[pb * 8 + c] = x[0][c];
      }
      __syncthreads();   // (ONE barrier per column: the partials and the pivot row are double-buffered by its parity)
      // totals: lane c < 8 of every wave sums the 16 partials of value c (same order in every wave), the others take them
      // from that lane through a scalar register
      double tot = 0.0;
      if (lane < 8) {
#pragma unroll
        for (int w2 = 0; w2 < kWgWaves; ++w2) tot += red[(pb * kWgWaves + w2) * 8 + lane];
      }
      double fin[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned long long tb = (unsigned long long)__double_as_longlong(tot);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)tb, c);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(tb >> 32), c);
        fin[c] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
      }
      const HH h = householder(piv[pb * 8 + jj], fin[jj]);
      if (tid == 0) { tri[TL.tau + j0 + j] = h.tau; tauL[jj] = h.tau; }
      double v[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) v[u] = rl[u] > j ? h.scale * x[u][jj] : (rl[u] == j ? 1.0 : 0.0);
#pragma unroll
      for (int c = jj + 1; c < 8; ++c) {
        const double wc = h.tau * (piv[pb * 8 + c] + h.scale * fin[c]);
#pragma unroll
        for (int u = 0; u < RU; ++u) x[u][c] -= v[u] * wc;
      }
      // column j is final: R entries of this inner block above the pivot, beta at it, v below; V in its explicit form
      // stays in the registers and goes to the three panel buffers
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        if (ok[u]) {
          if (rl[u] >= c0) st(P + (size_t)j * n, rl[u], rl[u] > j ? v[u] : (rl[u] == j ? h.beta : x[u][jj]));
          st(sb + SL.vw + (size_t)j * n + r0, rl[u], v[u]);
          st(sb + SL.wv + (size_t)(kB + j) * n + r0, rl[u], -v[u]);   // (-[W|V])
          st(sb + SL.xv + (size_t)(2 * kB + j) * n + r0, rl[u], v[u]);
        }
        x[u][jj] = ok[u] ? v[u] : 0.0;
      }
    }
    const int jn = c0 + 8;                        // first column to the right of the inner block
    if (jn >= kB) break;
    // ---- M[i][c] = v_i . P[:, c]: first the block's own columns (the Gram matrix of its reflectors), then the columns
    // to the right, CU at a time so that their loads are in flight together; wave partials -> LDS, summed below
#pragma unroll
    for (int ci = 0; ci < 8; ++ci) {
      double pr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < RU; ++u) acc += x[u][i] * x[u][ci];
        pr[i] = acc;
      }
      const double ps = wave_reduce8(pr);
      if (lane < 8) part[((size_t)wv * kB + c0 + ci) * 8 + wave_reduce8_index(lane)] = ps;
    }
    for (int cb = jn; cb < kB; cb += CU) {
      double a[CU][RU];
#pragma unroll
      for (int k = 0; k < CU; ++k)
#pragma unroll
        for (int u = 0; u < RU; ++u) a[k][u] = ld(P + (size_t)std::min(cb + k, kB - 1) * n, rc[u]);
#pragma unroll
      for (int k = 0; k < CU; ++k) {
        double pr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          double acc = 0.0;
#pragma unroll
          for (int u = 0; u < RU; ++u) acc += ok[u] ? x[u][i] * a[k][u] : 0.0;
          pr[i] = acc;
        }
        const double ps = wave_reduce8(pr);
        if (lane < 8 && cb + k < kB) part[((size_t)wv * kB + cb + k) * 8 + wave_reduce8_index(lane)] = ps;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 8 * (kB - c0); idx += kWgThreads) {
      const int i = idx & 7, c = c0 + (idx >> 3);
      double acc = 0.0;
#pragma unroll
      for (int w2 = 0; w2 < kWgWaves; ++w2) acc += part[((size_t)w2 * kB + c) * 8 + i];
      Ms[i * kB + c] = acc;
    }
    __syncthreads();
    // W = T^T M for the columns to the right without forming T: (D + striu(G))^T W = M with D = diag(1 / tau) and
    // G[l][i] = Ms[l][c0 + i] the Gram matrix of the block's reflectors, i.e. W[i] = tau_i (M[i] - sum_{l < i} G[l][i] W[l])
    // (rows of tau = 0 reflectors come out zero, as in larft's T); one thread per column, in place
    if (tid >= jn && tid < kB) {
      double wcol[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double acc = Ms[i * kB + tid];
#pragma unroll
        for (int l = 0; l < i; ++l) acc -= Ms[l * kB + c0 + i] * wcol[l];
        wcol[i] = tauL[i] * acc;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) Ms[i * kB + tid] = wcol[i];
    }
    __syncthreads();
    // P[:, c] -= V W[:, c]
    for (int cb = jn; cb < kB; cb += CU) {
      double a[CU][RU];
#pragma unroll
      for (int k = 0; k < CU; ++k)
#pragma unroll
        for (int u = 0; u < RU; ++u) a[k][u] = ld(P + (size_t)std::min(cb + k, kB - 1) * n, rc[u]);
#pragma unroll
      for (int k = 0; k < CU; ++k) {
        const int c = std::min(cb + k, kB - 1);
        double wcol[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wcol[i] = Ms[i * kB + c];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          double s2 = 0.0;
#pragma unroll
          for (int i = 0; i < 8; ++i) s2 += x[u][i] * wcol[i];
          if (ok[u] && rl[u] >= c0 && cb + k < kB) st(P + (size_t)c * n, rl[u], a[k][u] - s2);
        }
      }
    }
    __syncthreads();   // (the next inner block reads what this thread has just written to the same rows; the barrier
                       // separates the reuse of the LDS buffers)
  }
}

// ---- tall panels of a FEW matrices: the panel QR by several workgroups of one launch -------------------------------
// A panel of more than 4096 rows of one matrix (C5: up to 24 000; a single N = 2000 structure: 5 936) went through the
// chunked launches above: 80 dependent launches of ~8-10 us per panel, 0.77 ms per panel whatever its height (C5: 288 ms
// of a 1.45 s solve, a single n = 6000 solve: 52 of 190 ms).  One workgroup cannot take such a panel -- it would stream
// it at the rate of one CU.  Here G = ceil(m / 256) workgroups own 256 rows each, FOR THE WHOLE PANEL: the rows sit in
// LDS (64 columns x 256 rows = 131 KB), loaded once and never re-read from memory, the 8 columns of the inner block in
// registers as in k_panel_wg, and every sum over the rows of the panel is an exchange of 16-byte RECORDS
// {value, sequence number} between the workgroups: written and read as ONE access (global_store / global_load_dwordx4
// with sc1: agent scope, coherent across the XCDs without any fence or cache write-back -- the value and the tag that
// says it is this step's arrive together).  Same arithmetic as k_panel_wg (tail Gram row + pivot row per column, the
// inner block's reflectors applied to the columns on its right as one block update), other reduction trees:
//   * per column: wave partials -> LDS -> the workgroup's 8 sums (+ workgroup 0: the pivot row) as 16 records; every
//     workgroup polls the records of all G workgroups (the four waves a quarter each), sums them in the same order;
//   * per inner block: M = V^T P for the columns on the right (up to 8 x 56 values + the block's 8 x 8 Gram matrix) by
//     threads that own a COLUMN and a quarter of the rows (no cross-lane reduction: the column stride of 257 doubles
//     makes both the row-parallel and the column-parallel LDS access conflict-free); the sum over the workgroups in two
//     hops (value i is summed by workgroup i mod G, which publishes the total), because all-to-all would be G x 8 KB
//     of records per workgroup.
// All G workgroups must be resident together (each waits for all others): G <= 128 and the launch rule keeps
// matrices x G well below the number of CUs; a wait that runs into its bound (never expected: seconds) raises a flag that
// every workgroup sees in its polls, the kernel ends, the host returns an error and the context does not use the kernel
// again.  Sequence numbers are unique within a solve ((panel + 1) * 128 + step) and the records are zeroed per solve.
constexpr int kCoopRows = 256;
constexpr int kCoopLd = kCoopRows + 1;
constexpr int kCoopMaxG = 128;
constexpr int kCoopVals = 8 * kB;                 // values of one block exchange (M and the Gram matrix)
constexpr size_t kCoopLdsBytes = sizeof(double) * ((size_t)kB * kCoopLd + kCoopRows * 8 + 8 * kB + 64 + 16 + kB + 128 + 2);
// workspace of one matrix (16-byte records): column records [2][G][16] | block partials [G][512] | block totals [512]
__host__ __device__ inline size_t coop_recs_per_matrix(int G) { return (size_t)2 * G * 16 + (size_t)G * kCoopVals + kCoopVals; }

__device__ __forceinline__ void st16_agent(void* p, double v, int tag) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const v4i r = {(int)(unsigned)b, (int)(unsigned)(b >> 32), tag, 0};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ double rec_value(v4i r) {
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)r.y << 32) | (unsigned)r.x));
}
// one record, polled until it carries `want` (false: the abort flag is up or the bound was reached)
__device__ __forceinline__ bool poll_rec(const v4i* p, int want, int* ctl, double* out) {
  long spins = 0;
  while (true) {
    v4i r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if (r.z == want) { *out = rec_value(r); return true; }
    ++spins;
    if ((spins & 63) == 0 && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    if (spins > (1L << 22)) {
      // (what was waited for, for the host's message: kind 2 = a block record, the tag wanted, the tag seen, the record)
      if (atomicCAS(ctl, 0, 1) == 0) { ctl[1] = 2; ctl[2] = want; ctl[3] = r.z; ctl[4] = (int)(p - (const v4i*)nullptr); ctl[5] = blockIdx.x; }
      return false;
    }
  }
}
// The 16 column records of the workgroups g = w, w + 4, ... (wave w): lane = (g / 4 mod 4) * 16 + value, R rounds of 16
// workgroups; all loads of a poll in flight together.  Returns this wave's share of the 16 sums in lanes 0-15 (every lane
// l holds the sum of value l & 15).
template <int R>
__device__ __forceinline__ bool poll_cols(const v4i* base, int G, int w, int lane, int want, int* ctl, double* out) {
  const int v = lane & 15, q = lane >> 4;
  const v4i* ptr[R];
  bool need[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int g = w + 4 * (q + 4 * r);
    need[r] = g < G;
    ptr[r] = base + (size_t)(need[r] ? g : 0) * 16 + v;
  }
  v4i rec[R];
  long spins = 0;
  while (true) {
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(rec[r]) : "v"(ptr[r]) : "memory");
    if constexpr (R == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec[0])::"memory");
    if constexpr (R == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec[0]), "+v"(rec[1])::"memory");
    if constexpr (R == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec[0]), "+v"(rec[1]), "+v"(rec[2])::"memory");
    if constexpr (R == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec[0]), "+v"(rec[1]), "+v"(rec[2]), "+v"(rec[3])::"memory");
    if constexpr (R == 6)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(rec[0]), "+v"(rec[1]), "+v"(rec[2]), "+v"(rec[3]), "+v"(rec[4]), "+v"(rec[5])::"memory");
    if constexpr (R == 8)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(rec[0]), "+v"(rec[1]), "+v"(rec[2]), "+v"(rec[3]), "+v"(rec[4]), "+v"(rec[5]), "+v"(rec[6]), "+v"(rec[7])::"memory");
    bool fresh = true;
#pragma unroll
    for (int r = 0; r < R; ++r) fresh = fresh && (!need[r] || rec[r].z == want);
    if (__all(fresh)) break;
    ++spins;
    if ((spins & 63) == 0 && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    if (spins > (1L << 22)) {
      if (lane == 0 && atomicCAS(ctl, 0, 1) == 0) { ctl[1] = 1; ctl[2] = want; ctl[3] = rec[0].z; ctl[4] = w; ctl[5] = blockIdx.x; }
      return false;
    }
  }
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) s += need[r] ? rec_value(rec[r]) : 0.0;
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  *out = s;
  return true;
}

__global__ __launch_bounds__(kCoopRows) void k_panel_coop(double* __restrict__ a_all, long long stride_a,
                                                          double* __restrict__ tri_all, TriLayout TL,
                                                          double* __restrict__ sb_all, SbLayout SL, int j0, int G,
                                                          v4i* __restrict__ recs_all, int* __restrict__ ctl, int seq_base) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* Pl = sm;                          // [kB][kCoopLd]  the workgroup's rows of the panel
  double* Vl = Pl + kB * kCoopLd;           // [256][8]       the inner block's reflectors by row; then partial sums
  double* Ms = Vl + kCoopRows * 8;          // [8][kB]        M, then W = T^T M
  double* red = Ms + 8 * kB;                // [2][4][8]
  double* piv = red + 64;                   // [2][8]
  double* tauA = piv + 16;                  // [kB]           tau of every column (stored at the end)
  double* tot4 = tauA + kB;                 // [2][4][16]
  int* s_dead = reinterpret_cast<int*>(tot4 + 128);
  const int n = TL.n;
  const int r0 = j0 + kB, m = n - r0;
  const int g = blockIdx.x, mat = blockIdx.y;
  double* A = a_all + (size_t)mat * stride_a;
  double* tri = tri_all + (size_t)mat * TL.slab;
  double* sb = sb_all + (size_t)mat * SL.slab;
  // (round 6: one control record of 8 ints per matrix -- [0] the abort flag its workgroups poll, [1..5] what was waited
  // for --, so that a matrix whose wait timed out does not stop the matrices beside it half-way through their stores)
  ctl += 8 * (int)blockIdx.y;
  if (__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // given up in an earlier panel: k_panel_serial
  v4i* colrec = recs_all + (size_t)mat * coop_recs_per_matrix(G);
  v4i* mrec = colrec + (size_t)2 * G * 16;
  v4i* trec = mrec + (size_t)G * kCoopVals;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double* P = A + (size_t)j0 * n + r0;
  const int row = g * kCoopRows + tid;      // my row of the panel
  const bool ok = row < m;
  const int rc = min(row, m - 1);
  const int R = (G + 15) / 16;
  typedef char __attribute__((address_space(1)))* gbp;
  typedef double __attribute__((address_space(1)))* gdp;
  typedef const double __attribute__((address_space(1)))* gdp_c;
  auto off = [&](int rr) -> unsigned {
    unsigned e = 8u * (unsigned)rr;
    asm volatile("" : "+v"(e));
    return e;
  };
  auto st = [&](double* colbase, int rr, double val) { *(gdp)((gbp)colbase + off(rr)) = val; };
  if (tid == 0) *s_dead = 0;
  for (int c = 0; c < kB; ++c) {
    const double v = *(gdp_c)((gbp)(P + (size_t)c * n) + off(rc));
    Pl[c * kCoopLd + tid] = ok ? v : 0.0;
  }
  lds_barrier();

  for (int c0 = 0; c0 < kB; c0 += 8) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = Pl[(c0 + i) * kCoopLd + tid];
    // ---- the inner block's 8 reflectors
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int j = c0 + jj;
      const int pb = jj & 1;
      double gr[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) gr[c] = (c >= jj && row > j) ? x[jj] * x[c] : 0.0;
      const double gs = wave_reduce8(gr);
      if (lane < 8) red[(pb * 4 + wv) * 8 + wave_reduce8_index(lane)] = gs;
      if (row == j) {
#pragma unroll
        for (int c = 0; c < 8; ++c) piv[pb * 8 + c] = x[c];
      }
      lds_barrier();
      if (wv == 0 && lane < 16) {
        double val;
        if (lane < 8) val = (red[(pb * 4 + 0) * 8 + lane] + red[(pb * 4 + 1) * 8 + lane]) + (red[(pb * 4 + 2) * 8 + lane] + red[(pb * 4 + 3) * 8 + lane]);
        else val = g == 0 ? piv[pb * 8 + lane - 8] : 0.0;
        st16_agent(colrec + ((size_t)pb * G + g) * 16 + lane, val, seq_base + 1 + j);
      }
      {
        const v4i* base = colrec + (size_t)pb * G * 16;
        double s = 0.0;
        bool good;
        switch (R) {
          case 1: good = poll_cols<1>(base, G, wv, lane, seq_base + 1 + j, ctl, &s); break;
          case 2: good = poll_cols<2>(base, G, wv, lane, seq_base + 1 + j, ctl, &s); break;
          case 3: good = poll_cols<3>(base, G, wv, lane, seq_base + 1 + j, ctl, &s); break;
          case 4: good = poll_cols<4>(base, G, wv, lane, seq_base + 1 + j, ctl, &s); break;
          case 5: case 6: good = poll_cols<6>(base, G, wv, lane, seq_base + 1 + j, ctl, &s); break;
          default: good = poll_cols<8>(base, G, wv, lane, seq_base + 1 + j, ctl, &s); break;
        }
        if (lane < 16) tot4[(pb * 4 + wv) * 16 + lane] = s;
        if (!good && lane == 0) *s_dead = 1;
      }
      lds_barrier();
      if (*s_dead) return;
      // the 16 totals: lane l < 16 of every wave adds the four waves' shares of value l, the others take them from that
      // lane through scalar registers
      double fin[8], pv[8];
      {
        const int l16 = lane & 15;
        const double t = (tot4[(pb * 4 + 0) * 16 + l16] + tot4[(pb * 4 + 1) * 16 + l16]) +
                         (tot4[(pb * 4 + 2) * 16 + l16] + tot4[(pb * 4 + 3) * 16 + l16]);
        const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)tb, c);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(tb >> 32), c);
          const double val = __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
          if (c < 8) fin[c] = val; else pv[c - 8] = val;
        }
      }
      const HH h = householder(pv[jj], fin[jj]);
      if (tid == 0) tauA[j] = h.tau;
      const double v = row > j ? h.scale * x[jj] : (row == j ? 1.0 : 0.0);
#pragma unroll
      for (int c = jj + 1; c < 8; ++c) {
        const double wc = h.tau * (pv[c] + h.scale * fin[c]);
        x[c] -= v * wc;
      }
      // column j is final.  It stays in LDS in the form the matrix takes it (R entries above the pivot, beta at it, v below);
      // NOTHING is stored to memory inside the loop: a poll waits for vmcnt(0), i.e. for every store of the wave in flight
      Pl[j * kCoopLd + tid] = row > j ? v : (row == j ? h.beta : x[jj]);
      x[jj] = ok ? v : 0.0;
    }
    const int jn = c0 + 8;
    if (jn >= kB) break;
    const int ncols = kB - c0;               // the block's own columns (their Gram matrix) + the columns on the right
    const int nvals = 8 * ncols;
    const int blk = c0 >> 3;
    // ---- V by row (for the broadcast reads below; the block's own columns -- their Gram matrix -- are read from here too)
#pragma unroll
    for (int i = 0; i < 8; ++i) Vl[tid * 8 + i] = x[i];
    lds_barrier();
    // ---- M[i][c] = v_i . P[:, c] over this workgroup's rows: thread = (column, quarter of the rows)
    {
      const int c = tid & 63, q = tid >> 6;
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < ncols) {
        const double* vr = Vl + q * 64 * 8;
        const double* pc = c < 8 ? vr + c : Pl + (c0 + c) * kCoopLd + q * 64;
        const int stp = c < 8 ? 8 : 1;
#pragma unroll 4
        for (int rr = 0; rr < 64; ++rr) {
          const double pval = pc[rr * stp];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += vr[rr * 8 + i] * pval;
        }
      }
      lds_barrier();                        // (the partial sums take the place of V by row)
      if (c < ncols) {
#pragma unroll
        for (int i = 0; i < 8; ++i) Vl[(q * 64 + c) * 8 + i] = acc[i];
      }
    }
    lds_barrier();
    for (int idx = tid; idx < nvals; idx += kCoopRows) {
      const double t = (Vl[idx] + Vl[64 * 8 + idx]) + (Vl[2 * 64 * 8 + idx] + Vl[3 * 64 * 8 + idx]);
      st16_agent(mrec + (size_t)g * kCoopVals + idx, t, seq_base + 65 + blk);
    }
    lds_barrier();
    // ---- hop 1: value idx is summed by workgroup idx mod G
    bool good = true;
    const int nown = g < nvals ? (nvals - g + G - 1) / G : 0;
    for (int pi = tid; pi < nown * G; pi += kCoopRows) {
      const int k = pi / G, gg = pi - k * G;
      double val = 0.0;
      good = poll_rec(mrec + (size_t)gg * kCoopVals + (g + k * G), seq_base + 65 + blk, ctl, &val) && good;
      Vl[pi] = val;
    }
    if (!good) *s_dead = 1;
    lds_barrier();
    if (*s_dead) return;
    for (int k = tid; k < nown; k += kCoopRows) {
      double acc = 0.0;
      for (int gg = 0; gg < G; ++gg) acc += Vl[k * G + gg];
      st16_agent(trec + (g + k * G), acc, seq_base + 73 + blk);
    }
    // ---- hop 2: the totals
    for (int idx = tid; idx < nvals; idx += kCoopRows) {
      double val = 0.0;
      good = poll_rec(trec + idx, seq_base + 73 + blk, ctl, &val) && good;
      Ms[(idx & 7) * kB + c0 + (idx >> 3)] = val;
    }
    if (!good) *s_dead = 1;
    lds_barrier();
    if (*s_dead) return;
    // W = T^T M for the columns on the right (as in k_panel_wg): one thread per column, in place
    if (tid >= jn && tid < kB) {
      double wcol[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double acc = Ms[i * kB + tid];
#pragma unroll
        for (int l = 0; l < i; ++l) acc -= Ms[l * kB + c0 + i] * wcol[l];
        wcol[i] = tauA[c0 + i] * acc;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) Ms[i * kB + tid] = wcol[i];
    }
    lds_barrier();
    // P[:, c] -= V W[:, c] on my row
    for (int c = jn; c < kB; ++c) {
      double s2 = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) s2 += x[i] * Ms[i * kB + c];
      Pl[c * kCoopLd + tid] -= s2;
    }
    // (the next block reads its own row of Pl; Ms is still being read here by slower waves when a fast one is through
    // the next block's first barriers and writes it again: one barrier)
    lds_barrier();
  }
  // ---- the panel -> memory: the matrix' columns, and V in its explicit form into the three panel buffers
  lds_barrier();
  if (g == 0 && tid < kB) tri[TL.tau + j0 + tid] = tauA[tid];
  if (ok) {
    for (int c = 0; c < kB; ++c) {
      const double val = Pl[c * kCoopLd + tid];
      const double v = row > c ? val : (row == c ? 1.0 : 0.0);
      st(P + (size_t)c * n, row, val);
      st(sb + SL.vw + (size_t)c * n + r0, row, v);
      st(sb + SL.wv + (size_t)(kB + c) * n + r0, row, -v);   // (-[W|V])
      st(sb + SL.xv + (size_t)(2 * kB + c) * n + r0, row, v);
    }
  }
}

// The take-over of k_panel_coop, enqueued behind every one of its launches (round 6; until then a time-out failed the
// solve).  Workgroup b looks at matrix b's control record and returns unless its abort flag is up -- always, in practice.
// Otherwise the panel is intact in memory (k_panel_coop stores nothing before its last exchange has succeeded) and is
// factored here by this ONE workgroup, from memory, column by column: norm and pivot, reflector, the columns on the right
// eight at a time (w = tau (v^T P), P -= v w) -- slow (the panel is streamed 64 / 8 + 1 times per column group) and only
// there so that a solve whose cooperative launch could not get its workgroups resident together (another stream or
// process holds CUs with a persistent kernel of its own) still ends with LAPACK's numbers.  The flag stays up for the
// rest of the solve; the event is counted in stats[5] and the context keeps to the chunked launches afterwards.
__global__ __launch_bounds__(1024) void k_panel_serial(double* __restrict__ a_all, long long stride_a,
                                                       double* __restrict__ tri_all, TriLayout TL,
                                                       double* __restrict__ sb_all, SbLayout SL, int j0,
                                                       const int* __restrict__ ctl, unsigned long long* __restrict__ stats) {
  __shared__ double red[16 * 9];
  __shared__ double s_w[8];
  __shared__ double s_tau, s_beta, s_scale;
  const int mat = blockIdx.x;
  if (ctl[8 * mat] == 0) return;
  const int n = TL.n;
  const int r0 = j0 + kB, m = n - r0;
  double* A = a_all + (size_t)mat * stride_a;
  double* tri = tri_all + (size_t)mat * TL.slab;
  double* sb = sb_all + (size_t)mat * SL.slab;
  double* P = A + (size_t)j0 * n + r0;      // P(r, c) at P[c * n + r], r = 0 .. m - 1, c = 0 .. kB - 1
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) atomicAdd(stats + 5, 1ull);
  // sum of up to 8 values per thread over the workgroup: wave sums, then 16 partials per value
  auto block_sum8 = [&](double (&a)[8], int cnt) {
    for (int i = 0; i < cnt; ++i) {
      const double t = wave_sum(a[i]);
      if (lane == 0) red[wv * 9 + i] = t;
    }
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
      double t = 0.0;
      for (int w = 0; w < 16; ++w) t += red[w * 9 + i];
      a[i] = t;
    }
    __syncthreads();
  };
  for (int j = 0; j < kB; ++j) {
    double* pj = P + (size_t)j * n;
    {
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = j + 1 + tid; r < m; r += 1024) a[0] += pj[r] * pj[r];
      block_sum8(a, 1);
      if (tid == 0) {
        const HH h = householder(pj[j], a[0]);
        s_tau = h.tau; s_beta = h.beta; s_scale = h.scale;
        tri[TL.tau + j0 + j] = h.tau;
      }
      __syncthreads();
    }
    const double tau = s_tau, scale = s_scale;
    for (int c0 = j + 1; c0 < kB; c0 += 8) {
      const int cnt = min(8, kB - c0);
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = j + 1 + tid; r < m; r += 1024) {
        const double x = pj[r];
        for (int i = 0; i < cnt; ++i) a[i] += x * P[(size_t)(c0 + i) * n + r];
      }
      block_sum8(a, cnt);
      if (tid < cnt) s_w[tid] = tau * (P[(size_t)(c0 + tid) * n + j] + scale * a[tid]);
      __syncthreads();
      for (int r = j + tid; r < m; r += 1024) {
        const double v = r > j ? scale * pj[r] : 1.0;
        for (int i = 0; i < cnt; ++i) P[(size_t)(c0 + i) * n + r] -= v * s_w[i];
      }
      __syncthreads();
    }
    // column j in the form the matrix takes it: R above the pivot (untouched), beta at it, v below
    for (int r = j + tid; r < m; r += 1024) pj[r] = r > j ? scale * pj[r] : s_beta;
    __syncthreads();
  }
  // V in its explicit form into the three panel buffers (as k_panel_coop's last loop)
  for (int c = 0; c < kB; ++c)
    for (int r = tid; r < m; r += 1024) {
      const double val = P[(size_t)c * n + r];
      const double v = r > c ? val : (r == c ? 1.0 : 0.0);
      sb[SL.vw + (size_t)c * n + r0 + r] = v;
      sb[SL.wv + (size_t)(kB + c) * n + r0 + r] = -v;
      sb[SL.xv + (size_t)(2 * kB + c) * n + r0 + r] = v;
    }
}

// X1 / X2 = sum of their K slices (split-K SYMM, few matrices): blockIdx.y = column of [X1 | X2], rows r0 .. n - 1
__global__ __launch_bounds__(256) void k_sum_xslices(double* __restrict__ sb_all, SbLayout SL, int r0) {
  const int n = SL.n, p = SL.symm_split;
  double* sb = sb_all + (size_t)blockIdx.z * SL.slab;
  const int which = blockIdx.y / kB, c = blockIdx.y % kB;
  const int r = r0 + blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const double* src = sb + SL.xsplit + (size_t)which * p * n * kB + (size_t)c * n + r;
  double s = 0.0;
  for (int q = 0; q < p; ++q) s += src[(size_t)q * n * kB];
  sb[SL.xv + (size_t)(which * kB + c) * n + r] = s;
}

// second panel of a pair: P2 = [W1|V1]^T V2 (128 x 64) from its split-K slices
__global__ __launch_bounds__(256) void k_sum_p2(double* __restrict__ sb_all, SbLayout SL) {
  double* sb = sb_all + (size_t)blockIdx.y * SL.slab;
  const int i = blockIdx.x * 256 + threadIdx.x;   // 0 .. 2 kB kB - 1
  double acc = 0.0;
#pragma unroll
  for (int sl = 0; sl < kSmallSplit; ++sl) acc += sb[SL.small2 + (size_t)sl * 2 * kB * kB + i];
  sb[SL.p2 + i] = acc;
}

// One workgroup per matrix: T (larft, forward columnwise) from tau and G = V^T V;  S = T^T (V^T X) T;
// C = [T; T; -S/2]  (3 kB x kB, column-major), the right-hand factor of  W = [X1 | X2 | V] C.
__global__ __launch_bounds__(1024) void k_sb_small(const double* __restrict__ tri_all, TriLayout TL,
                                                   double* __restrict__ sb_all, SbLayout SL, int j0) {
  // One workgroup per matrix, on the critical path of every panel (QR -> SYMM -> Gram -> this -> W -> trailing update):
  // 1024 threads and a T factor by halving (16 x 16 diagonal blocks by substitution, then T12 = -T11 G12 T22 twice)
  // instead of 256 threads and larft's 64 dependent columns: 150-240 us -> see DESIGN section 7.
  constexpr int LD = kB + 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* G = sm;                 // [kB][LD]  G[i * LD + j]
  double* M1 = G + kB * LD;
  double* T = M1 + kB * LD;
  double* U = T + kB * LD;
  const double* tri = tri_all + (size_t)blockIdx.x * TL.slab;
  double* sb = sb_all + (size_t)blockIdx.x * SL.slab;
  const int tid = threadIdx.x, nthr = blockDim.x;
  // the split-K product is (kB x 3 kB), column-major ld kB: columns [X1 | X2 | V]
  const double* prod = sb + SL.small;
  const size_t slice = (size_t)kB * 3 * kB;
  for (int idx = tid; idx < kB * kB; idx += nthr) {
    const int i = idx & 63, jj = idx >> 6;
    double g = 0.0, x = 0.0;
#pragma unroll
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
  // ---- T = the compact-WY factor of the panel's reflectors: (D + striu(G)) T = I row by row, i.e.
  // T[i][c] = tau_i (delta_ic - sum_{l > i} G[i][l] T[l][c])  (larft's T; rows of tau = 0 reflectors come out zero).
  // Diagonal 16 x 16 blocks: one thread per column, rows bottom-up, the column in registers.
  if (tid < kB) {
    const int bb = tid >> 4, c = tid & 15, o = bb * 16;
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      double acc = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int l = i + 1; l < 16; ++l) acc -= G[(o + i) * LD + o + l] * x[l];
      x[i] = i <= c ? tri[TL.tau + j0 + o + i] * acc : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) T[(o + i) * LD + o + c] = x[i];
  }
  __syncthreads();
  // off-diagonal blocks by halving: T12 = -T11 (G12 T22), block sizes 16 then 32 (U serves as the scratch for G12 T22)
#pragma unroll
  for (int bs = 16; bs <= 32; bs *= 2) {
    const int npair = kB / (2 * bs);                 // 2 pairs of 16-blocks, then 1 pair of 32-blocks
    for (int idx = tid; idx < npair * bs * bs; idx += nthr) {
      const int pr = idx / (bs * bs), e = idx % (bs * bs), i = e / bs, jj = e % bs;
      const int o1 = pr * 2 * bs, o2 = o1 + bs;
      double acc = 0.0;
      for (int l = 0; l <= jj; ++l) acc += G[(o1 + i) * LD + o2 + l] * T[(o2 + l) * LD + o2 + jj];   // T22 upper triangular
      U[(o1 + i) * LD + o2 + jj] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < npair * bs * bs; idx += nthr) {
      const int pr = idx / (bs * bs), e = idx % (bs * bs), i = e / bs, jj = e % bs;
      const int o1 = pr * 2 * bs, o2 = o1 + bs;
      double acc = 0.0;
      for (int l = i; l < bs; ++l) acc += T[(o1 + i) * LD + o1 + l] * U[(o1 + l) * LD + o2 + jj];    // T11 upper triangular
      T[(o1 + i) * LD + o2 + jj] = -acc;
    }
    __syncthreads();
  }
  // U = M1 T ; S = T^T U
  for (int idx = tid; idx < kB * kB; idx += nthr) {
    const int i = idx >> 6, jj = idx & 63;
    double s = 0.0;
    for (int l = 0; l <= jj; ++l) s += M1[i * LD + l] * T[l * LD + jj];
    U[i * LD + jj] = s;
  }
  __syncthreads();
  double* cm = sb + SL.cmat;   // ld 3 kB
  for (int idx = tid; idx < kB * kB; idx += nthr) {
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

// Diagnostic build (-DBULGE_STAMPS): shader cycles of wave 0 between five points of every task, summed over all tasks
// since the last read (sc_dbg_bulge_stamps, tools/bulge_stamps.py); no stamp executes in the normal build.
#ifdef BULGE_STAMPS
__device__ unsigned long long g_bulge_stamps[8];
#define BULGE_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");
#else
#define BULGE_STAMP(var)
#endif

typedef double __attribute__((address_space(1)))* gdptr;          // global memory: global_load / global_store, never flat
typedef const double __attribute__((address_space(1)))* gdptr_c;
__device__ __forceinline__ double ld_l2(gdptr_c p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a pointer every lane of the wave holds the same value of, told so to the compiler (scalar registers, and memory
// instructions of the form scalar base + 32-bit lane offset instead of a 64-bit address per access)
__device__ __forceinline__ gdptr wave_uniform(double* p) {
  const unsigned long long b = (unsigned long long)(size_t)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  return (gdptr)(size_t)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

// One task (sweep s, chase position k) of the bulge chase by one workgroup of 256 threads, blocks from and to the band
// storage: the body of k_bulge_step (one launch per wavefront of tasks) and of k_chase_finish (the take-over after a
// persistent chase that gave up).
struct BulgeTaskLds {
  double E[kB * (kB + 1)];
  double vp[kB], vn[kB], u[kB], red[4 * kB];
  double s_tau, s_beta;
};
__device__ __forceinline__ void bulge_task(double* __restrict__ sb, const SbLayout& SL, int s, int k, BulgeTaskLds& W) {
  constexpr int LD = kB + 1;
  double* const E = W.E;
  double* const D = E;   // the diagonal block is processed after E has gone back to memory: same buffer
  double* const vp = W.vp;
  double* const vn = W.vn;
  double* const u = W.u;
  double* const red = W.red;
  double& s_tau = W.s_tau;
  double& s_beta = W.s_beta;
  const int n = SL.n;
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
  BULGE_STAMP(t0)

  // the diagonal block D = AB(r0 .. r0+L-1, r0 .. r0+L-1) (lower stored) is not touched before its own update:
  // fetch it right away (unconditional loads with clamped indices, masked when they go to LDS)
  // (one wave-uniform base per block + 32-bit element offsets instead of a 64-bit address per access)
  gdptr colbase = wave_uniform(ab + (size_t)r0 * kLdab);                    // AB(r0, r0)
  double d16[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int jc = std::min(q * 16 + u, L - 1);
    const int ic = std::min(std::max(i, jc), L - 1);
    d16[u] = colbase[(unsigned)((ic - jc) + jc * kLdab)];
  }

  if (k > 0) {
    gdptr ebase = wave_uniform(ab + (size_t)(r0 - kB) * kLdab);             // AB(r0 - kB, r0 - kB)
    const double* vdp = sb + SL.vd + (dia - 1) * kDiaSize + (size_t)cc * kG + cc;
    const double tau_p = sb[SL.tau2 + (dia - 1) * kG + cc];
    if (tid < kB) vp[tid] = vdp[(size_t)tid * kG];
    // E(i, j) = AB(r0 + i, c0 + j), rows i < L: thread (i, q) keeps row i, columns 16 q .. 16 q + 15 in registers for the
    // row-wise steps; only the column sums of the second reflector go through an LDS image.
    double t16[16];
    {
      const int ic = std::min(i, L - 1);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int jj = q * 16 + u;
        t16[u] = ebase[(unsigned)((kB + ic - jj) + jj * kLdab)];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) t16[u] = i < L ? t16[u] : 0.0;
    }
    lds_barrier();   // vp
#ifdef BULGE_STAMPS
    { BULGE_STAMP(t1) if (tid == 0) atomicAdd(&g_bulge_stamps[0], t1 - t0); }
#endif
    // u = tau_p E vp: partial sums over this wave's 16 columns, every thread adds the four partials of its own row
    {
      double a = 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) a += t16[u] * vp[q * 16 + u];
      red[q * kB + i] = a;
    }
    lds_barrier();
    {
      const double ui = tau_p * ((red[i] + red[kB + i]) + (red[2 * kB + i] + red[3 * kB + i]));
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        t16[u] -= ui * vp[q * 16 + u];
        E[i * LD + q * 16 + u] = t16[u];
      }
    }
    // new reflector from the first column of E, which wave 0 holds in t16[0]
    if (q == 0) {
      const double x = t16[0];
      const double t2 = wave_sum((i >= 1 && i < L) ? x * x : 0.0);
      const HH h = householder(__shfl(x, 0), t2);
      vn[i] = i == 0 ? 1.0 : (i < L ? x * h.scale : 0.0);
      if (i == 0) { s_tau = h.tau; s_beta = h.beta; }
    }
    lds_barrier();
    // z_j = sum_i E[i, j] v_i  (j >= 1);  thread (j = i, rows q*16..)
    {
      double a = 0.0;
#pragma unroll
      for (int ii = q * 16; ii < q * 16 + 16; ++ii) a += E[ii * LD + i] * vn[ii];
      red[q * kB + i] = a;
    }
    lds_barrier();
    if (tid < kB) u[tid] = s_tau * ((red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]));
    lds_barrier();
    // E <- H E (from the registers), first column = beta e1; write back.  Nobody reads the LDS image of E any more (the
    // column sums were two barriers ago), so the diagonal block may take over its buffer without another barrier.
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int jj = q * 16 + c;
      double e = t16[c] - vn[i] * u[jj];
      if (jj == 0) e = i == 0 ? s_beta : 0.0;
      if (i < L) ebase[(unsigned)((kB + i - jj) + jj * kLdab)] = e;
    }
#ifdef BULGE_STAMPS
    { BULGE_STAMP(t2) if (tid == 0) { atomicAdd(&g_bulge_stamps[1], t2 - t0); atomicAdd(&g_bulge_stamps[5], 1ull); } }
#endif
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
    lds_barrier();
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
  lds_barrier();
  BULGE_STAMP(t3)
  {
    double a = 0.0;
#pragma unroll
    for (int jj = q * 16; jj < q * 16 + 16; ++jj) a += D[i * LD + jj] * vn[jj];
    red[q * kB + i] = a;
  }
  lds_barrier();
  if (tid < 64) {
    const double p = s_tau * ((red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]));
    const double dot = wave_sum(p * vn[tid]);
    const double alpha2 = -0.5 * s_tau * dot;
    u[tid] = p + alpha2 * vn[tid];   // w
  }
  lds_barrier();
  for (int jj = q * 16; jj < q * 16 + 16; ++jj)
    if (i >= jj && i < L)
      colbase[(unsigned)((i - jj) + jj * kLdab)] = D[i * LD + jj] - vn[i] * u[jj] - u[i] * vn[jj];

  // the reflector goes into its diamond
  if (tid < L) vd[(size_t)tid * kG] = vn[tid];
  if (tid == 0) sb[SL.tau2 + dia * kG + cc] = s_tau;
#ifdef BULGE_STAMPS
  {
    BULGE_STAMP(t4)
    if (tid == 0) {
      atomicAdd(&g_bulge_stamps[2], t3 - t0);
      atomicAdd(&g_bulge_stamps[3], t4 - t0);
      atomicAdd(&g_bulge_stamps[4], 1ull);
    }
  }
#endif
}

// Launch t of the bulge chase: workgroup x handles task (s, k) with k = (t & 1) + 2x, s = (t - k) / 2.
__global__ __launch_bounds__(256, 4) void k_bulge_step(double* __restrict__ sb_all, SbLayout SL,
                                                    int t, const int* __restrict__ done = nullptr) {
  __shared__ BulgeTaskLds W;
  const int n = SL.n;
  const int k = (t & 1) + 2 * (int)blockIdx.x;
  const int s = (t - k) / 2;
  if (s < 0 || s > n - 3 || k >= chase_len(n, s)) return;
  // resuming a persistent chase that gave up (k_bulge_chase): done[b][s] tasks of sweep s are finished already
  if (done && k < done[(size_t)blockIdx.y * n + s]) return;
  bulge_task(sb_all + (size_t)blockIdx.y * SL.slab, SL, s, k, W);
}

// Take-over after a persistent chase (k_bulge_chase / k_bulge_pair), enqueued behind every one of them so that the host
// never has to look at the chase's outcome inside a solve (round 6: the solve only enqueues).  A chase that completed
// -- always, in practice -- costs this launch a look at the control block.  Otherwise (the stop flag is up: a wait ran
// into its bound or the test hook fired; or an XCD that owns matrices received no workgroup) workgroup b finishes matrix
// b alone, task after task in the order sweep, position from the published counts -- a valid order of the chase's
// dependences whatever the state the workgroups left behind, and slow: it is the last line of defence, counted in
// stats[] (read at the next synchronising call: sc_collect_events), after which the context keeps to the per-wavefront
// launches.  stats (unsigned long long, sc_ctx::d_status): [2] chases that needed it, [3] of those: stop flag raised by a
// time-out (not by the test hook), [4] sweeps finished by the persistent kernels (all chases), [6] incomplete without a flag.
__global__ __launch_bounds__(256) void k_chase_finish(double* __restrict__ sb_all, SbLayout SL, int batch, int nxcd,
                                                      int* __restrict__ progress, const int* __restrict__ ctl,
                                                      unsigned long long* __restrict__ stats) {
  __shared__ BulgeTaskLds W;
  const int n = SL.n;
  const int b = blockIdx.x;
  long long sweeps = 0;
  for (int x = 0; x < 8; ++x) sweeps += ctl[16 + x];
  const bool complete = ctl[0] == 0 && sweeps == (long long)batch * (n - 2);
  if (b == 0 && threadIdx.x == 0) {
    atomicAdd(stats + 4, (unsigned long long)sweeps);
    if (!complete) {
      atomicAdd(stats + 2, 1ull);
      if (ctl[0] && ctl[1] == 0) atomicAdd(stats + 3, 1ull);
      if (!ctl[0]) atomicAdd(stats + 6, 1ull);
    }
  }
  if (complete) return;
  int* prog = progress + (size_t)b * n;
  double* sb = sb_all + (size_t)b * SL.slab;
  for (int s = 0; s <= n - 3; ++s) {
    const int len = chase_len(n, s);
    for (int k = prog[s]; k < len; ++k) {
      bulge_task(sb, SL, s, k, W);
      __syncthreads();   // (the task's stores are visible to this workgroup's next task: same CU, same L1 / L2 path)
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// Persistent form of the bulge chase: ONE launch, no launch per wavefront.  A workgroup CLAIMS the next unclaimed sweep
// of a matrix (an atomic counter per matrix) and walks it down the band, task after task; task (s, k) starts when sweep
// s - 1 has published k + 2 finished tasks (progress[b][s - 1], one counter per sweep), which is the only dependence
// the chase has.  The previous reflector of the sweep stays in LDS (no round trip through the diamond storage), a sweep
// never waits for a launch boundary, and the matrices drift apart freely, so no round of workgroups is left half empty.
//
// Forward progress by construction: sweeps are claimed in order by workgroups that are running, so the owner of sweep
// s - 1 is always running or finished when somebody waits for it -- whatever the number of resident workgroups and
// however the dispatcher spreads them over the XCDs.  (Round 2 gave workgroup r of W the sweeps r, r + W, ... statically;
// a role that was not resident orphaned its sweeps and the successor spun into its bound.)  No cooperative launch and no
// placement census is needed any more.
//
// Coherence without L2 write-backs: all workgroups of a matrix sit on ONE XCD (matrix b on XCD b mod 8; a workgroup
// reads the XCD it runs on from XCC_ID and only ever touches that XCD's matrices, their claim counters and progress
// counters), i.e. behind one L2.  A task's stores are drained (vmcnt(0)) and fenced by a workgroup barrier before thread
// 0 publishes the counter; consumers read the counter and the band through agent-scope relaxed atomic loads, which
// bypass the CU's L1 and are served by that L2 (26 us per hand-off with release / acquire at agent scope, 3 us this way:
// tools/probe_handoff.hip).  A workgroup starts on its "home" matrix (ticket mod matrices of the XCD) and moves on to
// the XCD's other matrices when that one has no unclaimed sweep left, so every matrix of an XCD is finished as long as
// ONE workgroup lands there.  Every spin is bounded as a last line of defence: a time-out raises a flag that ends all
// workgroups between tasks; the host counts it (sc_ctx_get_counter "chase_timeouts"), finishes the chase with the
// per-wavefront launches from the published counters and does not use the persistent form on that context again.
//
// ctl (ints): [0] time-out flag, [1] workgroup-local tasks after which the test hook raised it, [2..9] tickets drawn
// per XCD, [10..12] (matrix, sweep, task) of the wait that timed out, [16..23] sweeps finished per XCD.
constexpr int kChaseCtlInts = 32;

// Early hand-off (round 4).  Task (s, k) reads three things from sweep s - 1: the blocks task (s - 1, k) has left --
// all its rows but the last --, and from task (s - 1, k + 1) its last row: the entry beta (what remains of the column that
// task annihilated) in the off-diagonal block and the row of the diagonal block.  The right update of the off-diagonal
// block and the new reflector only need beta, which task (s - 1, k + 1) knows as soon as it has generated ITS reflector --
// two thirds of a task before its stores have drained.  So a task publishes (beta, position) in a 16-byte record right
// there, its successor in the next sweep starts when task (s - 1, k) is complete and that record is in, loads its blocks
// and runs the first half of its work, and only then waits for (s - 1, k + 1) to be complete, fetches the last row of its
// diagonal block, and stores.  Nothing is stored before that second wait, so the order of the writes to the band is the
// one of the plain dependence.  The dependent chain of a latency-bound chase -- one task per link -- shortens by the
// loads and the first half of the arithmetic (profiles/r04_bulge_sweep.txt).
// Records: a ring of kEarlyRing per sweep, slot k mod kEarlyRing holds {beta of task k, k + 1}; written and read as ONE
// 16-byte access.  A reader that finds a later tag knows that its task (s - 1, k + 1) is long complete and takes the
// entry from the band instead.
constexpr int kEarlyRing = 8;
__device__ __forceinline__ void st16(void* p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st16_sc1(void* p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// a band entry as the chase's workgroups hand it on.  SPREAD = 0: all workgroups of a matrix sit behind ONE L2, a plain
// store (drained before the counter moves) is what the consumer's sc1 load finds there.  SPREAD = 1 (round 6: one large
// matrix chased from every XCD): a write-through store at agent scope, which a consumer on another XCD reads from memory
// with its sc1 load -- the pair MI355X_MICROARCH.md lists as valid without any fence (k_panel_coop's records use it too)
template <int SPREAD>
__device__ __forceinline__ void st_band(gdptr p, double v) {
  if (SPREAD) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
__device__ __forceinline__ v4i ld16_l2(const void* p) {
  v4i r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return r;
}

// Diagnostic build (-DCHASE_STAMPS): shader cycles of every wave of k_bulge_chase between ten points of a task, summed over
// all tasks of all workgroups (sc_dbg_chase_stamps, tools/chase_stamps.py): [1] wait for the predecessor sweep (poll +
// barrier), [2] loads of both blocks + first products up to the first barrier, [3] right update of E + the new reflector,
// [4] column sums + u, [5] second wait (early hand-off), [6] E stores issued + D into LDS, [7] D products, [8] w, [9] D update
// + stores issued, [10] store drain + barrier, [11] publish; [15] = waves x tasks.  Tasks at k = 0 have no E block: their
// segments [2] .. [4] are the reflector of the column alone.
#ifdef CHASE_STAMPS
__device__ unsigned long long g_chase_stamps[16];
#define CHASE_STAMP_DECL unsigned long long cs_prev = 0, cs_n = 0, cs_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CHASE_STAMP(i)                                                                 \
  {                                                                                    \
    unsigned long long t_;                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    if ((i) != 0) cs_sum[i] += t_ - cs_prev;                                           \
    cs_prev = t_;                                                                      \
  }
#define CHASE_STAMP_WRITE                                                                          \
  if ((threadIdx.x & 63) == 0) {                                                                   \
    for (int i_ = 1; i_ < 12; ++i_) atomicAdd(&g_chase_stamps[i_], cs_sum[i_]);                    \
    atomicAdd(&g_chase_stamps[15], cs_n);                                                          \
  }
#else
#define CHASE_STAMP_DECL
#define CHASE_STAMP(i)
#define CHASE_STAMP_WRITE
#endif
// SPREAD = 1 (round 6): the workgroups of ONE matrix on all XCDs -- the launch passes nxcd = 1, so every workgroup
// counts as "XCD 0" and serves every matrix; band stores and the early records are written through at agent scope
// (st_band).  For a few large matrices (config C5: one n = 24000 matrix has ~188 tasks per wavefront, three times the
// workgroups one XCD holds): until round 5 those took one launch per wavefront (48 000 launches of 8.7 us).
template <int SPREAD>
__global__ __launch_bounds__(256, 2) void k_bulge_chase(double* __restrict__ sb_all, SbLayout SL, int batch, int W, int nxcd,
                                                     int* __restrict__ progress, int* __restrict__ next_sweep,
                                                     int* __restrict__ ctl, int give_up_after, v4i* __restrict__ early) {
  constexpr int LD = kB + 1;
  __shared__ double E[kB * LD];
  double* D = E;
  __shared__ double vbuf[2][kB], u[kB], red[4 * kB];
  __shared__ double s_tau, s_beta, s_beta_in;
  __shared__ int s_go, s_claim, s_mode;

  const int n = SL.n;
  const int tid = threadIdx.x;
  const int i = tid & 63, q = tid >> 6;
  __shared__ int s_xcd, s_slot;
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    s_xcd = (int)(id & 7u) % nxcd;
    s_slot = atomicAdd(ctl + 2 + s_xcd, 1);
  }
  __syncthreads();
  const int xcd = s_xcd, slot = s_slot;
  const int mpx = xcd < batch ? (batch - xcd + nxcd - 1) / nxcd : 0;   // matrices of this XCD: xcd, xcd + nxcd, ...
  if (mpx == 0 || slot >= mpx * W) return;                      // no more than W workgroups per matrix
  const int K0 = chase_len(n, 0);
  int tasks_left = give_up_after;

  CHASE_STAMP_DECL
  int cur = slot % mpx, exhausted = 0;
  while (exhausted < mpx) {
    const int b = xcd + nxcd * cur;
    int* prog = progress + (size_t)b * n;
    // ---- claim the next sweep of matrix b (the counter is only ever touched from this XCD)
    if (tid == 0) s_claim = atomicAdd(next_sweep + b, 1);
    __syncthreads();
    const int s = __builtin_amdgcn_readfirstlane(s_claim);
    __syncthreads();
    if (s > n - 3) {
      ++exhausted;
      cur = cur + 1 < mpx ? cur + 1 : 0;
      continue;
    }
    exhausted = 0;
    double* sb = sb_all + (size_t)b * SL.slab;
    double* ab = sb + SL.ab;
    {
      const int len = chase_len(n, s);
      const int S = s / kG, cc = s - S * kG;
      const size_t dia0 = (size_t)S * K0 - (size_t)S * (S - 1) / 2;
      const int len_prev = s > 0 ? chase_len(n, s - 1) : 0;
      double tau_p = 0.0;
      for (int k = 0; k < len; ++k) {
        double* vp = vbuf[(k + 1) & 1];   // reflector of task k - 1 of this sweep
        double* vn = vbuf[k & 1];
        // ---- wait for sweep s - 1.  With a task (s - 1, k + 1) to wait for (early hand-off, see above): task (s - 1, k)
        // complete AND either (s - 1, k + 1) complete too (mode 2: everything is in the band) or its record in (mode 1:
        // beta from the record, the last row of the diagonal block after the second wait).  Else (mode 0): all of sweep
        // s - 1 that touches these rows is complete.
        const bool early_task = early != nullptr && s > 0 && k + 1 < len_prev;
        CHASE_STAMP(0)
        if (tid == 0) {
          // (the stop flag and the predecessor's progress are requested TOGETHER: one L2 round trip, ~0.7 us, per task on
          // the sweep-to-sweep critical path instead of two; while waiting, the flag is looked at every 16th poll only)
          const int need = s > 0 ? min(k + (early_task ? 1 : 2), len_prev) : 0;
          const int stop0 = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int have = s > 0 ? __hip_atomic_load(prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
          int go = stop0 ? 0 : 1;
          int mode = 0;
          double beta_in = 0.0;
          const v4i* rec = early_task ? early + ((size_t)b * n + s - 1) * kEarlyRing + ((k + 1) & (kEarlyRing - 1)) : nullptr;
          long spins = 0;
          while (go) {
            if (have >= need) {
              if (!early_task) break;
              if (have >= k + 2) { mode = 2; break; }
              const v4i r = ld16_l2(rec);
              if (r.z == k + 2) {
                mode = 1;
                beta_in = __longlong_as_double((long long)(((unsigned long long)(unsigned)r.y << 32) | (unsigned)r.x));
                break;
              }
            }
            // (2^21 polls of ~1 us: seconds, orders of magnitude above any wait for a running workgroup)
            ++spins;
            if (spins > (1L << 21) ||
                ((spins & 15) == 0 && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
              if (!__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                ctl[10] = b; ctl[11] = s; ctl[12] = k;
              }
              __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              go = 0;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            have = __hip_atomic_load(prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          s_go = go;
          s_mode = mode;
          s_beta_in = beta_in;
        }
        __syncthreads();
        if (!s_go) return;
        CHASE_STAMP(1)
        const int mode = s_mode;   // 1: the last row of both blocks is not in the band yet
        // second wait of a task that started on the record (mode 1): task (s - 1, k + 1) complete, then the last row of
        // the diagonal block.  Nothing of this task has been stored before.
#define CHASE_SECOND_WAIT()                                                                                       \
        if (mode == 1) {                                                                                          \
          if (tid == 0) {                                                                                         \
            int go = 1;                                                                                           \
            long spins = 0;                                                                                       \
            while (__hip_atomic_load(prog + s - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 2) {         \
              ++spins;                                                                                            \
              if (spins > (1L << 21) ||                                                                           \
                  ((spins & 15) == 0 && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {    \
                if (!__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {                        \
                  ctl[10] = b; ctl[11] = s; ctl[12] = k;                                                          \
                }                                                                                                 \
                __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                           \
                go = 0;                                                                                           \
                break;                                                                                            \
              }                                                                                                   \
              __builtin_amdgcn_s_sleep(1);                                                                        \
            }                                                                                                     \
            s_go = go;                                                                                            \
          }                                                                                                       \
          __syncthreads();                                                                                        \
          if (!s_go) return;                                                                                      \
        }

        const int r0 = s + 1 + k * kB;
        const int L = min(kB, n - r0);
        const size_t dia = dia0 + k;
        double* vd = sb + SL.vd + dia * kDiaSize + (size_t)cc * kG + cc;
        // (one wave-uniform base per block + 32-bit element offsets: a 64-bit pointer per access costs the kernel a
        // workgroup per CU in registers)
        gdptr colbase_k = wave_uniform(ab + (size_t)r0 * kLdab);                 // AB(r0, r0)
        double d16[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int jc = min(q * 16 + c, L - 1);
          const int ic = min(max(i, jc), L - 1);
          d16[c] = ld_l2(colbase_k + (unsigned)((ic - jc) + jc * kLdab));
        }
        if (k > 0) {
          gdptr ebase_k = wave_uniform(ab + (size_t)(r0 - kB) * kLdab);          // AB(r0 - kB, r0 - kB)
          double t16[16];
          {
            const int ic = min(i, L - 1);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
              const int jj = q * 16 + c;
              t16[c] = ld_l2(ebase_k + (unsigned)((kB + ic - jj) + jj * kLdab));
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) t16[c] = i < L ? t16[c] : 0.0;
            if (mode == 1 && i == kB - 1) {   // the last row: zeros and the predecessor's beta (the band still holds the old entry)
              const double bin = s_beta_in;
#pragma unroll
              for (int c = 0; c < 16; ++c) t16[c] = q * 16 + c == kB - 1 ? bin : 0.0;
            }
          }
          {
            double a = 0.0;
#pragma unroll
            for (int c = 0; c < 16; ++c) a += t16[c] * vp[q * 16 + c];
            red[q * kB + i] = a;
          }
          lds_barrier();
          CHASE_STAMP(2)
          {
            const double ui = tau_p * ((red[i] + red[kB + i]) + (red[2 * kB + i] + red[3 * kB + i]));
#pragma unroll
            for (int c = 0; c < 16; ++c) {
              t16[c] -= ui * vp[q * 16 + c];
              E[i * LD + q * 16 + c] = t16[c];
            }
          }
          if (q == 0) {
            const double x = t16[0];
            const double t2 = wave_sum((i >= 1 && i < L) ? x * x : 0.0);
            const HH h = householder(__shfl(x, 0), t2);
            vn[i] = i == 0 ? 1.0 : (i < L ? x * h.scale : 0.0);
            if (i == 0) {
              s_tau = h.tau; s_beta = h.beta;
              if (early) {   // the record of this task: all its successor in sweep s + 1 needs to start on
                const unsigned long long bb = (unsigned long long)__double_as_longlong(h.beta);
                if (SPREAD) st16_sc1(early + ((size_t)b * n + s) * kEarlyRing + (k & (kEarlyRing - 1)), v4i{(int)(unsigned)bb, (int)(unsigned)(bb >> 32), k + 1, 0});
                else st16(early + ((size_t)b * n + s) * kEarlyRing + (k & (kEarlyRing - 1)), v4i{(int)(unsigned)bb, (int)(unsigned)(bb >> 32), k + 1, 0});
              }
            }
          }
          lds_barrier();
          CHASE_STAMP(3)
          {
            double a = 0.0;
#pragma unroll
            for (int ii = q * 16; ii < q * 16 + 16; ++ii) a += E[ii * LD + i] * vn[ii];
            red[q * kB + i] = a;
          }
          lds_barrier();
          if (tid < kB) u[tid] = s_tau * ((red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]));
          lds_barrier();
          CHASE_STAMP(4)
          CHASE_SECOND_WAIT()
          CHASE_STAMP(5)
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const int jj = q * 16 + c;
            double e = t16[c] - vn[i] * u[jj];
            if (jj == 0) e = i == 0 ? s_beta : 0.0;
            if (i < L) st_band<SPREAD>(ebase_k + (unsigned)((kB + i - jj) + jj * kLdab), e);
          }
        } else {
          gdptr col_s = wave_uniform(ab + (size_t)s * kLdab);
          if (tid < 64) {
            const double xr = ld_l2(col_s + (unsigned)(1 + min(tid, L - 1)));
            double x = tid < L ? xr : 0.0;
            if (mode == 1 && tid == kB - 1) x = s_beta_in;   // (its last entry is task (s - 1, 1)'s beta)
            const double t2 = wave_sum(tid >= 1 ? x * x : 0.0);
            const HH h = householder(__shfl(x, 0), t2);
            vn[tid] = tid == 0 ? 1.0 : x * h.scale;
            if (tid == 0) { s_tau = h.tau; s_beta = h.beta; }
          }
          lds_barrier();
          CHASE_STAMP(4)
          CHASE_SECOND_WAIT()
          CHASE_STAMP(5)
          if (tid < L) st_band<SPREAD>(col_s + (unsigned)(1 + tid), tid == 0 ? s_beta : 0.0);
        }
#undef CHASE_SECOND_WAIT
        // ---- diagonal block (after the second wait its last row is in the band)
        if (mode == 1 && i == kB - 1) {
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const int jc = q * 16 + c;
            d16[c] = ld_l2(colbase_k + (unsigned)((kB - 1 - jc) + jc * kLdab));
          }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int jj = q * 16 + c;
          if (i >= jj) {
            const double x = (i < L) ? d16[c] : 0.0;
            D[i * LD + jj] = x;
            D[jj * LD + i] = x;
          }
        }
        lds_barrier();
        CHASE_STAMP(6)
        {
          double a = 0.0;
#pragma unroll
          for (int jj = q * 16; jj < q * 16 + 16; ++jj) a += D[i * LD + jj] * vn[jj];
          red[q * kB + i] = a;
        }
        lds_barrier();
        CHASE_STAMP(7)
        const double tau_now = s_tau;
        if (tid < 64) {
          const double pp = tau_now * ((red[tid] + red[kB + tid]) + (red[2 * kB + tid] + red[3 * kB + tid]));
          const double dot = wave_sum(pp * vn[tid]);
          u[tid] = pp - 0.5 * tau_now * dot * vn[tid];
        }
        lds_barrier();
        CHASE_STAMP(8)
        for (int jj = q * 16; jj < q * 16 + 16; ++jj)
          if (i >= jj && i < L)
            st_band<SPREAD>(colbase_k + (unsigned)((i - jj) + jj * kLdab), D[i * LD + jj] - vn[i] * u[jj] - u[i] * vn[jj]);
        if (tid < L) vd[(size_t)tid * kG] = vn[tid];
        if (tid == 0) sb[SL.tau2 + dia * kG + cc] = tau_now;
        tau_p = tau_now;
        // ---- publish: this task's stores are in the L2 before the counter moves
        CHASE_STAMP(9)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CHASE_STAMP(10)
        if (tid == 0) {
          __hip_atomic_store(prog + s, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (k + 1 == len) atomicAdd(ctl + 16 + xcd, 1);
          // test hook (sc_dbg_set_chase): raise the time-out flag after that many tasks of this workgroup
          if (give_up_after > 0 && --tasks_left == 0) {
            ctl[1] = give_up_after;
            __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        CHASE_STAMP(11)
#ifdef CHASE_STAMPS
        ++cs_n;
#endif
      }
    }
  }
  CHASE_STAMP_WRITE
}

// ----------------------------------------------------------------------------------------------------------------
// Pair form of the persistent chase (round 4): ONE workgroup walks TWO consecutive sweeps down the band.
//
// Why: every byte a task stores leaves the XCD's L2 (the L2 writes through), so the chase moves 96 KB of HBM traffic
// per task whatever the cache does -- 1.75 TB per 64 x n = 6000 step, the floor of both older forms -- and every
// hand-off between sweeps goes through memory.  Sweep s + 1 touches, two tasks later, the very blocks sweep s has just
// left (shifted by one row and one column).  Here team A (threads 0..255) runs the tasks of sweep sA = 2 p and leaves
// its blocks in LDS; team B (threads 256..511) runs sweep sB = sA + 1 two positions behind and takes its blocks from
// there: per PAIR of tasks one block set is read from memory (A) and one is written (B) -- half the traffic -- and the
// A -> B hand-off costs a workgroup barrier instead of store drain + publish + poll + L2 loads.
//
// LDS: a ring of three position slots (E 64 x 65 + D packed lower, 49 920 B each): at step m, A works on position m
// (slot m % 3), B on position m - 2, position m - 1 waits.  B's block at position k is A's block shifted by (1, 1):
//   E'(i, j) = E_k(i + 1, j + 1),  E'(i, 63) = D_k(i + 1, 0),  E'(63, 63) = E_{k+1}(0, 0),  E'(63, j < 63) = 0
//   D'(i, j) = D_k(i + 1, j + 1),  D'(63, j) = E_{k+1}(0, j + 1),  D'(63, 63) = D_{k+1}(0, 0)
// so every element A leaves behind is consumed by exactly one task of B -- except the entries A has just annihilated
// (E_k(1.., 0)), which no block of sweep sB covers, and the corner D_0(0, 0) of A's first block, which has no task of B
// above it: A stores those to the band itself.  tools/models/bulge_pair_model.py is the NumPy model of this data flow
// (tests/test_bulge_pair_model.py runs it on the CPU).  The column sums a task needs (z = E^T v and the strictly-lower
// half of p = D v) read an LDS image of the block: for A that is its slot, for B the slot it has just emptied.
//
// Both teams keep ONE barrier schedule -- eight workgroup barriers per step, [0] .. [7] -- and differ in where blocks come
// from (A: memory through the L2, B: LDS) and go to (A: LDS, B: memory).  A step is either the common one (both teams on
// full 64-row blocks, neither at a sweep start: pair_step_full<TEAM>, one instruction stream per team without masks,
// clamps or case distinctions) or the general one (pair_step_general: every case, out of line so that its registers are
// not charged to the common step).  B's store drain and its publish sit behind the NEXT step's barrier [1], where team
// A waits for its E loads anyway; A's D loads are issued behind that barrier and land during steps (2)-(4).
//
// Dependences across workgroups are those of k_bulge_chase: pairs are CLAIMED in order (so the owner of sweep sA - 1 is
// running or done), A's task k starts when progress[sA - 1] >= k + 2, B publishes progress[sB] after its stores have
// drained; all workgroups of a matrix sit on one XCD.  When a wait runs into its bound or the stop flag is up, the
// workgroup writes its live slots back to the band and publishes both sweeps' counts, so the per-wavefront launches can
// finish the chase from the counters exactly as after k_bulge_chase.
constexpr int kSlotE = kB * (kB + 1);          // E block, row stride 65 (conflict-free column access)
// D block, lower triangle packed by rows: (i, j) at i (i + 1) / 2 + j (2080 doubles).  The region is a little larger: the
// loader waves (below) land the triangle packed by COLUMNS in 16-byte pieces, column j as ceil((64 - j) / 2) pieces from
// piece pair_cc(j) on: 1056 pieces, fetched by 17 LDS-DMA instructions of 64 pieces = 2176 doubles.
constexpr int kSlotD = 17 * 128;
constexpr int kSlot = kSlotE + kSlotD;
// first 16-byte piece of column jj of the landed triangle: sum over j' < jj of ceil((64 - j') / 2)
__host__ __device__ constexpr int pair_cc(int jj) { return 32 * jj - (jj >> 1) * ((jj - 1) >> 1); }
static_assert(pair_cc(0) == 0 && pair_cc(1) == 32 && pair_cc(2) == 64 && pair_cc(3) == 95 && pair_cc(64) == 1056, "pieces");
constexpr int kTeamLds = 2 * kB + kB + 4 * kB + 8;   // vbuf[2][64], u[64], red[256], tau, beta (+ pad)
constexpr size_t kPairLdsBytes = sizeof(double) * (3 * kSlot + 2 * kTeamLds);

// Diagnostic build (-DPAIR_STAMPS): shader cycles of thread 0 between the barriers of every step, summed over all steps
// of all workgroups since the last read (sc_dbg_pair_stamps, tools/pair_stamps.py); nothing of it in the normal build.
// (round 6: the sums are kept in LDS -- one writer per entry: thread 0 [0..15], thread 256 [16..31], lane 0 of the first
// loader wave [32..47] -- and added to the global table when the thread leaves the kernel: a global atomic per stamp
// more than doubled the stage's time)
#ifdef PAIR_STAMPS
__device__ unsigned long long g_pair_stamps[48];
__shared__ unsigned long long s_pair_stamps[48];
#define PAIR_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");
#define PAIR_ACC(slot, a, b) if (tid == 0) s_pair_stamps[slot] += (b) - (a);
#define PAIR_ACC_B(slot, a, b) if (tid == 256) s_pair_stamps[16 + (slot)] += (b) - (a);
#define PAIR_ACC_L(slot, a, b) if (lw == 0 && lane == 0) s_pair_stamps[32 + (slot)] += (b) - (a);
#define PAIR_FLUSH(first)                                                                         \
  for (int i_ = 0; i_ < 16; ++i_)                                                                 \
    if (s_pair_stamps[(first) + i_]) atomicAdd(&g_pair_stamps[(first) + i_], s_pair_stamps[(first) + i_]);
#else
#define PAIR_STAMP(var)
#define PAIR_ACC(slot, a, b)
#define PAIR_ACC_B(slot, a, b)
#define PAIR_ACC_L(slot, a, b)
#define PAIR_FLUSH(first)
#endif

// The common step of k_bulge_pair: both teams at work on full 64-row blocks, neither at a sweep start.  Same barriers
// and the same arithmetic as the general form inside the kernel, without its masks, clamps and case distinctions -- the
// step is bound by instruction issue, not by flops -- and one instruction stream per team: A (TEAM 0) loads its blocks
// from the band and leaves them in its LDS slot, B (TEAM 1) takes the shifted blocks from the slots and stores to the band.
typedef double __attribute__((address_space(3)))* ldptr;   // LDS: ds_read / ds_write, never flat
struct PairStepArgs {
  double* ab;        // band of this matrix
  double* vd;        // its diamonds
  double* tau2;
  ldptr slots;       // LDS: ring of three position slots
  ldptr vbuf, u, red, sc;   // LDS: this team's vectors
  int* prog;
  int s, k;          // this team's sweep and position
  size_t dia;
  int cc;
  int tid;
  const int* prev;   // the predecessor sweep's counter (null: none, or no early read) and the stop flag: read behind [1]
  const int* ctl;
};

// LOADER = 1: team A's blocks have been landed in its slot by the loader waves (k_bulge_pair's header): E column by column
// ((i, j) at j * 64 + i), the triangle of D by columns in 16-byte pieces (pair_cc) -- A issues no global load.
template <int TEAM, int LOADER>
__device__ __forceinline__ void pair_step_full(const PairStepArgs& P, double& tau_p, int& pend_k, int& h_have, int& h_stop) {
  const int tid = P.tid, tt = tid & 255, i = tid & 63;
  const int q = LOADER ? __builtin_amdgcn_readfirstlane((tid >> 6) & 3) : (tid >> 6) & 3;
  const int k = P.k, s = P.s;
  const int r0 = s + 1 + k * kB;
  ldptr vp = P.vbuf + ((k + 1) & 1) * kB;
  ldptr vn = P.vbuf + (k & 1) * kB;
  ldptr Eimg = P.slots + (k % 3) * kSlot;   // A: its slot; B: the slot it consumes, then its scratch image
  ldptr Dimg = Eimg + kSlotE;
  ldptr red = P.red;
  ldptr u = P.u;
  gdptr colbase_k = wave_uniform(P.ab + (size_t)r0 * kLdab);
  gdptr ebase_k = wave_uniform(P.ab + (size_t)(r0 - kB) * kLdab);
  const unsigned o_e = (unsigned)(kB + i + q * 16 * (kLdab - 1));   // E(i, 16 q + c) at ebase[o_e + c (kLdab - 1)]
  const unsigned o_d = (unsigned)(i + q * 16 * (kLdab - 1));        // D(i, 16 q + c) at colbase[o_d + c (kLdab - 1)]
  double d16[16], t16[16], vr[16];
  PAIR_STAMP(f0)
  if (TEAM == 0) {
    if (LOADER) {
      const ldptr ecol = Eimg + q * 16 * kB + i;
#pragma unroll
      for (int c = 0; c < 16; ++c) t16[c] = ecol[c * kB];
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) t16[c] = ld_l2(ebase_k + (o_e + (unsigned)(c * (kLdab - 1))));
    }
  } else {
    const ldptr En = P.slots + ((k + 1) % 3) * kSlot;
    const int i1 = min(i + 1, kB - 1);
    const ldptr rowD = i < kB - 1 ? Dimg + i1 * (i1 + 1) / 2 : En;
    const ldptr rowE = Eimg + i1 * (kB + 1) + 1;
#pragma unroll
    for (int c = 0; c < 16; ++c) d16[c] = rowD[q * 16 + c + 1];
#pragma unroll
    for (int c = 0; c < 16; ++c) t16[c] = i < kB - 1 ? rowE[q * 16 + c] : 0.0;
    if (q == 3) {
      const double dcol = Dimg[i1 * (i1 + 1) / 2];
      t16[15] = i < kB - 1 ? dcol : En[0];
      if (i == kB - 1) d16[15] = En[kSlotE];
    }
    PAIR_STAMP(g1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous step's stores are in the L2
    PAIR_STAMP(g2)
    PAIR_ACC_B(10, f0, g1) PAIR_ACC_B(11, g1, g2)
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) vr[c] = vp[q * 16 + c];
  PAIR_STAMP(fa)
  lds_barrier();                                                                  // [1]
  PAIR_STAMP(f1)
  PAIR_ACC(12, f0, fa)
  if (TEAM == 1) {
    if (tt == 0 && pend_k >= 0) __hip_atomic_store(P.prog + s, pend_k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_k = k;
    // the NEXT step's look at the predecessor pair, requested a step ahead (k_bulge_pair: "early look"): this wave has no
    // load in flight and its stores of this step come later, so the values are there when the step ends
    if (tt == 0 && P.prev) {
      h_have = __hip_atomic_load(P.prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      h_stop = __hip_atomic_load(P.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (!LOADER) {
    // D is not touched before step (5): its loads are issued here, behind the barrier, so that the other team does not
    // wait for their issue (under load the memory pipeline takes a block's 16 load instructions per wave slowly: issuing
    // them costs as much as waiting for the data) and land behind steps (2)-(4).  Entries above the diagonal: whatever
    // lies in front of the column, masked in step (5).
#pragma unroll
    for (int c = 0; c < 16; ++c) d16[c] = ld_l2(colbase_k + (o_d + (unsigned)(c * (kLdab - 1))));
  }
  {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) a += t16[c] * vr[c];
    red[q * kB + i] = a;
  }
  lds_barrier();                                                                  // [2]
  {
    const double ui = tau_p * ((red[i] + red[kB + i]) + (red[2 * kB + i] + red[3 * kB + i]));
    ldptr erow = Eimg + i * (kB + 1) + q * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      t16[c] -= ui * vr[c];
      erow[c] = t16[c];
    }
    if (q == 0) {
      const double x = t16[0];
      const double t2 = wave_sum(i >= 1 ? x * x : 0.0);
      const HH h = householder(__shfl(x, 0), t2);
      vn[i] = i == 0 ? 1.0 : x * h.scale;
      if (i == 0) { P.sc[0] = h.tau; P.sc[1] = h.beta; }
    }
  }
  lds_barrier();                                                                  // [3]
  PAIR_STAMP(f3)
#pragma unroll
  for (int c = 0; c < 16; ++c) vr[c] = vn[q * 16 + c];
  {
    double a = 0.0;
    const ldptr ecol = Eimg + q * 16 * (kB + 1) + i;
#pragma unroll
    for (int c = 0; c < 16; ++c) a += ecol[c * (kB + 1)] * vr[c];
    red[q * kB + i] = a;
  }
  if (TEAM == 0 && LOADER) {
    // the landed triangle (complete since barrier [3]) into the registers, row i: entry (i, jj) is entry i - jj of
    // column jj; above the diagonal whatever lies in front of the column (masked below).  All of it is read before
    // barrier [4], behind which the row-packed image takes the region over.
#pragma unroll
    for (int c = 0; c < 16; ++c) d16[c] = Dimg[2 * pair_cc(q * 16 + c) - (q * 16 + c) + i];
  }
  lds_barrier();                                                                  // [4]
  if (tt < kB) u[tt] = P.sc[0] * ((red[tt] + red[kB + tt]) + (red[2 * kB + tt] + red[3 * kB + tt]));
  {
    ldptr rowp = Dimg + i * (i + 1) / 2 + q * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      d16[c] = q * 16 + c <= i ? d16[c] : 0.0;
      if (q * 16 + c <= i) rowp[c] = d16[c];
    }
  }
  lds_barrier();                                                                  // [5]
  PAIR_STAMP(f5)
  {
    const double vi = vn[i];
#pragma unroll
    for (int c = 0; c < 16; ++c) t16[c] -= vi * u[q * 16 + c];
    if (q == 0) t16[0] = i == 0 ? P.sc[1] : 0.0;
    if (TEAM == 0) {
      ldptr erow = Eimg + i * (kB + 1) + q * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) erow[c] = t16[c];
      if (q == 0 && i >= 1) ebase_k[(unsigned)(kB + i)] = 0.0;   // the annihilated entries, see the kernel's header
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) ebase_k[o_e + (unsigned)(c * (kLdab - 1))] = t16[c];
    }
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int ii = q * 16 + c;
      const double dcolv = Dimg[ii * (ii + 1) / 2 + min(i, ii)];
      a += (d16[c] + (ii > i ? dcolv : 0.0)) * vr[c];     // row half + column half share vn[16 q + c]
    }
    red[q * kB + i] = a;
  }
  lds_barrier();                                                                  // [6]
  PAIR_STAMP(f6)
  const double tau_f = P.sc[0];
  if (tt < kB) {
    const double pp = tau_f * ((red[tt] + red[kB + tt]) + (red[2 * kB + tt] + red[3 * kB + tt]));
    const double dot = wave_sum(pp * vn[tt]);
    u[tt] = pp - 0.5 * tau_f * dot * vn[tt];
  }
  lds_barrier();                                                                  // [7]
  PAIR_STAMP(f7)
  {
    const double vi = vn[i], wi = u[i];
    if (TEAM == 0) {
      ldptr rowp = Dimg + i * (i + 1) / 2 + q * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (q * 16 + c <= i) rowp[c] = d16[c] - vi * u[q * 16 + c] - wi * vr[c];
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (q * 16 + c <= i) colbase_k[o_d + (unsigned)(c * (kLdab - 1))] = d16[c] - vi * u[q * 16 + c] - wi * vr[c];
    }
    double* vd = P.vd + P.dia * kDiaSize + (size_t)P.cc * kG + P.cc;
    if (tt < kB) vd[(size_t)tt * kG] = vn[tt];
    if (tt == 0) P.tau2[P.dia * kG + P.cc] = tau_f;
    tau_p = tau_f;
  }
#ifdef PAIR_STAMPS
  {
    PAIR_STAMP(f8)
    PAIR_ACC(1, f0, f1) PAIR_ACC(2, f1, f3) PAIR_ACC(3, f3, f5) PAIR_ACC(4, f5, f6) PAIR_ACC(5, f6, f7) PAIR_ACC(6, f7, f8)
    PAIR_ACC_B(1, f0, f1) PAIR_ACC_B(2, f1, f3) PAIR_ACC_B(3, f3, f5) PAIR_ACC_B(4, f5, f6) PAIR_ACC_B(5, f6, f7)
    PAIR_ACC_B(13, f7, f8)
  }
#endif
}

// The general step of k_bulge_pair (sweep starts, partial last blocks, steps in which only one team has a position): the
// same barriers as pair_step_full, with every mask and case distinction.  Not inlined: it runs in a few steps per pair
// only, and its registers would otherwise be charged to the common step.
struct PairGenArgs {
  double* ab;
  double* vd;
  double* tau2;
  ldptr slots;
  ldptr vbuf, u, red, sc;
  int* prog;
  int* sweeps_done;
  int n, s, k, my_len, team, to_lds;
  size_t dia0;
  int cc, tid;
  const int* prev;   // (as in PairStepArgs)
  const int* ctl;
};
struct PairCarry {
  double tau_p;
  int pend_k;
  int h_have, h_stop;
};

// (one copy per kernel instantiation: a shared copy would be compiled for the smaller register budget of the two)
template <int LOADER>
__device__ __noinline__ PairCarry pair_step_general(const PairGenArgs P, double tau_p, int pend_k) {
  double* const ab = P.ab;
  const ldptr slots = P.slots, vbuf = P.vbuf, u = P.u, red = P.red, sc = P.sc;
  int* const prog = P.prog;
  const int n = P.n, s = P.s, k = P.k, my_len = P.my_len, team = P.team, cc = P.cc, tid = P.tid;
  const bool to_lds = P.to_lds != 0;
  const size_t dia0 = P.dia0;
  const int tt = tid & 255, i = tid & 63, q = (tid >> 6) & 3;
  const bool active = k >= 0 && k < my_len;
  const bool first = k == 0;
  const bool eph = active && !first;            // this team has an off-diagonal block in this step
  const int kk = active ? k : 0;
  const int r0 = s + 1 + kk * kB;
  const int L = active ? min(kB, n - r0) : 0;
  ldptr vp = vbuf + ((kk + 1) & 1) * kB;   // reflector of this sweep's previous task
  ldptr vn = vbuf + (kk & 1) * kB;
  ldptr Eimg = slots + (kk % 3) * kSlot;   // A: its slot; B: the slot it consumes, then its scratch image
  ldptr Dimg = Eimg + kSlotE;
  const size_t dia = dia0 + kk;
  gdptr colbase_k = wave_uniform(ab + (size_t)r0 * kLdab);                        // AB(r0, r0)
  gdptr ebase_k = wave_uniform(ab + (size_t)(r0 - (first ? 0 : kB)) * kLdab);    // AB(r0 - kB, r0 - kB)

  // ---- (1) the blocks into registers: thread (i, q) holds row i, columns 16 q .. 16 q + 15
  double d16[16], t16[16];
  double xcol = 0.0;   // sweep start (k == 0, first wave of the team): x = AB(s + 1 + i, s)
#pragma unroll
  for (int c = 0; c < 16; ++c) { d16[c] = 0.0; t16[c] = 0.0; }
  if (active && team == 0) {
    // (E first: it is needed at once; D is not touched before step (5), its loads land behind steps (2)-(4))
    if (!first) {
      const int ic = min(i, L - 1);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int jj = q * 16 + c;
        t16[c] = ld_l2(ebase_k + (unsigned)((kB + ic - jj) + jj * kLdab));
      }
    } else if (q == 0) {
      gdptr col_s = wave_uniform(ab + (size_t)s * kLdab);
      xcol = ld_l2(col_s + (unsigned)(1 + min(i, L - 1)));
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int jc = min(q * 16 + c, L - 1);
      const int ic = min(max(i, jc), L - 1);
      d16[c] = ld_l2(colbase_k + (unsigned)((ic - jc) + jc * kLdab));
    }
  } else if (active) {
    // B's block = A's, shifted by (1, 1) (see the header); row 63 comes from row 0 of A's next position.  Whatever is
    // read from a slot that does not exist (A's last position: then L <= 63) is masked out below
    const ldptr En = slots + ((kk + 1) % 3) * kSlot;
    const int i1 = min(i + 1, kB - 1);
    const ldptr rowD = i < kB - 1 ? Dimg + i1 * (i1 + 1) / 2 : En;   // D'(i, jj) = rowD[jj + 1]
    const ldptr rowE = Eimg + i1 * (kB + 1) + 1;                     // E'(i, jj) = rowE[jj]  (i, jj < 63)
#pragma unroll
    for (int c = 0; c < 16; ++c) d16[c] = rowD[q * 16 + c + 1];
    if (!first) {
#pragma unroll
      for (int c = 0; c < 16; ++c) t16[c] = i < kB - 1 ? rowE[q * 16 + c] : 0.0;
    }
    if (q == 3) {   // column 63: E'(i, 63) = D_k(i + 1, 0); corner (63, 63): D_{k+1}(0, 0) and E_{k+1}(0, 0)
      const double dcol = Dimg[i1 * (i1 + 1) / 2];
      if (!first) t16[15] = i < kB - 1 ? dcol : En[0];
      if (i == kB - 1) d16[15] = En[kSlotE];
    }
    if (first && q == 0) xcol = i < kB - 1 ? Dimg[i1 * (i1 + 1) / 2] : En[0];
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) t16[c] = i < L ? t16[c] : 0.0;
  xcol = i < L ? xcol : 0.0;
  // B (and a lone A): the stores of the previous step are in the L2 -- a wait that overlaps A's for its loads
  // (A's own few stores -- zeros, corner, reflector -- are older than its next loads, whose data it waits for: in
  // order; once A has run out of positions it drains like B)
  if (!to_lds || !active) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();                                                                    // [1] B's slot reads are done
  if (tt == 0 && pend_k >= 0) {   // publish the previous step's task
    __hip_atomic_store(prog + s, pend_k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pend_k + 1 == my_len) atomicAdd(P.sweeps_done, 1);
  }
  pend_k = -1;
  int h_have = 0, h_stop = 0;
  if (team == 1 && tt == 0 && P.prev) {   // the next step's early look (see pair_step_full)
    h_have = __hip_atomic_load(P.prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    h_stop = __hip_atomic_load(P.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  
  // ---- (2) u = tau_p E vp: partial sums over this wave's 16 columns
  if (eph) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) a += t16[c] * vp[q * 16 + c];
    red[q * kB + i] = a;
  }
  lds_barrier();                                                                    // [2]
  // ---- (3) E <- E (I - tau_p vp vp^T) in registers + its LDS image; new reflector; image of D (packed lower)
  if (eph) {
    const double ui = tau_p * ((red[i] + red[kB + i]) + (red[2 * kB + i] + red[3 * kB + i]));
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      t16[c] -= ui * vp[q * 16 + c];
      Eimg[i * (kB + 1) + q * 16 + c] = t16[c];
    }
    if (q == 0) xcol = t16[0];
  }
  if (active && q == 0) {
    const double t2 = wave_sum((i >= 1 && i < L) ? xcol * xcol : 0.0);
    const HH h = householder(__shfl(xcol, 0), t2);
    vn[i] = i == 0 ? 1.0 : (i < L ? xcol * h.scale : 0.0);
    if (i == 0) { sc[0] = h.tau; sc[1] = h.beta; }
    if (first && i < L) {   // the annihilated column: AB(s + 1 .., s) = (beta, 0, ..)
      gdptr col_s = wave_uniform(ab + (size_t)s * kLdab);
      col_s[(unsigned)(1 + i)] = i == 0 ? h.beta : 0.0;
    }
  }
  lds_barrier();                                                                    // [3]
    // ---- (4) z_j = sum_i E[i, j] vn_i: thread (j = i, rows 16 q ..)
  if (eph) {
    double a = 0.0;
#pragma unroll
    for (int ii = q * 16; ii < q * 16 + 16; ++ii) a += Eimg[ii * (kB + 1) + i] * vn[ii];
    red[q * kB + i] = a;
  }
  lds_barrier();                                                                    // [4]
  if (eph && tt < kB) u[tt] = sc[0] * ((red[tt] + red[kB + tt]) + (red[2 * kB + tt] + red[3 * kB + tt]));
  // image of D (packed lower) for the column half of p = D vn; A's D loads have had steps (2)-(4) to land
#pragma unroll
  for (int c = 0; c < 16; ++c) d16[c] = (i < L && q * 16 + c <= i) ? d16[c] : 0.0;
  if (active) {
    ldptr rowp = Dimg + i * (i + 1) / 2 + q * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (q * 16 + c <= i) rowp[c] = d16[c];
  }
  lds_barrier();                                                                    // [5]
    // ---- (6) E <- H E, first column = beta e1: to LDS (A) or to the band (B); then p = D vn, row half from the
  // registers, strictly-lower (column) half from the packed image
  if (eph) {
    const double vi = vn[i], beta = sc[1];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int jj = q * 16 + c;
      double e = t16[c] - vi * u[jj];
      if (jj == 0) e = i == 0 ? beta : 0.0;
      if (to_lds) {
        Eimg[i * (kB + 1) + jj] = i < L ? e : 0.0;
        // the annihilated entries (first column below its first row) lie outside every block of sweep sB: B's shifted
        // blocks never write them back, so A zeroes them in the band itself
        if (jj == 0 && i >= 1 && i < L) ebase_k[(unsigned)(kB + i)] = 0.0;
      } else if (i < L) {
        ebase_k[(unsigned)((kB + i - jj) + jj * kLdab)] = e;
      }
    }
  }
  if (active) {
    double a = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) a += d16[c] * vn[q * 16 + c];          // (entries above the diagonal are zero)
#pragma unroll
    for (int ii = q * 16; ii < q * 16 + 16; ++ii) {
      const double dv = Dimg[ii * (ii + 1) / 2 + min(i, ii)];
      a += ii > i ? dv * vn[ii] : 0.0;
    }
    red[q * kB + i] = a;
  }
  lds_barrier();                                                                    // [6]
    const double tau_now = sc[0];
  if (active && tt < kB) {
    const double pp = tau_now * ((red[tt] + red[kB + tt]) + (red[2 * kB + tt] + red[3 * kB + tt]));
    const double dot = wave_sum(pp * vn[tt]);
    u[tt] = pp - 0.5 * tau_now * dot * vn[tt];
  }
  lds_barrier();                                                                    // [7]
    // ---- (8) D <- D - vn w^T - w vn^T: to LDS (A) or to the band (B); the reflector goes into its diamond
  if (active) {
    const double vi = vn[i], wi = u[i];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int jj = q * 16 + c;
      if (jj <= i) {
        const double val = d16[c] - vi * u[jj] - wi * vn[jj];
        if (to_lds) {
          Dimg[i * (i + 1) / 2 + jj] = i < L ? val : 0.0;
          // row 0 of a block is consumed by B's task one position up -- which does not exist for A's first block:
          // its corner AB(sA + 1, sA + 1) is final already and goes to the band here
          if (first && i == 0) colbase_k[0] = val;
        } else if (i < L) {
          colbase_k[(unsigned)((i - jj) + jj * kLdab)] = val;
        }
      }
    }
    double* vd = P.vd + dia * kDiaSize + (size_t)cc * kG + cc;
    if (tt < L) vd[(size_t)tt * kG] = vn[tt];
    if (tt == 0) P.tau2[dia * kG + cc] = tau_now;
    tau_p = tau_now;
    if (!to_lds) pend_k = k;   // published once the stores have drained: behind the next step's loads
  }
  return PairCarry{tau_p, pend_k, h_have, h_stop};
}

// (the few fields of the band workspace's layout the kernel needs, instead of all of SbLayout in scalar registers)
struct PairLayout {
  int n;
  long long slab, ab, vd, tau2;
};

// ---- Loader waves (round 6, LOADER = 1: 768 threads).  In the form above only team A's 256 threads issue global loads,
// in one of the step's eight phases, and wait for them in the next: the memory system has this CU's requests in flight
// for a fraction of the step.  Here waves 8-11 issue nothing but LDS-DMA (global_load_lds_dwordx4: 64 x 16 bytes from
// per-lane addresses to one contiguous KB of LDS, no VGPR for the data, no VALU instruction in steady state) and fetch
// team A's blocks of position m + 1 while the compute waves are in step m:
//   * where to: the slot of position m + 1 is the slot of position m - 2, which team B emptied at the start of step m and
//     uses as its scratch image -- the E image until barrier [4], the D image until barrier [6].  So E(m + 1) is requested
//     behind [4] (32 instructions: two columns each, 8 per loader wave behind ONE write of M0) and the triangle of
//     D(m + 1) behind [7] (17 instructions of 64 pieces, see kSlotD; the second M0 write of a wave waits for its E pieces
//     in flight -- behind [7] nobody waits for the loader before the next [0], which needs E landed anyway);
//   * handshake: a loader wave arrives at the next step's barrier [0] with its E pieces landed (s_waitcnt vmcnt(#D)) and
//     at barrier [3] with everything landed.  Team A reads E from the slot in front of [1] (column-major as landed: row
//     accesses are conflict-free), re-writes it row-major behind [2] as before; it reads its rows of D between [3] and
//     [4] and writes the row-packed image behind [4] as before;
//   * only the common step is served (both teams on full blocks): the loaders evaluate the same predicate for m + 1 as
//     the compute waves, the general step loads for itself as before;
//   * the predecessor pair must have left position m + 1 by the time it is fetched: thread 0's wait in front of step m
//     covers m + 3 instead of m + 2 of its tasks when a fetch follows;
//   * only the triangle of D crosses the fabric (17 KB instead of the 32 KB square whose upper half was masked).
// The loader waves execute every workgroup barrier of the compute waves, in the same number.
__device__ __forceinline__ bool pair_common_step(bool hasB, int m, int lenA, int sA, int n) {
  return hasB && m >= 3 && m < lenA && sA + 1 + (m + 1) * kB <= n;
}

// The loader waves of k_bulge_pair<1> (its header): the same claims, steps and workgroup barriers as the compute waves.
__device__ __forceinline__ void pair_loader_run(double* __restrict__ sb_all, const PairLayout& SL, int nxcd, int xcd, int slot,
                                                int mpx, const volatile __attribute__((address_space(3))) int* s_claim,
                                                const volatile __attribute__((address_space(3))) int* s_go,
                                                unsigned lds_base, int lw, int lane) {
  const int n = SL.n;
  // per-lane byte offsets of this wave's pieces, once per kernel (nothing of it changes from step to step).
  // E: wave lw fetches column pairs 8 lw .. 8 lw + 7; instruction x lands at M0 + (1024 x - 3584) + 16 lane; lanes 0-31
  // take rows (2 l, 2 l + 1) of the even column, lanes 32-63 those of the odd one (its pieces start 127 doubles later).
  // D: wave lw fetches instructions d0 .. d0 + nd - 1 of the 17 (5, 4, 4, 4), instruction x lands at M0 + (1024 x - 2048)
  // + 16 lane; piece g = 64 (d0 + x) + lane is piece g - pair_cc(j) of column j; pieces beyond the 1056 re-read the last.
  const int ld_d0 = lw == 0 ? 0 : 1 + 4 * lw;
  unsigned voff_e[8], voff_d[5];
  {
    const unsigned lane_off = lane < 32 ? 16u * lane : (unsigned)(8 * (kLdab - 1)) + 16u * (lane - 32);
#pragma unroll
    for (int x = 0; x < 8; ++x) voff_e[x] = lane_off + (unsigned)(16 * (kLdab - 1) * (8 * lw + x)) - (unsigned)(1024 * x) + 3584u;
#pragma unroll
    for (int x = 0; x < 5; ++x) {
      const int g = min(64 * (ld_d0 + x) + lane, pair_cc(kB) - 1);
      int j = 0;
      while (pair_cc(j + 1) <= g) ++j;
      voff_d[x] = (unsigned)(8 * (j * kLdab + 2 * (g - pair_cc(j)))) - (unsigned)(1024 * x) + 2048u;
    }
  }
  int cur = slot % mpx, exhausted = 0;
  while (exhausted < mpx) {
    const int b = xcd + nxcd * cur;
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int sA = 2 * __builtin_amdgcn_readfirstlane(*s_claim);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (sA > n - 3) {
      ++exhausted;
      cur = cur + 1 < mpx ? cur + 1 : 0;
      continue;
    }
    exhausted = 0;
    const unsigned long long ab_u = (unsigned long long)(size_t)wave_uniform(sb_all + (size_t)b * SL.slab + SL.ab);
    const bool hasB = sA + 1 <= n - 3;
    const int lenA = chase_len(n, sA);
    const int nsteps = hasB ? lenA + 2 : lenA;
    for (int m = 0; m < nsteps; ++m) {
      PAIR_STAMP(l0)
      if (pair_common_step(hasB, m, lenA, sA, n)) {
        // this step's E pieces have landed (the D pieces requested after them may still be on their way)
        if (lw == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      PAIR_STAMP(l1)
      __builtin_amdgcn_s_barrier();                                                     // [0]
      PAIR_STAMP(l2)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int go = __builtin_amdgcn_readfirstlane(*s_go);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!go) {
        // give-up: the pieces in flight land in the slot nobody flushes; then the compute waves' barriers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (hasB) __builtin_amdgcn_s_barrier();
        if (lw == 0 && lane == 0) { PAIR_FLUSH(32) }
        return;
      }
      // ---- a step of a loader wave: the compute waves' seven barriers, position m + 1 requested behind [4] and [7]
      const bool fetch = pair_common_step(hasB, m + 1, lenA, sA, n);
      __builtin_amdgcn_s_barrier();                                                     // [1]
      __builtin_amdgcn_s_barrier();                                                     // [2]
      PAIR_STAMP(l3)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this step's D pieces have landed
      PAIR_STAMP(l4)
      __builtin_amdgcn_s_barrier();                                                     // [3]
      __builtin_amdgcn_s_barrier();                                                     // [4]
      PAIR_STAMP(l5)
      const int r0n = sA + 1 + (m + 1) * kB;
      const unsigned slot_b = lds_base + (unsigned)(((m + 1) % 3) * kSlot * 8);
#ifdef PAIR_VAR_NODMA
      if (false) {
#else
      if (fetch) {
#endif
        const unsigned long long eb = ab_u + ((unsigned long long)(r0n - kB) * kLdab + kB) * 8ull;   // E(0, 0) of position m + 1
        const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(slot_b + (unsigned)(1024 * 8 * lw + 3584)));
        asm volatile(
            "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %9 offset:-3584\n\t"
            "global_load_lds_dwordx4 %2, %9 offset:-2560\n\t"
            "global_load_lds_dwordx4 %3, %9 offset:-1536\n\t"
            "global_load_lds_dwordx4 %4, %9 offset:-512\n\t"
            "global_load_lds_dwordx4 %5, %9 offset:512\n\t"
            "global_load_lds_dwordx4 %6, %9 offset:1536\n\t"
            "global_load_lds_dwordx4 %7, %9 offset:2560\n\t"
            "global_load_lds_dwordx4 %8, %9 offset:3584"
            :
            : "s"(m0v), "v"(voff_e[0]), "v"(voff_e[1]), "v"(voff_e[2]), "v"(voff_e[3]), "v"(voff_e[4]), "v"(voff_e[5]),
              "v"(voff_e[6]), "v"(voff_e[7]), "s"(eb)
            : "memory");
      }
      PAIR_STAMP(l6)
      __builtin_amdgcn_s_barrier();                                                     // [5]
      __builtin_amdgcn_s_barrier();                                                     // [6]
      __builtin_amdgcn_s_barrier();                                                     // [7]
      PAIR_STAMP(l7)
#ifdef PAIR_VAR_NODMA
      if (false) {
#else
      if (fetch) {
#endif
        const unsigned long long db = ab_u + (unsigned long long)r0n * kLdab * 8ull;                  // D(0, 0) of position m + 1
        const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(slot_b + (unsigned)(kSlotE * 8 + 1024 * ld_d0 + 2048)));
        if (lw == 0)
          asm volatile(
              "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
              "global_load_lds_dwordx4 %1, %6 offset:-2048\n\t"
              "global_load_lds_dwordx4 %2, %6 offset:-1024\n\t"
              "global_load_lds_dwordx4 %3, %6\n\t"
              "global_load_lds_dwordx4 %4, %6 offset:1024\n\t"
              "global_load_lds_dwordx4 %5, %6 offset:2048"
              :
              : "s"(m0v), "v"(voff_d[0]), "v"(voff_d[1]), "v"(voff_d[2]), "v"(voff_d[3]), "v"(voff_d[4]), "s"(db)
              : "memory");
        else
          asm volatile(
              "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
              "global_load_lds_dwordx4 %1, %5 offset:-2048\n\t"
              "global_load_lds_dwordx4 %2, %5 offset:-1024\n\t"
              "global_load_lds_dwordx4 %3, %5\n\t"
              "global_load_lds_dwordx4 %4, %5 offset:1024"
              :
              : "s"(m0v), "v"(voff_d[0]), "v"(voff_d[1]), "v"(voff_d[2]), "v"(voff_d[3]), "s"(db)
              : "memory");
      }
#ifdef PAIR_STAMPS
      if (fetch && pair_common_step(hasB, m, lenA, sA, n)) {
        PAIR_STAMP(l8)
        PAIR_ACC_L(0, l0, l1) PAIR_ACC_L(1, l1, l2) PAIR_ACC_L(2, l2, l3) PAIR_ACC_L(3, l3, l4) PAIR_ACC_L(4, l4, l5)
        PAIR_ACC_L(5, l5, l6) PAIR_ACC_L(6, l6, l7) PAIR_ACC_L(7, l7, l8) PAIR_ACC_L(8, 0ull, 1ull)
      }
#endif
    }
    // ---- end of the pair (nothing is in flight: the last steps fetch nothing)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (lw == 0 && lane == 0) { PAIR_FLUSH(32) }
}

// (diagnostic variants, tools/r06_pair_variants.sh: PAIR_VAR_CAP168 compiles the 512-thread form for the register budget
// of the 768-thread one; PAIR_VAR_NODMA keeps the loader waves and their barriers but requests nothing -- wrong results)
#ifdef PAIR_VAR_CAP168
#define PAIR_BOUNDS(LOADER) 768
#else
#define PAIR_BOUNDS(LOADER) (LOADER ? 768 : 512)
#endif
template <int LOADER>
__global__ __launch_bounds__(PAIR_BOUNDS(LOADER), 1) void k_bulge_pair(double* __restrict__ sb_all, PairLayout SL, int batch, int W, int nxcd,
                                                      int* __restrict__ progress, int* __restrict__ next_pair,
                                                      int* __restrict__ ctl, int give_up_after, int early_look) {
  extern __shared__ __attribute__((aligned(16))) double pair_lds[];
  __shared__ int s_go, s_claim, s_xcd, s_slot;

  const int n = SL.n;
  const int tid = threadIdx.x;
  // (the team is uniform over a wave: told to the compiler, so that the position, the step's case and the branches on
  // them are scalar)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >= 8 ? 0 : wave >> 2;
  const int tt = tid & 255;
  double* const slots = pair_lds;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) double*)pair_lds;
  double* const tl = pair_lds + 3 * kSlot + team * kTeamLds;
  double* const vbuf = tl;
  double* const u = tl + 2 * kB;
  double* const red = tl + 3 * kB;
  double* const sc = tl + 7 * kB;   // [0] tau, [1] beta of the reflector being generated

  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    s_xcd = (int)(id & 7u) % nxcd;
    s_slot = atomicAdd(ctl + 2 + s_xcd, 1);
  }
#ifdef PAIR_STAMPS
  if (tid < 48) s_pair_stamps[tid] = 0;
#endif
  __syncthreads();
  const int xcd = s_xcd, slot = s_slot;
  const int mpx = xcd < batch ? (batch - xcd + nxcd - 1) / nxcd : 0;   // matrices of this XCD: xcd, xcd + nxcd, ...
  if (mpx == 0 || slot >= mpx * W) return;
  if (LOADER && wave >= 8) {
    pair_loader_run(sb_all, SL, nxcd, xcd, slot, mpx, (const volatile __attribute__((address_space(3))) int*)&s_claim,
                    (const volatile __attribute__((address_space(3))) int*)&s_go, lds_base, wave - 8, tid & 63);
    return;
  }
  const int K0 = chase_len(n, 0);
  int steps_left = give_up_after;

  int cur = slot % mpx, exhausted = 0;
  while (exhausted < mpx) {
    const int b = xcd + nxcd * cur;
    int* prog = progress + (size_t)b * n;
    if (tid == 0) s_claim = atomicAdd(next_pair + b, 1);
    __syncthreads();
    const int sA = 2 * __builtin_amdgcn_readfirstlane(s_claim);
    __syncthreads();
    if (sA > n - 3) {
      ++exhausted;
      cur = cur + 1 < mpx ? cur + 1 : 0;
      continue;
    }
    exhausted = 0;
    double* sb = sb_all + (size_t)b * SL.slab;
    double* ab = sb + SL.ab;
    const bool hasB = sA + 1 <= n - 3;
    const int lenA = chase_len(n, sA), lenB = hasB ? chase_len(n, sA + 1) : 0;
    const int s = sA + team;                 // this team's sweep
    const int my_len = team ? lenB : lenA;
    const int S = s / kG, cc = s - S * kG;
    const size_t dia0 = (size_t)S * K0 - (size_t)S * (S - 1) / 2;
    const int len_prev = sA > 0 ? chase_len(n, sA - 1) : 0;
    const bool to_lds = team == 0 && hasB;   // A's results stay in LDS (a lone last sweep writes to memory itself)
    double tau_p = 0.0;

    const int nsteps = hasB ? lenA + 2 : lenA;
    // ---- the look at the predecessor pair (A's dependence; B's are inside the workgroup), by thread 256: the first wave of
    // team B opens a step with LDS reads only.  Round 6, "early look": followers run at the dependence's limit -- pair p + 1
    // starts when its workgroup has finished pair p - 3, a few steps behind pair p, and both advance at the same rate -- so a
    // look taken when the step begins was needed in 0.6 of all steps and cost 2 300 cycles of a 16 900-cycle step, all
    // other waves waiting at [0] (profiles/r06_pair_stamps.txt).  Now the counter and the stop flag are requested behind
    // barrier [1] of the PREVIOUS step and are in a register when this step begins; a value that does not cover the step
    // falls back to polling, for one task more than needed, which buys the step of distance that keeps the early values
    // sufficient from there on.
    int have_c = 0;      // thread 256: last value seen of the predecessor's counter (it only grows)
    int h_have = 0, h_stop = 0;
    const int* prev = (early_look && sA > 0) ? prog + sA - 1 : nullptr;
    int pend_k = -1;     // task of this team whose stores are on their way: published behind the next step's loads
    for (int m = 0; m < nsteps; ++m) {
      PAIR_STAMP(ts0)
      if (tid == 256) {
        // (a step whose successor is fetched ahead by the loader waves needs the predecessor one task further)
        const int ahead = LOADER && pair_common_step(hasB, m + 1, lenA, sA, n) ? 3 : 2;
        const int need = (sA > 0 && m < lenA) ? min(m + ahead, len_prev) : 0;
        int go = 1;
        if (prev) {
          have_c = max(have_c, h_have);
          if (h_stop) go = 0;
        }
        if (go && (have_c < need || (!prev && (m & 7) == 0))) {
#ifdef PAIR_STAMPS
          s_pair_stamps[16 + 14] += 1ull;
#endif
          const int want = prev ? min(need + 1, len_prev) : need;
          const int stop0 = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int have = sA > 0 ? __hip_atomic_load(prog + sA - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
          if (stop0) go = 0;
          long spins = 0;
          while (go && have < want) {
            // (2^21 polls of ~1 us: seconds, orders of magnitude above any wait for a running workgroup)
            ++spins;
            if (spins > (1L << 21) ||
                ((spins & 15) == 0 && __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
              if (!__hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                ctl[10] = b; ctl[11] = sA; ctl[12] = m;
              }
              __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              go = 0;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            have = __hip_atomic_load(prog + sA - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          have_c = have;
        }
        s_go = go;
      }
      PAIR_STAMP(tsp)
      lds_barrier();   // [0] (also: A's last slot writes of the previous step are ordered before B's reads below)
      if (!s_go) {
        // ---- give up between steps.  First the pending publish (its stores drained), then: A has finished positions
        // < min(m, lenA), B positions < m - 2; what A left for B (positions m - 2 and m - 1) goes back to the band, and
        // both counts are published
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tt == 0 && pend_k >= 0) __hip_atomic_store(prog + s, pend_k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hasB) {
          for (int pos = max(m - 2, 0); pos < min(m, lenA); ++pos) {
            const double* Es = slots + (pos % 3) * kSlot;
            const double* Ds = Es + kSlotE;
            const int r0p = sA + 1 + pos * kB;
            const int Lp = min(kB, n - r0p);
            // (row 0 of the older slot has been consumed and rewritten in the band by B's task one position up)
            const int i_lo = (pos == m - 2 && pos >= 1) ? 1 : 0;
            if (pos > 0)
              for (int e = tid; e < kB * kB; e += 512) {
                const int ii = e & 63, jj = e >> 6;
                if (ii >= i_lo && ii < Lp) ab[(size_t)(r0p - kB + jj) * kLdab + (kB + ii - jj)] = Es[ii * (kB + 1) + jj];
              }
            for (int e = tid; e < kB * kB; e += 512) {
              const int ii = e & 63, jj = e >> 6;
              if (ii >= jj && ii >= i_lo && ii < Lp) ab[(size_t)(r0p + jj) * kLdab + (ii - jj)] = Ds[ii * (ii + 1) / 2 + jj];
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) {
            __hip_atomic_store(prog + sA, min(m, lenA), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(prog + sA + 1, min(max(m - 2, 0), lenB), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (tid == 0) { PAIR_FLUSH(0) }
        if (tid == 256) { PAIR_FLUSH(16) }
        return;
      }

      PAIR_STAMP(ts1)
      const int k = team ? m - 2 : m;
      // ---- the common step: both teams at work on full 64-row blocks, neither at a sweep start (pair_step_full)
      if (pair_common_step(hasB, m, lenA, sA, n)) {
        PairStepArgs pa{ab, sb + SL.vd, sb + SL.tau2, (ldptr)slots, (ldptr)vbuf, (ldptr)u, (ldptr)red, (ldptr)sc, prog, s, k,
                        dia0 + k, cc, tid, prev, ctl};
        if (team == 0) pair_step_full<0, LOADER>(pa, tau_p, pend_k, h_have, h_stop);
        else pair_step_full<1, LOADER>(pa, tau_p, pend_k, h_have, h_stop);
#ifdef PAIR_STAMPS
        if (tid == 0) { s_pair_stamps[9] += 1ull; s_pair_stamps[0] += ts1 - ts0; }
        if (tid == 256) { s_pair_stamps[16] += ts1 - ts0; s_pair_stamps[16 + 7] += tsp - ts0; }
#endif
      } else {
        PairGenArgs ga{ab, sb + SL.vd, sb + SL.tau2, (ldptr)slots, (ldptr)vbuf, (ldptr)u, (ldptr)red, (ldptr)sc, prog,
                       ctl + 16 + xcd, n, s, k, my_len, team, to_lds ? 1 : 0, dia0, cc, tid, prev, ctl};
        const PairCarry pc = pair_step_general<LOADER>(ga, tau_p, pend_k);
        tau_p = pc.tau_p;
        pend_k = pc.pend_k;
        h_have = pc.h_have;
        h_stop = pc.h_stop;
      }
      // (no barrier here: the next step's [0] orders this step's slot writes before their readers, and nothing else of
      // the next step touches LDS before it)
#ifdef PAIR_STAMPS
      if (tid == 0) s_pair_stamps[8] += 1ull;
#endif
      if (tid == 0) {
        if (to_lds && m + 1 == lenA) atomicAdd(ctl + 16 + xcd, 1);   // (A's last task; its count is published below)
        // test hook (sc_dbg_set_chase): raise the time-out flag after that many steps of this workgroup
        if (give_up_after > 0 && --steps_left == 0) {
          ctl[1] = give_up_after;
          __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // ---- end of the pair: the last task's stores, then its count; sweep sA lived in LDS until B had consumed it -- its
    // count only matters to a take-over and is complete now
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tt == 0 && pend_k >= 0) {
      __hip_atomic_store(prog + s, pend_k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (pend_k + 1 == my_len) atomicAdd(ctl + 16 + xcd, 1);
    }
    if (tid == 0 && hasB) __hip_atomic_store(prog + sA, lenA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) { PAIR_FLUSH(0) }
  if (tid == 256) { PAIR_FLUSH(16) }
}

// ================================================================================================================
// Z <- Q2 Z (k_dia_tfactor2 + k_bt2_apply; tools/models/bt2_model.py is their NumPy specification).
//
// The f64 matrix pipe issues one 16x16x4 MFMA per 64 cycles (profiles/r02_probe_clock.txt).  The columns of Z are
// independent, so ONE launch lets every workgroup take 16 NW columns through ALL diamonds in order (sweep groups last to
// first, chase positions first to last) with no synchronisation between workgroups:
//   * the 128-row window of Z lives in the accumulators as Z itself (rows x columns), split over the waves by COLUMNS:
//     a wave owns 16 columns x 128 rows = 8 accumulator tiles.  An accumulator register of this MFMA has exactly the lane
//     layout of a B operand (k = lane >> 4 <-> tile row 4 r + k, j = lane & 15), so  W = V^T Z  takes its B operands
//     straight from the window registers and  Z -= (V T) W  takes them straight from the W accumulators: no LDS
//     round trip for Z or W; going from one chase position to the next the window slides by 64 rows (64 finished rows
//     are stored, 64 new ones loaded): Z is read and written once per sweep group;
//   * a diamond (64 sweeps at one chase position) is applied as FOUR compact-WY blocks of 16 sweeps ("minis", sweep tile
//     st = 3, 2, 1, 0: Q = Q_0 Q_1 Q_2 Q_3, the last acts first).  The reflectors of a mini span rows 16 st .. 16 st + 78
//     = 5 row tiles for V^T Z and for (V T) W alike: 4 x (20 + 20) = 160 MFMAs per diamond and 16 columns, 1.25 x the
//     algorithmic flops.  (Round 2 applied all 64 sweeps as one block: 80 + 104 MFMAs = 1.44 x, because V T of a
//     64-sweep block is a trapezoid rather than a parallelogram.)
//   * the A operands (V^T and -(V T)) are the same for every wave and every column chunk: k_dia_tfactor2 writes them
//     once per diamond as ready-made MFMA fragments (512 B = one wave-wide ds_read_b64 each) in exactly the order the
//     products consume them, 160 per diamond = 80 KB;
//   * a workgroup streams the fragments into a ring of three half-diamond buffers (40 KB each) with LDS-DMA
//     (global_load_lds_dwordx4, no staging registers), two halves ahead of the MFMAs; one barrier per half.
#ifndef BT2_DBG
#define BT2_DBG 0
#endif
// Diagnostic build (-DBT2_STAMPS): per wave, the shader cycles between eight points of the diamond loop are summed and left
// in g_bt2_stamps (read with sc_dbg_bt2_stamps, tools/bt2_stamps.py); no stamp executes in the normal build.
#ifdef BT2_STAMPS
__device__ unsigned long long g_bt2_stamps[64 * 8 * 17];
#define BT2_STAMP_DECL unsigned long long st_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0, st_n = 0;
#define BT2_STAMP(i)                                                              \
  {                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                            \
    unsigned long long t_;                                                        \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");   \
    if ((i) != 0 || st_n != 0) st_sum[i] += t_ - st_prev;                         \
    st_prev = t_;                                                                 \
    if ((i) == 15) ++st_n;                                                        \
    __builtin_amdgcn_sched_barrier(0);                                            \
  }
#define BT2_STAMP_WRITE                                                           \
  if (blockIdx.x < 64 && lane == 0) {                                             \
    for (int i = 0; i < 16; ++i) g_bt2_stamps[(blockIdx.x * 8 + (w & 7)) * 17 + i] = st_sum[i]; \
    g_bt2_stamps[(blockIdx.x * 8 + (w & 7)) * 17 + 16] = st_n;                    \
  }
#else
#define BT2_STAMP_DECL
#define BT2_STAMP(i)
#define BT2_STAMP_WRITE
#endif
// Diagnostic build (-DBT2_CLOCK): the shader clock while k_bt2_apply / k_bt2_role run -- wave 0 of workgroup 0 reads
// s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end (sc_dbg_bt2_clock, tools/bt2_clock.py).
#ifdef BT2_CLOCK
__device__ unsigned long long g_bt2_clk[2];
#define BT2_CLOCK_BEGIN                                                                                      \
  unsigned long long ck0_ = 0, rt0_ = 0;                                                                     \
  if (blockIdx.x == 0 && threadIdx.x < 64)                                                                   \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck0_), "=s"(rt0_)::"memory");
#define BT2_CLOCK_END                                                                                        \
  if (blockIdx.x == 0 && threadIdx.x < 64) {                                                                 \
    unsigned long long ck1_, rt1_;                                                                           \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck1_), "=s"(rt1_)::"memory"); \
    if (threadIdx.x == 0) { g_bt2_clk[0] = ck1_ - ck0_; g_bt2_clk[1] = rt1_ - rt0_; }                        \
  }
#else
#define BT2_CLOCK_BEGIN
#define BT2_CLOCK_END
#endif
// Diagnostic build (-DBT2_TRACE): s_memtime in front of every MFMA of ONE diamond (workgroup 0, sweep group
// ngroups / 2, fourth chase position), every wave: g_bt2_trace[wave][half][step] (sc_dbg_bt2_trace, tools/bt2_trace.py).
#ifdef BT2_TRACE
__device__ unsigned long long g_bt2_trace[8 * 2 * 81];
#define BT2_TRACE_POINT(H, f)                                                                       \
  if (trace_on) {                                                                                   \
    unsigned long long t_;                                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
    if (lane == 0) g_bt2_trace[((w & 7) * 2 + (H)) * 81 + (f)] = t_;                                \
  }
#else
#define BT2_TRACE_POINT(H, f)
#endif
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kMini = 16;                          // sweeps per compact-WY block
constexpr int kMiniFrags = 40;                     // fragments per mini: 20 of V^T, 20 of -(V T)
constexpr int kDiaFrags = 4 * kMiniFrags;          // 160 per diamond
constexpr int kHalfFrags = 2 * kMiniFrags;         // a half-diamond (two minis) is the unit of the LDS ring
constexpr int kHalfDoubles = kHalfFrags * 64;      // 5120 doubles = 40 KB
constexpr int kFragDoubles = kDiaFrags * 64;       // 10240 doubles = 80 KB
// Fragment index inside a diamond = issue index: minis in the order st = 3, 2, 1, 0; inside a mini first the 20 steps
// of W = V_st^T Z (j = 4 (rt - st) + r: Z tile rt register r is the B operand), then the 20 of Z -= (V_st T_st) W
// (j = 5 r + (rt - st): W register r is the B operand, the five row tiles take turns).
__host__ __device__ constexpr int mf_p1(int st, int rt, int r) { return (3 - st) * kMiniFrags + 4 * (rt - st) + r; }
__host__ __device__ constexpr int mf_p2(int st, int rt, int r) { return (3 - st) * kMiniFrags + 20 + 5 * r + (rt - st); }
static_assert(mf_p2(0, 4, 3) == kDiaFrags - 1 && mf_p1(3, 3, 0) == 0, "fragment order");
// Where lane l's value of fragment f lives inside a diamond's block: the two fragments of an even / odd pair of steps
// side by side, so that a lane fetches both with ONE 16-byte LDS read (ds_read_b128 runs at the full LDS rate, 8-byte
// reads at half of it -- and the fragment reads are most of the kernel's LDS traffic).
__host__ __device__ constexpr int frag_off(int f, int l) { return ((f >> 1) * 64 + l) * 2 + (f & 1); }

typedef const double __attribute__((address_space(1)))* zptr_c;   // global_load / global_store, never flat
typedef double __attribute__((address_space(1)))* zptr;
typedef const void __attribute__((address_space(1)))* gvoid_c;
typedef void __attribute__((address_space(3)))* lvoid;

// acc += A B over k-steps [k4_lo, k4_hi): lane (fr, fk) supplies A[i = fr][k = 4 k4 + fk] = a(fr, k) and
// B[k][j = fr] = b(k, fr); acc[r] is D[i = 4 r + fk][j = fr].
template <class FA, class FB>
__device__ __forceinline__ d4 mma_range(d4 acc, int k4_lo, int k4_hi, int fr, int fk, FA a, FB b) {
  for (int k4 = k4_lo; k4 < k4_hi; ++k4) {
    const int k = 4 * k4 + fk;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a(fr, k), b(k, fr), acc, 0, 0, 0);
  }
  return acc;
}

// Per diamond and mini: T = (diag(1 / tau) + striu(V^T V))^-1 of the mini's 16 reflectors (the dlarft recurrence is the
// back substitution for this inverse; reflectors with tau = 0 are decoupled), then the fragments of V^T and -(V T)^T.
// Gram blocks and (V T)^T on the matrix cores; (V T)^T is computed transposed because an accumulator register of the
// transposed tile IS the fragment (same lane, same order), so it is stored with one coalesced 512-byte write.
__global__ __launch_bounds__(256) void k_dia_tfactor2(double* __restrict__ sb_all, SbLayout SL, int dia0) {
  constexpr int LD = kG + 1;
  __shared__ double Vc[kG * LD];          // Vc[c * LD + i] = V[c + i, c]
  __shared__ double Us[4][kMini * 17];    // per mini: U[a * 17 + b], a <= b
  __shared__ double Ts[4][kMini * 17];    // per mini: T[a * 17 + b]
  __shared__ double tau_s[kG];
  double* sb = sb_all + (size_t)blockIdx.y * SL.slab;
  const size_t dia = (size_t)dia0 + blockIdx.x;
  const double* vd = sb + SL.vd + dia * kDiaSize;
  const double* tau = sb + SL.tau2 + dia * kG;
  double* frag = sb + SL.frag + dia * kFragDoubles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  {
    const int c = tid & 63, q = tid >> 6;   // lanes along the sweeps (contiguous in memory)
    for (int i = q; i < kB; i += 4) Vc[c * LD + i] = vd[(size_t)(c + i) * kG + c];
    if (tid < kG) tau_s[tid] = tau[tid];
  }
  __syncthreads();
  auto V = [&](int row, int c) -> double {
    const int i = row - c;
    return (unsigned)i < (unsigned)kB ? Vc[c * LD + i] : 0.0;
  };
  // ---- U of mini st = w: strictly upper part of its Gram block (rows 16 st .. 16 st + 78), 1 / tau on the diagonal
  {
    const int st = w;
    const d4 g = mma_range(d4{0, 0, 0, 0}, 4 * st, 4 * st + 20, fr, fk,
                           [&](int i, int k) { return V(k, 16 * st + i); }, [&](int k, int j) { return V(k, 16 * st + j); });
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = 4 * r + fk, b = fr;
      const double ta = tau_s[16 * st + a], tb = tau_s[16 * st + b];
      double u = 0.0;
      if (a < b) u = (ta != 0.0 && tb != 0.0) ? g[r] : 0.0;
      else if (a == b) u = ta != 0.0 ? 1.0 / ta : 1.0;
      Us[st][a * 17 + b] = u;
    }
  }
  __syncthreads();
  // ---- lane (st, j) of the first wave solves U_st x = e_j by back substitution
  if (w == 0) {
    const int st = lane >> 4, j = lane & 15;
    const double* Ub = Us[st];
    double x[16];
#pragma unroll
    for (int i = 15; i >= 0; --i) {
      double s = i == j ? 1.0 : 0.0;
#pragma unroll
      for (int l = i + 1; l < 16; ++l) s -= Ub[i * 17 + l] * x[l];
      x[i] = i <= j ? s / Ub[i * 17 + i] : 0.0;
    }
    if (tau_s[16 * st + j] == 0.0) x[j] = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) Ts[st][i * 17 + j] = x[i];
  }
  __syncthreads();
  // ---- fragments of V^T: lane l holds V[16 rt + 4 r + (l >> 4)][16 st + (l & 15)]
  for (int idx = tid; idx < 4 * 20 * 64; idx += 256) {
    const int l = idx & 63, f = idx >> 6;
    const int st = f / 20, j = f % 20, rt = st + j / 4, r = j % 4;
    frag[frag_off(mf_p1(st, rt, r), l)] = V(16 * rt + 4 * r + (l >> 4), 16 * st + (l & 15));
  }
  // ---- fragments of -(V T): tile (st, rt) transposed,  D'[sweep i][row j] = sum_l T_st[l][i] V[16 rt + j][16 st + l]
  for (int q = w; q < 20; q += 4) {
    const int st = q / 5, rt = st + q % 5;
    const d4 d = mma_range(d4{0, 0, 0, 0}, 0, 4, fr, fk, [&](int i, int k) { return Ts[st][k * 17 + i]; },
                           [&](int k, int j) { return V(16 * rt + j, 16 * st + k); });
#pragma unroll
    for (int r = 0; r < 4; ++r) frag[frag_off(mf_p2(st, rt, r), lane)] = -d[r];
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_bt2_apply(const double* __restrict__ sb_all, SbLayout SL,
                                                          const int* __restrict__ dia_off, double* __restrict__ z_all,
                                                          long long stride_z, int ncols, int batch, int xcd_map) {
  // ablation builds only (tools/ablate_bt2.sh; results are wrong by construction): 1 no fragment DMA, 2 no Z traffic,
  // 4 one MFMA in ten, 8 no workgroup barriers in the diamond loop, 16 no fragment reads from LDS, 32 finished rows not
  // stored, 64 entering rows not loaded / scattered, 128 finished rows transposed but not stored, 256 entering rows
  // loaded but not scattered
  constexpr int dbg = BT2_DBG;
  extern __shared__ __attribute__((aligned(16))) double lds[];   // ring of 3 half-diamond buffers | transposition tiles
  const int n = SL.n;
  constexpr int kCols = 16 * NW;
  int mat, chunk;
  if (xcd_map) {   // all column chunks of a matrix on one XCD: they stream the same fragments through one L2
    const int nchunk = (ncols + kCols - 1) / kCols;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    mat = xcd + 8 * (slot / nchunk);
    chunk = slot % nchunk;
    if (mat >= batch) return;
  } else {
    mat = blockIdx.y;
    chunk = blockIdx.x;
  }
  // (workgroup-uniform, and told so: the fragment addresses below are then scalar arithmetic)
  const unsigned long long sb_bits = (unsigned long long)(size_t)(sb_all + (size_t)mat * SL.slab);
  const double* sb = (const double*)(size_t)(
      ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sb_bits >> 32)) << 32) |
      (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sb_bits));
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int col = chunk * kCols + 16 * w + fr;          // this lane's column in the accumulator layout
  const bool col_ok = col < ncols;

  // ---- Z <-> accumulator tiles.  In the accumulator layout a lane owns (row 4 r + fk, column fr): a global access in
  // that shape is 16 columns x 32 bytes per instruction, which the memory path serves at ~3 TB/s (measured: the Z
  // traffic alone then takes longer than all MFMAs).  So global memory is touched in the row-contiguous shape - lane l
  // moves 16 bytes: rows 2 (l & 7), + 1 of column (l >> 3) [and of column (l >> 3) + 8 in a second instruction], i.e.
  // 8 full 128-byte column segments per instruction - and each 16 x 16 tile is transposed through a wave-private LDS
  // tile [column][18] on its way to / from the accumulator layout (LDS executes a wave's accesses in order: no waits).
  constexpr int kStg = 16 * 18;
  double* stg = lds + 3 * kHalfDoubles + w * kStg;
  const int gc = lane >> 3, gr = (lane & 7) * 2;
  const int col_a = chunk * kCols + 16 * w + gc, col_b = col_a + 8;
  double* z_mat = z_all + (size_t)mat * stride_z;
  // addresses = one wave-uniform base (scalar registers; the row of an access is added there) + a 32-bit element
  // offset per lane: no 64-bit vector arithmetic per access
  const int base_col = std::max(0, std::min(chunk * kCols + 16 * w, ncols - 16));
  const gdptr zw = wave_uniform(z_mat + (size_t)base_col * n);
  // (BYTE offsets: a 32-bit offset the compiler has to scale by 8 may exceed 32 bits for all it knows, and it falls back
  // to 64-bit vector adds)
  typedef char __attribute__((address_space(1)))* gbptr;
  // the base of row `row` as an opaque scalar value (otherwise the compiler re-associates base + row + lane offset into
  // a hoisted 64-bit per-lane pointer + row, one 64-bit vector add per access again)
  // (and the lane offset re-materialised in the block of the access: hoisted out of the loop it arrives there as a 64-bit
  // value the instruction selector cannot see the zero extension of, and the scalar-base form is not chosen)
  auto lane_off = [&](unsigned e) -> unsigned {
    asm volatile("" : "+v"(e));
    return e;
  };
  auto row_base = [&](int row) -> gbptr {
    unsigned long long b = (unsigned long long)(size_t)(zw + row);
    asm volatile("" : "+s"(b));
    return (gbptr)(size_t)b;
  };
  const unsigned ea = 8u * (unsigned)(((col_a < ncols ? col_a : ncols - 1) - base_col) * n + gr);
  const unsigned eb = 8u * (unsigned)(((col_b < ncols ? col_b : ncols - 1) - base_col) * n + gr);
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  typedef const d2u __attribute__((address_space(1)))* z2ptr_c;
  typedef d2u __attribute__((address_space(1)))* z2ptr;
  struct Raw { d2u a, b; };   // one tile as it comes from / goes to memory
  // raw load, branch-free and without a use of the values (they are consumed much later; a select or a branch here makes
  // hipcc wait for every load on the spot).  A tile that sticks out of the matrix is loaded from rows n - 16 .. n - 1
  // instead and shifted back when it is scattered.
  auto load_raw = [&](Raw& t, int row0) {
    const int rs = row0 < n - 16 ? row0 : n - 16;
    const gbptr zr = row_base(rs);
    t.a = *(z2ptr_c)(zr + lane_off(ea));
    t.b = *(z2ptr_c)(zr + lane_off(eb));
  };
  // raw tile -> accumulator layout, rows beyond the matrix and columns beyond ncols masked to zero
  auto scatter_tile = [&](d4& t, const Raw& raw, int row0) {
    *(d2u*)(stg + gc * 18 + gr) = raw.a;
    *(d2u*)(stg + (gc + 8) * 18 + gr) = raw.b;
    asm volatile("" ::: "memory");   // compiler ordering only: the tile is read back through another type
    const int shift = row0 < n - 16 ? 0 : row0 - (n - 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + fk + shift;
      // (a select, not a multiplication by a 0 / 1 mask: f64 VALU work queues behind the MFMAs in flight)
      const double x = stg[fr * 18 + (i < 16 ? i : 15)];
      t[r] = (col_ok && row0 + 4 * r + fk < n) ? x : 0.0;
    }
    asm volatile("" ::: "memory");
  };
  // accumulator layout -> memory in two stages, so that the LDS round trip can sit in the shadow of MFMAs:
  // stage 1 (tile_to_rows): through the staging tile into two row-contiguous register pairs; stage 2 (store_rows).
  auto tile_to_rows = [&](const d4& t, Raw& out) {
#pragma unroll
    for (int r = 0; r < 4; ++r) stg[fr * 18 + 4 * r + fk] = t[r];
    asm volatile("" ::: "memory");
    out.a = *(const d2u*)(stg + gc * 18 + gr);
    out.b = *(const d2u*)(stg + (gc + 8) * 18 + gr);
    asm volatile("" ::: "memory");
  };
  auto store_rows = [&](const Raw& v, int row0) {
    const gbptr zr = row_base(row0);
    if (row0 + 16 <= n) {
      if (col_a < ncols) *(z2ptr)(zr + lane_off(ea)) = v.a;
      if (col_b < ncols) *(z2ptr)(zr + lane_off(eb)) = v.b;
    } else {
      if (col_a < ncols) {
        if (row0 + gr < n) *(gdptr)(zr + ea) = v.a[0];
        if (row0 + gr + 1 < n) *(gdptr)(zr + ea + 8) = v.a[1];
      }
      if (col_b < ncols) {
        if (row0 + gr < n) *(gdptr)(zr + eb) = v.b[0];
        if (row0 + gr + 1 < n) *(gdptr)(zr + eb + 8) = v.b[1];
      }
    }
  };
  auto store_tile = [&](const d4& t, int row0) {
    Raw v;
    tile_to_rows(t, v);
    store_rows(v, row0);
  };
  // the same in pieces of a few instructions (one piece per MFMA in the diamond loop): one column half per piece, the
  // tile that sticks out of the matrix handled by lane predicates instead of a second code path
  // (all 16 NW columns of the workgroup inside the matrix: with the tile inside too, the store needs no lane predicate
  // -- one scalar branch per piece instead of a dozen exec-mask branches)
  const bool full_cols = chunk * kCols + kCols <= ncols;
  auto store_half = [&](unsigned ec, bool col_in, const d2u& v, int row0) {
    const gbptr zr = row_base(row0);
    if (full_cols && row0 + 16 <= n) {
      *(z2ptr)(zr + lane_off(ec)) = v;
    } else if (col_in) {
      if (row0 + gr + 1 < n) *(z2ptr)(zr + lane_off(ec)) = v;
      else if (row0 + gr < n) *(gdptr)(zr + lane_off(ec)) = v[0];
    }
  };
  auto store_rows_a = [&](const Raw& v, int row0) { store_half(ea, col_a < ncols, v.a, row0); };
  auto store_rows_b = [&](const Raw& v, int row0) { store_half(eb, col_b < ncols, v.b, row0); };
  // scatter_tile in two pieces: raw tile -> transposition tile -> four values per lane (scatter_in); masks and the move
  // into the accumulator layout a few MFMAs later, when the LDS reads have come back (scatter_out)
  // (round 5: a tile inside the matrix, all columns of the workgroup inside too, takes ONE scalar branch and no vector
  // compare / select / index arithmetic at all -- every VALU instruction between two MFMAs costs matrix-pipe time: eight
  // v_xor per 32 MFMAs cost k_gemm3's K loop 6 %)
  auto scatter_in = [&](const Raw& raw, double (&tmp)[4], int row0) {
    *(d2u*)(stg + gc * 18 + gr) = raw.a;
    *(d2u*)(stg + (gc + 8) * 18 + gr) = raw.b;
    asm volatile("" ::: "memory");
    if (row0 < n - 16) {
#pragma unroll
      for (int r = 0; r < 4; ++r) tmp[r] = stg[fr * 18 + 4 * r + fk];
    } else {
      const int shift = row0 - (n - 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + fk + shift;
        tmp[r] = stg[fr * 18 + (i < 16 ? i : 15)];
      }
    }
    asm volatile("" ::: "memory");
  };
  auto scatter_out = [&](d4& t, const double (&tmp)[4], int row0) {
    double x[4];
    if (full_cols && row0 + 16 <= n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[r] = tmp[r];
        asm volatile("" : "+v"(x[r]));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[r] = (col_ok && row0 + 4 * r + fk < n) ? tmp[r] : 0.0;
        // (pinned here: the values are only used by the next diamond, and the compiler would otherwise sink the selects
        // to the end of this one, where nothing overlaps them)
        asm volatile("" : "+v"(x[r]));
      }
    }
    t = d4{x[0], x[1], x[2], x[3]};
  };
  // LDS-DMA: instruction q moves bytes [1024 q, 1024 q + 1024) of a half-diamond; wave w issues q = kDmaPer w + j.
  // Issued from inline asm: hipcc then keeps no scoreboard entry for them (with the builtin it guards later LDS reads
  // and register reuse with vmcnt(0), i.e. waits for the DMA it has just issued); their completion is counted by hand:
  // every wait for them below is an explicit vmcnt.
  // M0 (the LDS destination base) is written ONCE per five instructions: a write to M0 waits until the LDS-DMA
  // instructions in flight have landed (measured: with M0 saved / set / restored around every instruction the five
  // issued one memory latency apart, ~2600 cycles per half in which the wave issued nothing else).  The instruction's
  // immediate offset moves both the global and the LDS address, so five consecutive kilobytes share one M0 and one
  // scalar base, both pointing at the middle one.  Nothing else in this kernel uses M0 (saved / restored around the
  // whole loop).
  const unsigned lds_base = (unsigned)(size_t)(lvoid)lds;
  constexpr int kDmaInstr = kHalfDoubles * 8 / 1024;          // 40 per half
  constexpr int kDmaPer = kDmaInstr / NW;                     // per wave: 5 (8 waves) or 10 (4 waves)
  static_assert(kDmaPer * NW == kDmaInstr && kDmaPer % 5 == 0, "DMA instructions are shared in fives");
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const unsigned voff = (unsigned)lane * 16u;
  // 8 waves = two per SIMD: waves 4 .. 7 run one half-diamond behind waves 0 .. 3 (see the diamond loop)
#ifndef BT2_LAGSEL
#define BT2_LAGSEL 0   // experiments: 1 = odd waves trail instead of waves 4 .. 7, 2 = nobody trails
#endif
#ifndef BT2_BURST
#define BT2_BURST 0    // experiment: 1 = the heavy pieces of a first half in one burst in front of its first MFMA
#endif
#define PF(x) (BT2_BURST ? 0 : (x))
#ifndef BT2_DMA_H0
#define BT2_DMA_H0 1   // experiment: 0 = every wave issues its share of the DMA in both halves
#endif
  constexpr bool kPhased = NW == 8 && BT2_LAGSEL != 2;
  const int lag = (kPhased && (BT2_LAGSEL == 1 ? (wu & 1) != 0 : wu >= 4)) ? 1 : 0;
  // DMA instructions a wave issues in a first / second half: with the phase offset exactly one wave group is in a first
  // half during any time slot, and it issues ALL 40 (the first half is the piece-heavy one anyway; the partner's second
  // half stays as bare as possible)
  constexpr bool kDmaAllH0 = kPhased && BT2_LAGSEL == 0 && BT2_DMA_H0;
  constexpr int kDmaH0 = kDmaAllH0 ? 2 * kDmaPer : kDmaPer, kDmaH1 = kDmaAllH0 ? 0 : kDmaPer;
  const int dma_w = kDmaAllH0 ? (wu & 3) : wu;
  constexpr int kFin0 = kDmaH0 == 5 ? 10 : 16;
  // M0 and the scalar base for instructions 5 g .. 5 g + 4 of this wave, half at `src`, ring slot `slot`
  auto dma_begin = [&](const double* src, int slot, int g, int widx, int per) -> unsigned long long {
    const int qc = widx * per + 5 * g + 2;
    const unsigned long long ga = (unsigned long long)(size_t)src + (unsigned long long)qc * 1024ull;
    // (the builtin returns int: widen through unsigned, or a low half with bit 31 set would smear into the high half)
    const unsigned ga_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ga);
    const unsigned ga_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32));
    const unsigned lq = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(slot * kHalfDoubles) * 8u + (unsigned)qc * 1024u);
    // (s_nop 4: the scalar base below comes out of v_readfirstlane, and an SGPR written by a VALU instruction needs five
    // wait states before a global_* instruction reads it as its base -- hipcc pads nothing around an asm statement;
    // found in round 5 on k_gemm3, where a base reloaded from a spill lane right in front of the DMA read as zero)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4" : : "s"(lq) : "memory");
    return ((unsigned long long)ga_hi << 32) | (unsigned long long)ga_lo;
  };
  auto dma_go = [&](unsigned long long sbase, int i) {   // i = 0 .. 4 (a constant after unrolling)
    switch (i) {
      case 0: asm volatile("global_load_lds_dwordx4 %0, %1 offset:-2048" : : "v"(voff), "s"(sbase) : "memory"); break;
      case 1: asm volatile("global_load_lds_dwordx4 %0, %1 offset:-1024" : : "v"(voff), "s"(sbase) : "memory"); break;
      case 2: asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase) : "memory"); break;
      case 3: asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voff), "s"(sbase) : "memory"); break;
      default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(voff), "s"(sbase) : "memory"); break;
    }
  };
  auto dma_half = [&](const double* src, int slot) {     // all of this wave's instructions at once (group prologue)
    unsigned long long sbase = 0;
#pragma unroll
    for (int j = 0; j < kDmaPer; ++j) {
      if (j % 5 == 0) sbase = dma_begin(src, slot, j / 5, wu, kDmaPer);
      dma_go(sbase, j % 5);
    }
  };
  auto wait_vm0 = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto barrier = [&]() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // Window registers: three arrays of four tiles (64 rows each).  The window of diamond k is arrays ph | ph + 1 (mod 3),
  // ph = k mod 3; the third array first holds the 64 rows finished at the last slide (stored behind the MFMAs of this
  // diamond's first half) and is then refilled with the 64 rows that enter the window at the next slide (scattered in
  // behind the MFMAs of the second half): sliding is a renaming, no register moves and nothing left outside the shadow
  // of the MFMA runs.  The diamond body is instantiated once per phase so that every tile index is a constant.
  d4 zz[12];
  int fin_row = 0;
  bool have_fin = false;

  BT2_STAMP_DECL
  BT2_CLOCK_BEGIN
  int S_cur = 0;   // (read by the -DBT2_TRACE build only)
  (void)S_cur;
  auto diamond = [&](auto PH, const double* fgrp, int k, int nh, bool more, int& slot, int win) {
    constexpr int ph = decltype(PH)::value;
#ifdef BT2_TRACE
    const bool trace_on = blockIdx.x == 0 && S_cur == SL.ngroups / 2 && k == 3;
#endif
#define ZT(rt) zz[4 * ((ph + (rt) / 4) % 3) + (rt) % 4]
    Raw zn[2], fin_rows;
#pragma unroll
    for (int H = 0; H < 2; ++H) {
      const int q = 2 * k + H;                               // half index inside the group
      typedef double d2l __attribute__((ext_vector_type(2)));
      const d2l* ldsP = (const d2l*)(lds + slot * kHalfDoubles) + lane;   // pair p of this half: ldsP[64 p]
      // the half fetched during this time slot is the one the leading wave group runs NEXT (slot index = q + lag):
      // ring position (q + lag + 1) mod 3, the one the trailing group left at the last barrier
      int slot_pre = slot + 1 + lag;
      slot_pre = slot_pre >= 3 ? slot_pre - 3 : slot_pre;
      // (past the end of the group the last half is fetched again, into a slot nobody reads: no branch in the loop)
      const double* src_pre = fgrp + (size_t)(q + lag + 1 < nh ? q + lag + 1 : nh - 1) * kHalfDoubles;
      BT2_STAMP(7 * H)
      if (!(dbg & 8)) barrier();   // half q complete in LDS (every wave waited for its own part); half q - 1 finished
      BT2_STAMP(7 * H + 1)
      // ---- 80 MFMAs: minis st = 3 - 2 H and 2 - 2 H, everything else in their shadow: the DMA of half q + 2; in the
      // first half the store of the rows finished at the last slide (through the transposition tile, then to memory);
      // the loads of the 64 rows that enter the window at the slide and their way through the transposition tile into
      // the spare array (issued and consumed in EVERY iteration - after the last diamond of a group they are not needed,
      // the clamped addresses are still valid: hipcc's wait-count bookkeeping is not path sensitive, and loads that
      // are only issued / consumed under `more` stay "maybe pending" around the loop, which costs a vmcnt(0) wherever
      // their registers are reused).
      // One MFMA at a time: what else the half has to do is cut into pieces of a few instructions each, and at most one
      // piece sits in front of every MFMA, so that it issues while the previous MFMA occupies the matrix pipe (64
      // cycles).  (Measured with the in-kernel stamps: with the same work bunched between groups of 4 or 8 MFMAs a wave
      // spends as long issuing its ~600 other instructions per half as the pipe needs for its 80 MFMAs, and the second
      // wave of the SIMD cannot fill the holes because it runs the same pattern.)
      d4 wa = d4{0, 0, 0, 0};
#ifndef BT2_KAHEAD
#define BT2_KAHEAD 8
#endif
      constexpr int kAhead = BT2_KAHEAD;                      // fragments in flight between LDS and the MFMA that uses them
      double fq[kAhead];
      double sc[4];                                           // a new tile between the transposition tile and its select
      unsigned long long dma_base = 0;                        // scalar base of the DMA instructions being issued
#pragma unroll
      for (int j = 0; j < kAhead / 2; ++j) {
        const d2l t = ldsP[j * 64];
        fq[2 * j] = t[0];
        fq[2 * j + 1] = t[1];
      }
#pragma unroll
      for (int f = 0; f < ((dbg & 4) ? 8 : kHalfFrags); ++f) {
        // ---- this step's piece
        // Two waves share a SIMD (w and w + 4) and the trailing one runs one half behind (lag): the pieces are placed
        // so that the partner of a wave in a piece-heavy stretch is in a stretch of (nearly) bare MFMAs and can keep
        // the matrix pipe fed -- the heavy pieces (DMA issue, the stores of the finished rows) sit in the first 40
        // steps of the FIRST half, which run beside the partner's second half, whose first 40 steps carry three light
        // pieces; the second 40 steps of both halves carry a few light ones each.
        if (H == 0) {
#if BT2_BURST
          if (f == 0 && !(dbg & 1)) {
#pragma unroll
            for (int j = 0; j < kDmaH0; ++j) {
              if (j % 5 == 0) dma_base = dma_begin(src_pre, slot_pre, j / 5, dma_w, kDmaH0);
              dma_go(dma_base, j % 5);
            }
          }
#else
          if (f < kDmaH0 && !(dbg & 1)) {
            if (f % 5 == 0) dma_base = dma_begin(src_pre, slot_pre, f / 5, dma_w, kDmaH0);
            dma_go(dma_base, f % 5);
          }
#endif
        } else {
          if (f >= 46 && f < 46 + kDmaH1 && !(dbg & 1)) {
            if ((f - 46) % 5 == 0) dma_base = dma_begin(src_pre, slot_pre, (f - 46) / 5, dma_w, kDmaH1);
            dma_go(dma_base, (f - 46) % 5);
          }
        }
        if (!(dbg & 2)) {
          if (H == 0) {
            // rows entering at the slide, tiles 0 and 1: requested right after the half's DMA instructions (the compiler
            // does not see those: its vmcnt for the first use of a row then covers them, being older, and nothing younger)
            if (f == PF(kDmaH0 + 1) && !(dbg & 64)) load_raw(zn[0], win + 128);
            if (f == PF(kDmaH0 + 3) && !(dbg & 64)) load_raw(zn[1], win + 128 + 16);
            // rows finished at the last slide: tile i through the transposition tile at step 10 + 8 i (8 waves; 14 + 8 i
            // with the 10 DMA steps of the 4-wave variant), column halves a / b stored 4 and 6 steps later
            if (have_fin && !(dbg & 32)) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (f == PF(kFin0 + 8 * i)) tile_to_rows(ZT(8 + i), fin_rows);
                if (dbg & 128) {   // (ablation: the transposition without the stores)
                  if (f == PF(kFin0 + 4 + 8 * i)) asm volatile("" : : "v"(fin_rows.a), "v"(fin_rows.b));
                } else {
                  if (f == PF(kFin0 + 4 + 8 * i)) store_rows_a(fin_rows, fin_row + 16 * i);
                  if (f == PF(kFin0 + 6 + 8 * i)) store_rows_b(fin_rows, fin_row + 16 * i);
                }
              }
            }
            // tiles 0, 1 of the new rows into the spare array (free since step kFin0 + 24), then the requests for
            // tiles 2, 3 into the same two raw registers: the youngest four loads of the half, the only ones its closing
            // vmcnt(4) leaves in flight
            if (f == 52 && (dbg & 256)) asm volatile("" : : "v"(zn[0].a), "v"(zn[0].b));
            if (f == 52 && !(dbg & 64) && !(dbg & 256)) scatter_in(zn[0], sc, win + 128);
            if (f == 58 && !(dbg & 64) && !(dbg & 256)) scatter_out(ZT(8), sc, win + 128);
            if (f == 60 && (dbg & 256)) asm volatile("" : : "v"(zn[1].a), "v"(zn[1].b));
            if (f == 60 && !(dbg & 64) && !(dbg & 256)) scatter_in(zn[1], sc, win + 128 + 16);
            if (f == 66 && !(dbg & 64) && !(dbg & 256)) scatter_out(ZT(9), sc, win + 128 + 16);
            if (f == 68 && !(dbg & 64)) load_raw(zn[0], win + 128 + 32);
            if (f == 70 && !(dbg & 64)) load_raw(zn[1], win + 128 + 48);
          } else {
            // tiles 2, 3 (requested ~40 steps ago) before this half's DMA instructions are issued: the compiler's
            // vmcnt(0) for them must not cover the DMA
            if (f == 28 && (dbg & 256)) asm volatile("" : : "v"(zn[0].a), "v"(zn[0].b));
            if (f == 28 && !(dbg & 64) && !(dbg & 256)) scatter_in(zn[0], sc, win + 128 + 32);
            if (f == 34 && !(dbg & 64) && !(dbg & 256)) scatter_out(ZT(10), sc, win + 128 + 32);
            if (f == 36 && (dbg & 256)) asm volatile("" : : "v"(zn[1].a), "v"(zn[1].b));
            if (f == 36 && !(dbg & 64) && !(dbg & 256)) scatter_in(zn[1], sc, win + 128 + 48);
            if (f == 42 && !(dbg & 64) && !(dbg & 256)) scatter_out(ZT(11), sc, win + 128 + 48);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef BT2_STAMPS
#ifdef BT2_STAMPS_FINE   // the first half's four inner stamps after steps 2, 5, 8, 11 (or 13, 16, 19, 22 with =2) instead
        if (H == 0 && f >= 3 + 11 * (BT2_STAMPS_FINE - 1) && f <= 12 + 11 * (BT2_STAMPS_FINE - 1) &&
            (f - 11 * (BT2_STAMPS_FINE - 1)) % 3 == 0) BT2_STAMP(1 + (f - 11 * (BT2_STAMPS_FINE - 1)) / 3)
        if (H == 1 && (f == 16 || f == 32 || f == 48 || f == 64)) BT2_STAMP(7 * H + 1 + f / 16)
#else
        if (f == 16 || f == 32 || f == 48 || f == 64) BT2_STAMP(7 * H + 1 + f / 16)
#endif
#endif
        BT2_TRACE_POINT(H, f)
        // ---- the MFMA
        {
          const int st = 3 - 2 * H - f / kMiniFrags, p = f % kMiniFrags;
          const double a = fq[f % kAhead];
          if (p < 20) {            // W += V^T Z: one accumulator (a dependent chain of this MFMA issues at the full
            const int rt = st + p / 4, r = p % 4;   // rate of one per 64 cycles: tools/probe_mini_chain.hip)
            if (p == 0) wa = d4{0, 0, 0, 0};
            wa = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ZT(rt)[r], wa, 0, 0, 0);
          } else {                 // Z -= (V T) W: the five row tiles take turns
            const int jj = p - 20, r = jj / 5, rt = st + jj % 5;
            ZT(rt) = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wa[r], ZT(rt), 0, 0, 0);
          }
        }
        // ---- the fragments kAhead steps ahead take the registers the last two MFMAs have read (one 16-byte LDS read)
        if ((f & 1) && f + kAhead - 1 < kHalfFrags && !(dbg & 16)) {
          const d2l t = ldsP[((f + kAhead - 1) / 2) * 64];
          fq[(f - 1) % kAhead] = t[0];
          fq[f % kAhead] = t[1];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      BT2_TRACE_POINT(H, 80)
      BT2_STAMP(7 * H + 6)
      // this wave's part of the half fetched during this slot must have landed before the barrier that opens the next
      // slot (and the stores of the finished rows with it); after a first half the four youngest loads (new rows, tiles
      // 2 and 3) stay in flight
      if (H == 0 && !(dbg & 2) && !(dbg & 64)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else wait_vm0();
      slot = slot == 2 ? 0 : slot + 1;
    }
    // ---- slide by 64 rows = the next phase.  After the last diamond of a group the whole window goes back to memory.
    BT2_STAMP(14)
    if (more) {
      fin_row = win;
      have_fin = true;
    } else {
      if (!(dbg & 2)) {
#pragma unroll
        for (int t = 0; t < 8; ++t) store_tile(ZT(t), win + 16 * t);
      }
      have_fin = false;
    }
    BT2_STAMP(15)
    // nothing of this diamond may sink into the next one: behind its LDS-DMA issue the compiler would wait vmcnt(0) for it
    __builtin_amdgcn_sched_barrier(0);
#undef ZT
  };
  unsigned m0_keep;
  asm volatile("s_mov_b32 %0, m0" : "=s"(m0_keep));
  for (int S = SL.ngroups - 1; S >= 0; --S) {
    S_cur = S;
    const int d0 = dia_off[S], nk = dia_off[S + 1] - d0;
    const int nh = 2 * nk;                                    // halves of this group, streamed back to back
    int win = S * kG + 1;
    const double* fgrp = sb + SL.frag + (size_t)d0 * kFragDoubles;
    {
      Raw raw[8];
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) load_raw(raw[rt], win + 16 * rt);
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) scatter_tile(zz[rt], raw[rt], win + 16 * rt);
    }
    barrier();                       // every wave has left the previous group: the whole ring is free
    if (!(dbg & 1)) {
      dma_half(fgrp, 0);
    }
    wait_vm0();
    __builtin_amdgcn_sched_barrier(0);
    if (lag) {
      // time slot 0 of the trailing wave group: the leading group runs half 0; only this group's share of half 1's DMA
      if (!(dbg & 8)) barrier();
      if (!(dbg & 1) && !kDmaAllH0) {
        dma_half(fgrp + (size_t)(1 < nh ? 1 : 0) * kHalfDoubles, 1);
      }
      wait_vm0();
      __builtin_amdgcn_sched_barrier(0);
    }
    int slot = 0;                    // ring slot of the half about to run
    int ph = 0;
    for (int k = 0; k < nk; ++k, win += 64) {
      const bool more = k + 1 < nk;
      if (ph == 0) diamond(std::integral_constant<int, 0>{}, fgrp, k, nh, more, slot, win);
      else if (ph == 1) diamond(std::integral_constant<int, 1>{}, fgrp, k, nh, more, slot, win);
      else diamond(std::integral_constant<int, 2>{}, fgrp, k, nh, more, slot, win);
      ph = ph == 2 ? 0 : ph + 1;
    }
    // time slot nh of the leading group (the trailing one runs its last half): one barrier, so that both count the same
    if (kPhased && !lag && !(dbg & 8)) barrier();
  }
  asm volatile("s_mov_b32 m0, %0" : : "s"(m0_keep));
  BT2_STAMP_WRITE
  BT2_CLOCK_END
}

// ================================================================================================================
// k_bt2_role (round 5): the same product Z <- Q2 Z with the ROLES of a workgroup's waves split, as in k_gemm3
// (gemm3.hip).  k_bt2_apply's eight waves each do everything -- MFMAs, the LDS-DMA of the fragments, the loads /
// transposition / masks / stores of the window rows -- and every one of their memory and vector instructions costs the
// SIMD's matrix pipe time (the pipes are busy 0.79 of that launch).  Here
//   * waves 0-3 are one MFMA wave per SIMD, 16 columns each (a workgroup = 64 columns of one matrix): the 128-row window
//     in eight accumulator tiles that alternate between two code phases (see below), 160 MFMAs per diamond from
//     ready-made fragments in LDS -- and nothing else but LDS reads / writes: the 64 rows that enter the window at a slide
//     are read from an LDS image in accumulator layout, the 64 finished rows are written to one;
//   * waves 4-7 stream the fragments a quarter-diamond (one mini: 20 KB, five 1-KB pieces per wave behind ONE write of
//     M0) at a time into a ring of three quarter buffers, two quarters ahead;
//   * waves 8-11 (one per MFMA wave) move the window rows: LDS-DMA of the entering rows, column by column (512 bytes per
//     instruction: lanes 0-31), into one of two images, a whole diamond before the MFMA wave reads them, and
//     16-byte-per-lane stores of the finished rows out of a third image; they also zero what lies below the matrix, so
//     the MFMA waves need no masks.
// One barrier per quarter-diamond; per sweep group two more in front (the window's two halves come in through the two
// images) and four behind (they go out).  LDS: 3 x 20 KB + 3 x 4 x 8 448 B = 162 816 B.
// (Version 1 -- ring of two half-diamond buffers, one image in, everything fetched one half ahead -- was correct and took
// 999 ms per C3 step against k_bt2_apply's 605: a loader's two M0 groups per half land one after the other, and the
// entering rows come from HBM: both sat on the critical path of every diamond.  Version 3, one barrier at the head of
// every quarter: 645 ms; its stamps (profiles/r05_bt2_role_stamps.txt) showed the MFMA waves waiting 1 250 + 710 cycles
// per diamond for the Z waves -- whose vector-ALU instructions only issue in the gaps the MFMA wave leaves -- and 1 480
// cycles in the slide, when all four MFMA waves push 64 KB of window rows through the LDS at once.  This version: 614 ms
// against 594, the same 12.2 k cycles per diamond and SIMD as k_bt2_apply -- 0.84 of the MFMA issue slots -- at the
// 2.17 - 2.20 GHz both kernels run at, profiles/r05_bt2_clock.txt.  Not the default: SPRINGCRAFT_BT2_ROLE = 1.)
// Image layout: column c of a wave's 16 at c * 528 B (64 rows + 16 B: an accumulator register's 16 columns x 2 rows fall
// into 32 distinct 8-byte bank pairs).  Needs n even (the window starts at an odd row and moves in pairs of rows) --
// bt2_batched checks.
constexpr int kRoleCol = 528;                       // bytes between the columns of a window image
constexpr int kRoleImg = 16 * kRoleCol;             // one wave's image: 16 columns x 64 rows
constexpr int kQuarterBytes = kMiniFrags * 64 * 8;  // 20 480
constexpr int kRoleRing = 3 * kQuarterBytes;
constexpr int kRoleLds = kRoleRing + 3 * 4 * kRoleImg;

__device__ __forceinline__ unsigned long long role_uni64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// (diagnostic build -DBT2_STAMPS -DBT2_ROLE_STAMPS: cycles of an MFMA wave per diamond -- slots 4 q: the MFMAs of quarter
// q up to the barrier inside it (+ the tail of the quarter before), 4 q + 1: the wait in that barrier, 14: the tail of
// quarter 3, 15: the slide; tools/bt2_role_stamps.py)
#if defined(BT2_STAMPS) && defined(BT2_ROLE_STAMPS)
#define ROLE_STAMP(i) BT2_STAMP(i)
#define ROLE_STAMP_DECL BT2_STAMP_DECL
#define ROLE_STAMP_WRITE BT2_STAMP_WRITE
#else
#define ROLE_STAMP(i)
#define ROLE_STAMP_DECL
#define ROLE_STAMP_WRITE
#endif
__global__ __launch_bounds__(768, 1) void k_bt2_role(const double* __restrict__ sb_all, SbLayout SL,
                                                     const int* __restrict__ dia_off, double* __restrict__ z_all,
                                                     long long stride_z, int ncols, int batch) {
  extern __shared__ __attribute__((aligned(16))) char rl[];   // ring of 3 quarter buffers | 2 x 4 images in | 4 images out
  const int n = SL.n;
  const int nchunk = (ncols + 63) / 64;
  const int xcd = blockIdx.x & 7, slot_wg = blockIdx.x >> 3;
  const int mat = xcd + 8 * (slot_wg / nchunk), chunk = slot_wg % nchunk;   // all column chunks of a matrix on one XCD
  if (mat >= batch) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)rl;
  const int ngroups = SL.ngroups;
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // diamonds of sweep group S (read from global memory: made uniform for the compiler)
  auto group_d0 = [&](int S) { return __builtin_amdgcn_readfirstlane(dia_off[S]); };

  if (w >= 4 && w < 8) {
    // ------------------------------------------------------------------------------------------ fragment loaders
    // wave lw fetches pieces 5 lw .. 5 lw + 4 of a quarter (20 pieces of 1 KB) behind ONE write of M0.  A write to M0
    // waits for the wave's LDS-DMA in flight: issuing quarter q + 2 therefore also means that quarter q + 1 has landed.
    const int lw = w - 4;
    const unsigned voff = (unsigned)lane * 16u;
    const double* sb = sb_all + (size_t)mat * SL.slab;
    auto fetch_quarter = [&](const double* src, int buf) {
      // (scalar by construction: kernel arguments, blockIdx, readfirstlane'd counters -- no VALU instruction in this wave's
      // loop: a helper wave's VALU instructions wait for issue slots of a SIMD that the MFMA wave keeps busy)
      const unsigned long long g = (unsigned long long)(size_t)src + (unsigned long long)(5 * lw + 2) * 1024ull;
      const unsigned m0v = lds_base + (unsigned)(buf * kQuarterBytes + (5 * lw + 2) * 1024);
      asm volatile(
          "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:-2048\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:-1024\n\t"
          "global_load_lds_dwordx4 %1, %2\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:2048"
          :
          : "s"(m0v), "v"(voff), "s"(g)
          : "memory");
    };
    for (int S = ngroups - 1; S >= 0; --S) {
      const int d0 = group_d0(S), nk = group_d0(S + 1) - d0;
      const int nq = 4 * nk;
      const char* fgrp = (const char*)(sb + SL.frag + (size_t)d0 * kFragDoubles);
      fetch_quarter((const double*)fgrp, 0);
      fetch_quarter((const double*)(fgrp + kQuarterBytes), 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      barrier();   // b0
      barrier();   // b1
      int buf = 2;
      for (int q = 0; q < nq; ++q) {
        barrier();   // quarter q starts; the buffer of quarter q - 1 is free
        if (q + 2 < nq) {
          fetch_quarter((const double*)(fgrp + (size_t)(q + 2) * kQuarterBytes), buf);
          // quarter q + 1 (the five instructions before these five) must have LANDED at the next barrier: an explicit
          // count -- the wait that a write to M0 implies is not one to build on (version 2 without it: wrong results in
          // some members of a batch)
          asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        buf = buf == 2 ? 0 : buf + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      barrier(); barrier(); barrier(); barrier();   // e0 .. e3
    }
    return;
  }

  if (w >= 8) {
    // ------------------------------------------------------------------------------------------------- Z waves
    // NO vector-ALU instruction in this wave's steady state: a helper wave's VALU instruction only issues when the MFMA
    // wave of its SIMD leaves a gap, i.e. at that wave's next barrier -- the stamps of the first version showed the MFMA
    // waves waiting 1 250 cycles per diamond for this wave, which was still computing lane addresses and predicates.  So:
    // lane offsets are computed once per kernel, everything that changes goes into the scalar base, lanes are switched off
    // by scalar writes of EXEC inside the asm statements, and only windows that reach below the matrix (the last one or two
    // of a sweep group) take the general path.
    const int zw = w - 8;
    const int c0 = chunk * 64 + 16 * zw;                       // first of this wave's 16 columns
    double* z_mat = z_all + (size_t)mat * stride_z;
    const unsigned img_in0 = lds_base + (unsigned)(kRoleRing + zw * kRoleImg);
    char* p_in0 = rl + kRoleRing + zw * kRoleImg;
    char* p_out = rl + kRoleRing + 8 * kRoleImg + zw * kRoleImg;
    const unsigned long long col8 = (unsigned long long)n * 8ull;
    const int pos = lane & 31, jj = lane >> 5;
    unsigned lane16 = (unsigned)pos * 16u;                                        // a pair of rows per lane
    unsigned lane_st = (unsigned)((unsigned long long)jj * col8) + (unsigned)pos * 16u;   // stores: two columns per instruction
    unsigned img_st = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p_out + (unsigned)(jj * kRoleCol + pos * 16);
    asm volatile("" : "+v"(lane16), "+v"(lane_st), "+v"(img_st));
    // rows row0 .. row0 + 63 of the 16 columns -> image `which` (lanes 0-31: rows row0 + 2 l, + 1 of one column per
    // instruction; M0 = the middle of the image)
    // (one instruction per column; the scalar base moves on by a column less the 528 bytes that the immediate grows by.
    // The base is an in-out operand of the asm statement: hipcc then cannot compute the sixteen bases ahead -- it did, kept
    // them across the diamond loop, ran out of scalar registers and reloaded them with v_readlane, a VALU instruction)
#define ROLE_DMA(J, IMM)                                                                                      \
  asm volatile("s_mov_b32 exec_hi, 0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %0 offset:" #IMM                \
               "\n\ts_mov_b32 exec_hi, -1" : "+s"(b_) : "v"(vo) : "memory");                                   \
  b_ += (c0 + (J) + 1 < nc) ? step : (unsigned long long)(-528ll);
#define ROLE_DMA16                                                                                     \
  ROLE_DMA(0, -3960) ROLE_DMA(1, -3432) ROLE_DMA(2, -2904) ROLE_DMA(3, -2376)                        \
  ROLE_DMA(4, -1848) ROLE_DMA(5, -1320) ROLE_DMA(6, -792) ROLE_DMA(7, -264)                          \
  ROLE_DMA(8, 264) ROLE_DMA(9, 792) ROLE_DMA(10, 1320) ROLE_DMA(11, 1848)                            \
  ROLE_DMA(12, 2376) ROLE_DMA(13, 2904) ROLE_DMA(14, 3432) ROLE_DMA(15, 3960)
    auto fetch_rows = [&](int row0, int which) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(img_in0 + (unsigned)(which * 4 * kRoleImg) + 3960u) : "memory");
      int nc = ncols;
      asm volatile("" : "+s"(nc));
      const unsigned long long step = col8 - 528ull;
      if (row0 + 64 <= n) {
        // all 64 rows exist: the row goes into the scalar base; lanes 32-63 are off inside each asm statement (EXEC is
        // whole again at its end: the compiler knows nothing of the change and may place its own instructions in between)
        unsigned long long b_ = (unsigned long long)(size_t)z_mat + (unsigned long long)row0 * 8ull +
                                (unsigned long long)min(c0, nc - 1) * col8 + 3960ull;
        const unsigned vo = lane16;
        ROLE_DMA16
      } else {
        // (rows beyond the matrix re-read its last pair and are put right by fix_rows)
        unsigned long long b_ = (unsigned long long)(size_t)z_mat + (unsigned long long)min(c0, nc - 1) * col8 + 3960ull;
        const unsigned vo = (unsigned)min(row0 + 2 * pos, n - 2) * 8u;
        ROLE_DMA16
      }
    };
#undef ROLE_DMA16
#undef ROLE_DMA
    // (after the rows have landed) what lies below the matrix reads as zero; a pair that straddles the last row was read
    // one row early: its second value is the last row
    auto fix_rows = [&](int row0, int which) {
      if (row0 + 64 <= n) return;
      typedef double d2z __attribute__((ext_vector_type(2)));
      if (lane < 32) {
        const int r = row0 + 2 * lane;
        for (int j = 0; j < 16; ++j) {
          d2z* u = (d2z*)(p_in0 + which * 4 * kRoleImg + j * kRoleCol + lane * 16);
          if (r >= n) *u = d2z{0.0, 0.0};
          else if (r == n - 1) { const d2z v = *u; *u = d2z{v[1], 0.0}; }
        }
      }
    };
    // image out -> rows row0 .. row0 + 63 of the 16 columns, two columns per instruction: all eight reads first, then
    // the stores with the scalar base of the column pair (+ the row).  Windows inside the matrix and waves whose 16 columns
    // all exist -- all but the last one or two windows of a sweep group and the last column chunk -- store without any
    // predicate; the others take plain predicated stores.  (Switching lanes off by writing EXEC inside the asm statement,
    // as the row fetch does, gave wrong results for the stores: tools/r05_role_dbg.sh, dbg 0 against dbg 1.)
    auto store_rows = [&](int row0) {
      typedef double d2z __attribute__((ext_vector_type(2)));
      const __attribute__((address_space(3))) char* img = (const __attribute__((address_space(3))) char*)(size_t)img_st;
      d2z vv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) vv[i] = *(const __attribute__((address_space(3))) d2z*)(img + 2 * i * kRoleCol);
      __builtin_amdgcn_sched_barrier(0);
      int nc = ncols;
      asm volatile("" : "+s"(nc));
      if (row0 + 64 <= n && c0 + 16 <= nc) {
        unsigned long long b = (unsigned long long)(size_t)z_mat + (unsigned long long)c0 * col8 + (unsigned long long)row0 * 8ull;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("s_nop 4\n\tglobal_store_dwordx4 %1, %2, %0\n\ts_nop 1" : "+s"(b) : "v"(lane_st), "v"(vv[i]) : "memory");
          b += 2ull * col8;
        }
      } else {
        const int r = row0 + 2 * pos;
        for (int i = 0; i < 8; ++i) {
          double* dst = z_mat + (size_t)(c0 + 2 * i + jj) * n + r;
          if (c0 + 2 * i + jj < nc) {
            if (r + 1 < n) { dst[0] = vv[i][0]; dst[1] = vv[i][1]; }
            else if (r + 1 == n) dst[0] = vv[i][0];
          }
        }
      }
    };
    for (int S = ngroups - 1; S >= 0; --S) {
      const int d0 = group_d0(S), nk = group_d0(S + 1) - d0;
      const int win0 = S * kG + 1;
      fetch_rows(win0, 0);
      fetch_rows(win0 + 64, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      fix_rows(win0, 0);
      fix_rows(win0 + 64, 1);
      barrier();   // b0: both halves of the window are in the images
      barrier();   // b1: the MFMA wave has read them
      // E_k = the rows that enter after diamond k (rows win_k + 128 ..), read by the MFMA wave during the last quarter of
      // diamond k out of image k & 1; fetched more than a diamond earlier
      if (1 < nk) fetch_rows(win0 + 128, 0);
      for (int k = 0; k < nk; ++k) {
        const int win = win0 + 64 * k;
        barrier();   // quarter 0 (the rows finished at the last slide went into the image out at the end of the last diamond)
        barrier();   // quarter 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // E_k (fetched a diamond ago) has landed; the last diamond's stores are out
        if (k + 1 < nk) fix_rows(win + 128, k & 1);
        if (k > 0) store_rows(win - 64);
        barrier();   // quarter 2
        // E_{k+1} into the image that the MFMA wave read a diamond ago.  Here, not behind the barrier of quarter 3: the
        // slide that follows that one is the busiest moment of the LDS
        if (k + 2 < nk) fetch_rows(win + 192, (k + 1) & 1);
        barrier();   // quarter 3: the MFMA wave reads E_k
      }
      const int winl = win0 + 64 * (nk - 1);
      barrier();   // e0
      barrier();   // e1: rows winl .. + 63 are in the image out
      store_rows(winl);
      barrier();   // e2 (they have been read)
      barrier();   // e3: rows winl + 64 .. + 127
      store_rows(winl + 64);
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------- MFMA waves
  const int fr = lane & 15, fk = lane >> 4;
  // this lane's place in a window image: column fr, row fk (+ 16 per tile, + 4 per register)
  char* q_in0 = rl + kRoleRing + w * kRoleImg + fr * kRoleCol + fk * 8;
  char* q_out = q_in0 + 8 * kRoleImg;
  // The window: eight accumulator tiles (128 rows x 16 columns) in ONE array.  Diamonds alternate between two code PHASES:
  // logical tile t of the window is zz[(t + 4 * phase) & 7], so the slide -- the upper half leaves, the lower half becomes
  // the upper, 64 new rows enter below -- moves no register: the four tiles that leave are written to the image out and the
  // entering rows are read into the same registers, which are the lower half of the next phase.  (k_bt2_apply rotates
  // three arrays of four tiles through three phases; with the 168 registers of a 12-wave workgroup hipcc moved accumulator
  // tuples through scratch where those phases join: 1 130 ms.  Version 3 kept one phase and moved 16 registers per
  // diamond: 645 ms.)
  d4 zz[8];
  ROLE_STAMP_DECL
  BT2_CLOCK_BEGIN
  // fragment registers: two rings of eight that alternate from quarter to quarter.  The barrier that opens quarter q + 1
  // sits INSIDE quarter q, behind its 36th MFMA: by then every fragment of quarter q is in registers (its buffer may be
  // overwritten), the first eight fragments of quarter q + 1 are requested right behind the barrier, and the last four
  // MFMAs of quarter q cover their way from LDS -- no MFMA waits at a quarter's start.
  double fa[8], fb[8];
  typedef double d2l __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) char* lcp;
  auto frag_ptr = [&](const char* ring) {
    // (opaque: the constants below go into the LDS instructions' offset fields instead of registers of their own)
    unsigned ring_a = (unsigned)(size_t)(lcp)ring + (unsigned)lane * 16u;
    asm volatile("" : "+v"(ring_a));
    return (const __attribute__((address_space(3))) d2l*)(size_t)ring_a;
  };
  auto first_frags = [&](const char* ring, double (&f)[8]) {
    const __attribute__((address_space(3))) d2l* lp = frag_ptr(ring);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const d2l t = lp[j * 64];
      f[2 * j] = t[0];
      f[2 * j + 1] = t[1];
    }
  };
  // One quarter-diamond = one mini (st = 3 - QI): 20 MFMAs of W = V^T Z, 20 of Z -= (V T) W, fragments at `ring`; `more`:
  // another quarter follows in this sweep group (its fragments at `next_ring`).
  // the 64 rows that enter at the next slide: read tile by tile during quarter 3 (four reads behind its MFMAs 1, 9, 17, 25)
  d4 ze[4];
  auto quarter = [&](auto QI, auto PH, const char* ring, const char* next_ring, double (&cur)[8], double (&nxt)[8],
                     const char* enter) {
    // (no "another quarter follows" flag: the last quarter of a sweep group does the same -- its barrier is the first of the
    // group's epilogue, e0, and the fragments and rows it reads ahead are simply not used.  With a branch around the barrier
    // and the reads, hipcc joined the two paths' pending-read counts conservatively and made MFMA 36 wait for the fragments
    // of the NEXT quarter; with a second copy of the diamond for "no successor" it moved accumulator tuples through
    // scratch: 301 spills)
    constexpr bool more = true;
    constexpr int st = 3 - decltype(QI)::value;
    constexpr int ph4 = 4 * decltype(PH)::value;
    const __attribute__((address_space(3))) d2l* ldsP = frag_ptr(ring);
    d4 wa = d4{0, 0, 0, 0};
    constexpr int kAhead = 8;
#pragma unroll
    for (int p = 0; p < kMiniFrags; ++p) {
      const double a = cur[p % kAhead];
      if (p < 20) {
        const int rt = (st + p / 4 + ph4) & 7, r = p % 4;
        wa = __builtin_amdgcn_mfma_f64_16x16x4f64(a, zz[rt][r], p == 0 ? d4{0, 0, 0, 0} : wa, 0, 0, 0);
      } else {
        const int jj = p - 20, r = jj / 5, rt = (st + jj % 5 + ph4) & 7;
        zz[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wa[r], zz[rt], 0, 0, 0);
      }
      if ((p & 1) && p + kAhead - 1 < kMiniFrags) {
        const d2l t = ldsP[((p + kAhead - 1) / 2) * 64];
        cur[(p - 1) % kAhead] = t[0];
        cur[p % kAhead] = t[1];
      }
      if (decltype(QI)::value == 3 && more && (p & 7) == 1 && p < 32) {
        unsigned img_a = (unsigned)(size_t)(lcp)enter;
        asm volatile("" : "+v"(img_a));
        lcp img = (lcp)(size_t)img_a;
#pragma unroll
        for (int r = 0; r < 4; ++r) ze[p >> 3][r] = *(const __attribute__((address_space(3))) double*)(img + (16 * (p >> 3) + 4 * r) * 8);
      }
      if (p == 35 && more) {
        ROLE_STAMP(4 * decltype(QI)::value)
        barrier();                    // quarter q + 1 starts (the last fragment reads of this one were issued at p = 31)
        ROLE_STAMP(4 * decltype(QI)::value + 1)
        first_frags(next_ring, nxt);
      }
      __builtin_amdgcn_sched_barrier(0);   // (the order as written: fragment reads a few MFMAs ahead, not all at once)
    }
  };
  // four tiles <-> an image (64 rows x 16 columns), a tile at a time
  auto tiles_in = [&](int base, const char* img_) {
    unsigned img_a = (unsigned)(size_t)(lcp)img_;
    asm volatile("" : "+v"(img_a));
    lcp img = (lcp)(size_t)img_a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) zz[base + i][r] = *(const __attribute__((address_space(3))) double*)(img + (16 * i + 4 * r) * 8);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto tiles_out = [&](int base) {
    unsigned out_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)q_out;
    asm volatile("" : "+v"(out_a));
    __attribute__((address_space(3))) char* qo = (__attribute__((address_space(3))) char*)(size_t)out_a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *(__attribute__((address_space(3))) double*)(qo + (16 * i + 4 * r) * 8) = zz[base + i][r];
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the slide behind a diamond of phase PH: tile by tile, the finished rows to the image out, the entering rows E_k (read
  // into `ze` during quarter 3: all four MFMA waves slide at the same moment, and 64 KB of window rows + the fragments
  // of the next quarter + the rows the Z waves fetch through the LDS in one burst took 1 480 cycles) into the same registers
  auto slide = [&](int base) {
    unsigned out_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)q_out;
    asm volatile("" : "+v"(out_a));
    __attribute__((address_space(3))) char* qo = (__attribute__((address_space(3))) char*)(size_t)out_a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *(__attribute__((address_space(3))) double*)(qo + (16 * i + 4 * r) * 8) = zz[base + i][r];
      zz[base + i] = ze[i];
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  for (int S = ngroups - 1; S >= 0; --S) {
    const int nk = group_d0(S + 1) - group_d0(S);
    barrier();                 // b0: the window's two halves are in the two images
    tiles_in(0, q_in0);
    tiles_in(4, q_in0 + 4 * kRoleImg);
    barrier();                 // b1
    barrier();                 // the group's first quarter starts (all later quarters: inside their predecessor)
    first_frags(rl, fa);
    int buf = 0;               // 4 k % 3: the ring buffer of the diamond's first quarter
    auto diamond = [&](auto PH, int k) {
      constexpr int base = 4 * decltype(PH)::value;     // registers of the logical tiles 0 .. 3
      const char* r0 = rl + buf * kQuarterBytes;
      const char* r1 = rl + (buf + 1 > 2 ? buf - 2 : buf + 1) * kQuarterBytes;
      const char* r2 = rl + (buf + 2 > 2 ? buf - 1 : buf + 2) * kQuarterBytes;
      quarter(I0{}, PH, r0, r1, fa, fb, nullptr);
      quarter(I1{}, PH, r1, r2, fb, fa, nullptr);
      quarter(I2{}, PH, r2, r0, fa, fb, nullptr);  // (4 k + 3) % 3 = 4 k % 3
      quarter(I3{}, PH, r0, r1, fb, fa, q_in0 + (k & 1) * 4 * kRoleImg);   // 4 (k + 1) % 3 = (4 k + 1) % 3
      ROLE_STAMP(14)
      if (k + 1 < nk) slide(base);
      ROLE_STAMP(15)
      buf = buf == 2 ? 0 : buf + 1;
    };
#pragma clang loop unroll(disable)
    for (int k = 0; k < nk; k += 2) {
      diamond(I0{}, k);
      if (k + 1 < nk) diamond(I1{}, k + 1);
    }
    // the window of the last diamond goes back to memory (its phase: (nk - 1) & 1); e0 -- the image out is free -- was the
    // barrier inside the last quarter
    if ((nk - 1) & 1) tiles_out(4); else tiles_out(0);
    barrier();                                   // e1
    barrier();                                   // e2
    if ((nk - 1) & 1) tiles_out(0); else tiles_out(4);
    barrier();                                   // e3
  }
  ROLE_STAMP_WRITE
  BT2_CLOCK_END
}

// ---- few columns (partial spectrum): one launch per WAVEFRONT of diamonds ------------------------------------------
// k_bt2_apply gives a workgroup 16 NW columns and lets it walk all diamonds in order: with the ~100 columns of a
// partial-spectrum solve that is two workgroups on the whole chip, each applying ~n^2 / 8192 diamonds one after the other
// (config C5, n = 24000: 70 000 diamonds, 470 ms).  Diamond (S, k) touches rows 64 S + 1 + 64 k .. + 126 and must follow
// (S, k - 1) and (S + 1, k .. k + 2): all diamonds with the same t = 3 (Smax - S) + k are independent.  Launch t runs
// them side by side, one workgroup per (diamond, 64 columns): window from memory, the same 160 MFMAs with the
// fragments read straight from L2, window back to memory.  n / 64 + 3 n / 64 launches instead of n^2 / 8192 serial steps.
__global__ __launch_bounds__(256) void k_bt2_wave(const double* __restrict__ sb_all, SbLayout SL,
                                                  const int* __restrict__ dia_off, double* __restrict__ z_all,
                                                  long long stride_z, int ncols, int t) {
  __shared__ double stg_all[4][16 * 18];
  const int n = SL.n;
  const int S = SL.ngroups - 1 - (int)blockIdx.x;
  const int k = t - 3 * (int)blockIdx.x;
  if (S < 0 || k < 0) return;
  const int d0 = dia_off[S], nk = dia_off[S + 1] - d0;
  if (k >= nk) return;
  const double* sb = sb_all + (size_t)blockIdx.z * SL.slab;
  const double* frag = sb + SL.frag + (size_t)(d0 + k) * kFragDoubles;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int col = ((int)blockIdx.y * 4 + w) * 16 + fr;
  const bool col_ok = col < ncols;
  const int win = S * kG + 1 + kB * k;
  double* stg = stg_all[w];
  const int gc = lane >> 3, gr = (lane & 7) * 2;
  const int col_a = ((int)blockIdx.y * 4 + w) * 16 + gc, col_b = col_a + 8;
  double* z_mat = z_all + (size_t)blockIdx.z * stride_z;
  double* za = z_mat + (size_t)(col_a < ncols ? col_a : ncols - 1) * n + gr;
  double* zb = z_mat + (size_t)(col_b < ncols ? col_b : ncols - 1) * n + gr;
  d4 zt[8];
  // window -> accumulator layout through the wave's transposition tile (same maps as k_bt2_apply)
#pragma unroll
  for (int rt = 0; rt < 8; ++rt) {
    const int row0 = win + 16 * rt;
    const int rs = row0 < n - 16 ? row0 : n - 16;
    const int shift = row0 < n - 16 ? 0 : row0 - (n - 16);
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (rs >= 0) { a0 = za[rs]; a1 = za[rs + 1]; b0 = zb[rs]; b1 = zb[rs + 1]; }
    stg[gc * 18 + gr] = a0; stg[gc * 18 + gr + 1] = a1;
    stg[(gc + 8) * 18 + gr] = b0; stg[(gc + 8) * 18 + gr + 1] = b1;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + fk + shift;
      const double x = stg[fr * 18 + (i < 16 ? i : 15)];
      zt[rt][r] = (col_ok && row0 + 4 * r + fk < n) ? x : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // the 160 MFMAs, fragments eight ahead
  constexpr int kAhead = 8;
  double fq[kAhead];
#pragma unroll
  for (int j = 0; j < kAhead; ++j) fq[j] = frag[frag_off(j, lane)];
  d4 wa = d4{0, 0, 0, 0};
#pragma unroll
  for (int f = 0; f < kDiaFrags; ++f) {
    const int st = 3 - f / kMiniFrags, p = f % kMiniFrags;
    const double a = fq[f % kAhead];
    if (p < 20) {
      const int rt = st + p / 4, r = p % 4;
      if (p == 0) wa = d4{0, 0, 0, 0};
      wa = __builtin_amdgcn_mfma_f64_16x16x4f64(a, zt[rt][r], wa, 0, 0, 0);
    } else {
      const int jj = p - 20, r = jj / 5, rt = st + jj % 5;
      zt[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, wa[r], zt[rt], 0, 0, 0);
    }
    if (f + kAhead < kDiaFrags) fq[f % kAhead] = frag[frag_off(f + kAhead, lane)];
  }
  // window back to memory
#pragma unroll
  for (int rt = 0; rt < 8; ++rt) {
    const int row0 = win + 16 * rt;
#pragma unroll
    for (int r = 0; r < 4; ++r) stg[fr * 18 + 4 * r + fk] = zt[rt][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const double a0 = stg[gc * 18 + gr], a1 = stg[gc * 18 + gr + 1];
    const double b0 = stg[(gc + 8) * 18 + gr], b1 = stg[(gc + 8) * 18 + gr + 1];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (col_a < ncols) {
      if (row0 + gr < n) za[row0] = a0;
      if (row0 + gr + 1 < n) za[row0 + 1] = a1;
    }
    if (col_b < ncols) {
      if (row0 + gr < n) zb[row0] = b0;
      if (row0 + gr + 1 < n) zb[row0 + 1] = b1;
    }
  }
}

}  // namespace

// ================================================================================================================
// Layout
// K slices of the SYMM X = A22 V: with few matrices its launch has only n / 64 tiles per matrix, each walking a K range of
// up to n (longest first, but the longest IS the critical path): slices of the K range give the launch 2 - 8 times the
// workgroups and a fraction of the critical path.  Config C5 (one 24000 x 24000 matrix): 596 -> 307 ms.
// Whether the band reduction of order-n matrices runs its SYMM on k_symm3 (symm3.hip; the launcher may still decline a
// panel -- too few tiles late in the reduction -- which then takes the two triangular-operand launches below)
bool symm3_for(int n) {
  static const int env = [] { const char* e = getenv("SPRINGCRAFT_SYMM3"); return e ? atoi(e) : 1; }();
  return env != 0 && n >= 512 && (n & 1) == 0;
}

int symm_split_for(int n, int batch) {
  static const int env = [] { const char* e = getenv("SPRINGCRAFT_SYMM_SPLIT"); return e ? atoi(e) : 0; }();
  if (env >= 1) return std::min(env, 16);
  if (symm3_for(n)) {
    // k_symm3: a work item is (128-row tile, K slice), all of equal length; 256 persistent workgroups want a few rounds
    // of them (a launch holds half the batch from 32 matrices on: the two half batches run on two streams)
    const long long items = (long long)((n + 127) / 128) * (batch >= 32 ? batch / 2 : batch);
    // (C5, one n = 24000 matrix = 188 tile rows, tools/r06_cfgs.sh: 3 / 4 / 5 / 6 / 8 slices 1143 / 1114 / 1105 / 1092 / 1083 ms)
    if (items >= 768) return 1;
    return (int)std::min<long long>(16, (1536 + items - 1) / items);
  }
  const long long tiles = (long long)batch * ((n + 63) / 64);
  if (tiles >= 1024 || n < 2048) return 1;
  // (one 24000 x 24000 matrix: 3 slices 338 ms, 6 slices 307 ms, 8 slices 378 ms of SYMM time)
  // Round 5, same matrix, `tools/r05_symm_sweep.sh`: 4 slices 407 ms, 5: 312, 6: 325, 7: 299, 8: 395, 9: 285, 10: 295,
  // 12: 303 -- slice counts that share a factor with the 8 XCDs put the same slice of every tile on the same XCD (the
  // workgroups are dealt round-robin) and with it the long K ranges of the triangle: ODD counts spread them.  Nine where six
  // were taken (at most 409 tiles: one large matrix or a few); the smaller counts of larger launches are unchanged
  // (n = 6000, one to four matrices: 6, 7 and 9 slices within 1 %).
  // The middle regime follows the same rule (`tools/r05_symm_mid.sh`, 6 x n = 6000 = 564 tiles: 4 slices 38.9 ms of SYMM
  // per step, 5 slices 31.0): an even count takes the next odd one.
  // (ceil(2048 / tiles) is 3 for 683 .. 1023 tiles, 4 for 512 .. 682, 5 for 410 .. 511, 6 and more below: the only even
  // count the rule replaces is the measured 4 -> 5)
  const long long s0 = std::min<long long>(6, (2048 + tiles - 1) / tiles);
  return s0 >= 6 ? 9 : (s0 == 4 ? 5 : (int)s0);
}

size_t sb_slab_doubles(int n, int batch, SbLayout* out) {
  SbLayout L{};
  L.n = n;
  long long off = 0;
  auto take = [&](long long cnt) { long long o = off; off += (cnt + 31) / 32 * 32; return o; };
  const int nchunk = (n + kQrRows - 1) / kQrRows + 1;
  L.vw = take((long long)n * 4 * kB);
  L.wv = take((long long)n * 4 * kB);
  L.xv = take((long long)n * 3 * kB);
  L.symm_split = symm_split_for(n, batch);
  L.xsplit = L.symm_split > 1 ? take((long long)2 * L.symm_split * n * kB) : 0;
  L.qrpart = take((long long)2 * nchunk * kB);
  L.qrpiv = take(2 * (kB + 8));
  L.qrpart8 = take((long long)nchunk * kIb * kB);
  L.small = take((long long)kSmallSplit * kB * 3 * kB);
  L.cmat = take(3 * kB * kB);
  L.small2 = take((long long)kSmallSplit * 2 * kB * kB);
  L.p2 = take(2 * kB * kB);
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
  L.frag = take(ndia * kFragDoubles);
  L.tau2 = take(ndia * kG);
  L.slab = off;
  if (out) *out = L;
  return (size_t)off;
}

int sb_desc_count(int n, int batch) {
  const int npanels = n / kB + 1;
  return npanels * kDescKinds * batch;
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
  ScopedEvents<3> ev;
  const bool prof = ctx->profiling && ms_stage1 && ms_stage2;
  if (prof) {
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    SC_HIP(ctx, hipEventRecord(ev[0], st));
  }

  // ---- descriptors of stage 1: per panel kDescKinds kinds x batch
  // Panels are taken in PAIRS where the trailing matrix is large enough: the second panel of a pair is factored from
  // columns that received the first panel's update alone (record D), its X = A22 V2 is computed from the NOT yet updated
  // trailing matrix and corrected by - [V1|W1] ([W1|V1]^T V2) (records Q, Corr), and one trailing update
  // A22 -= [V1|W1|V2|W2] [W1|V1|W2|V2]^T (K = 256) serves both panels: the lower triangle is read and written once per
  // 128 columns of band instead of once per 64 (at K = 128 the C traffic of a tile takes as long as its MFMAs).
  int npanels = 0;
  while (n - (npanels + 1) * kB >= 2) ++npanels;
  static const bool pair_env = [] { const char* e = getenv("SPRINGCRAFT_PANEL_PAIRS"); return !e || atoi(e) != 0; }();
  std::vector<int> role((size_t)npanels, 0);   // 0 single, 1 first of a pair, 2 second of a pair
  if (pair_env)
    for (int p = 0; p + 1 < npanels;) {
      const int m2 = n - (p + 2) * kB;
      if (m2 >= 4 * kB) { role[(size_t)p] = 1; role[(size_t)p + 1] = 2; p += 2; }
      else ++p;
    }
  std::vector<GemmDesc> h((size_t)npanels * kDescKinds * batch);
  for (int p = 0; p < npanels; ++p) {
    const int j0 = p * kB, r0 = j0 + kB, m = n - r0;
    const int rl = role[(size_t)p];
    const long long vw_w = rl == 2 ? 3 * kB : kB, wv_w = rl == 2 ? 2 * kB : 0;   // columns W goes to
    for (int b = 0; b < batch; ++b) {
      double* A = d_a + (size_t)b * stride_a;
      double* sb = d_sb_ws + (size_t)b * SL.slab;
      double* a22 = A + (size_t)r0 * n + r0;
      GemmDesc* g = &h[((size_t)p * batch + b) * kDescKinds];
      // X1 = L V
      GemmDesc X1{};
      X1.a = a22; X1.sa_i = 1; X1.sa_k = n; X1.a_tri = 1;
      X1.b = sb + SL.xv + (size_t)2 * kB * n + r0; X1.sb_k = 1; X1.sb_j = n;
      X1.c = sb + SL.xv + r0; X1.ldc = n;
      X1.m = m; X1.n = kB; X1.k = m; X1.alpha = 1.0; X1.beta = 0.0;
      if (SL.symm_split > 1) {   // K slices into their own buffers, summed into X1 / X2 by k_sum_xslices
        X1.c = sb + SL.xsplit + r0;
        X1.split_stride = (long long)n * kB;
      }
      g[0] = X1;
      // X2 = strict(L)^T V
      GemmDesc X2 = X1;
      X2.sa_i = n; X2.sa_k = 1; X2.a_tri = 2;
      X2.c = sb + SL.xv + (size_t)kB * n + r0;
      if (SL.symm_split > 1) X2.c = sb + SL.xsplit + (size_t)SL.symm_split * n * kB + r0;
      g[1] = X2;
      // V^T [X1 | X2 | V], split-K slices
      GemmDesc P{};
      P.a = sb + SL.xv + (size_t)2 * kB * n + r0; P.sa_i = n; P.sa_k = 1;
      P.b = sb + SL.xv + r0; P.sb_k = 1; P.sb_j = n;
      P.c = sb + SL.small; P.ldc = kB;
      P.m = kB; P.n = 3 * kB; P.k = m; P.alpha = 1.0; P.beta = 0.0;
      P.split_stride = (long long)kB * 3 * kB;
      g[2] = P;
      // W = [X1 | X2 | V] C  -> its column block of [V|W..] and of [W|V..]
      GemmDesc W{};
      W.a = sb + SL.xv + r0; W.sa_i = 1; W.sa_k = n;
      W.b = sb + SL.cmat; W.sb_k = 1; W.sb_j = 3 * kB;
      W.c = sb + SL.vw + (size_t)vw_w * n + r0; W.ldc = n;
      W.m = m; W.n = kB; W.k = 3 * kB; W.alpha = 1.0; W.beta = 0.0;
      g[3] = W;
      // (the [W|V] panel holds -[W|V], so that every product that subtracts -- the trailing updates, the correction of X
      // in a pair -- has alpha = 1: the role-split kernel k_gemm3 keeps C itself in its accumulators and folds no sign;
      // negation is exact, the results are bit for bit those of alpha = -1 on [W|V])
      W.c = sb + SL.wv + (size_t)wv_w * n + r0;
      W.alpha = -1.0;
      g[4] = W;
      // trailing update, lower triangle: A22 -= [V|W] [W|V]^T (single panel) or the four-block form (second of a pair)
      GemmDesc R{};
      R.a = sb + SL.vw + r0; R.sa_i = 1; R.sa_k = n;
      R.b = sb + SL.wv + r0; R.sb_k = n; R.sb_j = 1;
      R.c = a22; R.ldc = n;
      R.m = m; R.n = m; R.k = rl == 2 ? 4 * kB : 2 * kB; R.alpha = 1.0; R.beta = 1.0;   // b = -[W|V]
      // (with k_symm3 in use the trailing updates also keep the first super-diagonal entry of every even row: symm3.hip)
      R.lower_only = symm3_for(n) ? 2 : 1;
      g[5] = R;
      // first of a pair: only the next panel's 64 columns (and the band block above them) get this panel's update now
      GemmDesc D = R;
      D.n = kB; D.k = 2 * kB;
      g[6] = D;
      // second of a pair (rows r0 .. of the first panel's blocks): P2 = [W1|V1]^T V2, X1 -= [V1|W1] P2
      GemmDesc Q{};
      Q.a = sb + SL.wv + r0; Q.sa_i = n; Q.sa_k = 1;
      Q.b = sb + SL.xv + (size_t)2 * kB * n + r0; Q.sb_k = 1; Q.sb_j = n;
      Q.c = sb + SL.small2; Q.ldc = 2 * kB;
      Q.m = 2 * kB; Q.n = kB; Q.k = m; Q.alpha = 1.0; Q.beta = 0.0;
      Q.split_stride = (long long)2 * kB * kB;
      g[7] = Q;
      GemmDesc Cr{};
      Cr.a = sb + SL.vw + r0; Cr.sa_i = 1; Cr.sa_k = n;
      Cr.b = sb + SL.p2; Cr.sb_k = 1; Cr.sb_j = 2 * kB;
      Cr.c = sb + SL.xv + r0; Cr.ldc = n;
      Cr.m = m; Cr.n = kB; Cr.k = 2 * kB; Cr.alpha = 1.0; Cr.beta = 1.0;   // P2 comes out negated (a = -[W1|V1])
      g[8] = Cr;
    }
  }
  // regroup so that each launch's records are contiguous: [panel][kind][batch]
  std::vector<GemmDesc> hs(h.size());
  for (int p = 0; p < npanels; ++p)
    for (int b = 0; b < batch; ++b) {
      const GemmDesc* g = &h[((size_t)p * batch + b) * kDescKinds];
      GemmDesc* o = &hs[(size_t)p * kDescKinds * batch];
      for (int kd = 0; kd < kDescKinds; ++kd) o[(size_t)kd * batch + b] = g[kd];
    }
  if (!hs.empty()) SC_TRY(sc_stage_upload(ctx, d_descs, hs.data(), hs.size() * sizeof(GemmDesc)));
  const std::vector<int> doff = dia_offsets(n);
  SC_TRY(sc_stage_upload(ctx, d_dia_off, doff.data(), doff.size() * sizeof(int)));

  // tau of columns without a reflector must read 0
  for (int b = 0; b < batch; ++b)
    SC_HIP(ctx, hipMemsetAsync(d_tri_ws + (size_t)b * TL.slab + TL.tau, 0, sizeof(double) * n, st));

  const size_t lds_qr = sizeof(double) * ((size_t)kB * (kQrRows + 1) + kB + kQrRows + 4 * kB);
  const size_t lds_qr_blk = sizeof(double) * ((size_t)(kIb + 1) * (kQrRows + 1) + kB + kQrRows + 4 * kB);
  const size_t lds_small = sizeof(double) * 4 * kB * (kB + 1);
  const size_t lds_blk_a = sizeof(double) * ((size_t)kB * (kQrRows + 1) + kQrRows + 8 * kB);
  const size_t lds_blk_b = sizeof(double) * ((size_t)kB * (kQrRows + 1) + kIb * kB + 4 * kB);
  static const bool blocked_qr = getenv("SPRINGCRAFT_QR_UNBLOCKED") == nullptr;
  PhaseTimer t_qr(ctx, "panel_qr", st), t_symm(ctx, "symm", st), t_syr2k(ctx, "syr2k", st), t_bulge(ctx, "bulge", st);
  // ---- the cooperative panel kernel (k_panel_coop) for tall panels of a few matrices: SPRINGCRAFT_QR_COOP = 0 keeps the
  // chunked launches, SPRINGCRAFT_QR_COOP_MIN = rows from which a panel takes it (default: above k_panel_wg's 4096)
  static const int env_coop = [] { const char* e = getenv("SPRINGCRAFT_QR_COOP"); return e ? atoi(e) : 1; }();
  // (one matrix, per panel: 0.77 ms by the chunked launches whatever the height, 0.3 - 0.5 ms by k_panel_wg's single
  // workgroup, 0.25 - 0.3 ms here -- C5: panel QR 289 -> 109 ms, a single n = 6000 matrix: 52 -> 23 ms.  With four and more
  // matrices the single-workgroup kernels run them side by side and keep their range; the taller panels come here)
  static const int coop_min_env = [] { const char* e = getenv("SPRINGCRAFT_QR_COOP_MIN"); return e ? std::max(2 * kB, atoi(e)) : 0; }();
  const int coop_min_few = ctx->coop_min_rows > 0 ? ctx->coop_min_rows : (coop_min_env > 0 ? coop_min_env : 300);
  const int coop_min_many = ctx->coop_min_rows > 0 ? ctx->coop_min_rows : (coop_min_env > 0 ? coop_min_env : 12 * 512 + 1);
  const int coop_min = std::min(coop_min_few, coop_min_many);
  // every workgroup of a launch waits for all others: matrices x workgroups stays well below the CUs of the device
  const int coop_budget = ctx->num_cus * 3 / 4;
  const int coop_gmax = (std::max(n - kB, 1) + kCoopRows - 1) / kCoopRows;
  v4i* coop_recs = nullptr;
  int* coop_ctl = nullptr;
  // (only when the whole batch runs on one stream: the parts of a split batch must not be factored by different kernels --
  // other reduction trees, other last bits -- or the same structure at two batch positions gives different eigenvalues)
  static const int env_s1_early = [] { const char* e = getenv("SPRINGCRAFT_STAGE1_STREAMS"); return e ? atoi(e) : 0; }();
  const bool one_stream = std::max(1, std::min(env_s1_early > 0 ? env_s1_early : ((batch >= 32 && !prof) ? 2 : 1), std::min(batch, 4))) == 1;
  if ((ctx->coop_min_rows > 0 || (ctx->coop_min_rows < 0 && env_coop != 0)) && ctx->coop_ok != 0 && n - kB >= coop_min &&
      coop_gmax <= kCoopMaxG && one_stream) {
    if (ctx->coop_attr < 0)
      ctx->coop_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_panel_coop),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoopLdsBytes) == hipSuccess;
    const int nb_max = std::min(batch, coop_budget / coop_gmax);
    if (ctx->coop_attr == 1 && nb_max >= 1) {
      const size_t rec_bytes = (size_t)batch * coop_recs_per_matrix(coop_gmax) * sizeof(v4i);
      const size_t ctl_bytes = align_up((size_t)batch * 8 * sizeof(int), 256);   // one control record per matrix
      SC_TRY(sc_reserve_dc_aux(ctx, ctl_bytes + rec_bytes));
      coop_ctl = reinterpret_cast<int*>(ctx->dc_aux);
      coop_recs = reinterpret_cast<v4i*>(reinterpret_cast<char*>(ctx->dc_aux) + ctl_bytes);
      // (sequence numbers are unique within a solve only)
      SC_HIP(ctx, hipMemsetAsync(ctx->dc_aux, 0, ctl_bytes + rec_bytes, st));
    }
  }
  // k_symm3 writes X into the X1 block of [X1 | X2 | V]; the X2 block reads zero for the whole solve (the panels at the
  // end of the reduction that fall back to the two triangular-operand launches rewrite their rows of it themselves)
  const bool use_symm3 = symm3_for(n) && (n & 1) == 0;
  // (whether a panel takes it is decided for the SMALLEST part of a batch that is split over streams, and then holds for
  // every part: the same structure at two batch positions must not meet two kernels -- other summation orders, other last
  // bits; found by tools/test_matrix.sh with three parts of unequal size, profiles/r06_test_matrix.txt)
  const int symm3_parts = std::max(1, std::min(env_s1_early > 0 ? env_s1_early : ((batch >= 32 && !prof) ? 2 : 1), std::min(batch, 4)));
  const int symm3_nb_min = std::max(1, batch / symm3_parts);
  if (use_symm3)
    for (int b = 0; b < batch; ++b)
      SC_HIP(ctx, hipMemsetAsync(d_sb_ws + (size_t)b * SL.slab + SL.xv + (size_t)kB * n, 0, sizeof(double) * (size_t)kB * n, st));
  // One panel of the matrices [lo, hi) on `ps`: QR of the panel, X = A22 V, the small products, W, the trailing update
  // its role asks for (single: SYR2K; first of a pair: the next panel's columns only; second: the joint update).
  auto run_panel = [&](int p, int lo, int hi, hipStream_t ps, bool timed) -> int {
    struct StreamScope {   // launch_gemm_f64 launches on the context's stream
      sc_ctx* c; hipStream_t old;
      StreamScope(sc_ctx* c_, hipStream_t s) : c(c_), old(c_->stream) { c->stream = s; }
      ~StreamScope() { c->stream = old; }
    } scope(ctx, ps);
    const int nb = hi - lo;
    const int rl = role[(size_t)p];
    double* a_h = d_a + (size_t)lo * stride_a;
    double* tri_h = d_tri_ws + (size_t)lo * TL.slab;
    double* sb_h = d_sb_ws + (size_t)lo * SL.slab;
    const int j0 = p * kB, r0 = j0 + kB, m = n - r0;
    const int nr = std::min(kB, m - 1);
    const int nchunks = (m + kQrRows - 1) / kQrRows;
    const dim3 qgrid((unsigned)nchunks, (unsigned)nb);
    SbLayout SQ = SL;     // where the panel QR leaves V: the second panel of a pair uses the second half of [V|W..], [W|V..]
    if (rl == 2) { SQ.vw += (long long)2 * kB * n; SQ.wv += (long long)2 * kB * n; }
    if (timed) t_qr.start();
    // (one workgroup per matrix while the panel is short enough for its rows to sit in the registers of 1024 threads --
    // also for a single matrix: N = 512 two-stage 45 -> 37 ms; SPRINGCRAFT_QR_WG = 0 keeps the chunked launches)
    static const int env_wg = [] { const char* e = getenv("SPRINGCRAFT_QR_WG"); return e ? atoi(e) : -1; }();
    // (round 4: panels of 4097 .. 6144 rows -- the first 29 of n = 6000 -- by 512 threads with up to 12 rows each: twice
    // the registers per thread; SPRINGCRAFT_QR_WG = 1 keeps them on the chunked launches)
    const bool use_wg = nr == kB && m <= 4 * kWgThreads && env_wg != 0;
    // (not for a few matrices: one workgroup streams its panel at the rate of ONE CU -- a single N = 2000 structure's
    // panel QRs take 70 instead of 52 ms that way)
    const bool use_wg512 = nr == kB && !use_wg && m <= 12 * 512 && env_wg != 0 && env_wg != 1 && nb >= 4;
    const int coop_g = (m + kCoopRows - 1) / kCoopRows;
    const bool use_coop = coop_recs != nullptr && nr == kB && m >= (nb < 4 ? coop_min_few : coop_min_many) &&
                          nb * coop_g <= coop_budget && ps == st;
    if (use_coop) {
      // (test hook, sc_dbg_set_panel_coop_fail: from this panel on the matrices' abort flags are up, as after a time-out)
      if (p == ctx->coop_fail_panel) SC_HIP(ctx, hipMemsetAsync(coop_ctl + 8 * lo, 1, sizeof(int) * 8 * nb, ps));
      hipLaunchKernelGGL(k_panel_coop, dim3((unsigned)coop_g, (unsigned)nb), dim3(kCoopRows), kCoopLdsBytes, ps, a_h, stride_a,
                         tri_h, TL, sb_h, SQ, j0, coop_g, coop_recs + (size_t)lo * coop_recs_per_matrix(coop_g),
                         coop_ctl + 8 * lo, (p + 1) * 128);
      // the take-over: returns at once unless a wait of the launch above (or of an earlier panel) timed out
      hipLaunchKernelGGL(k_panel_serial, dim3((unsigned)nb), dim3(1024), 0, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0,
                         (const int*)(coop_ctl + 8 * lo), ctx->d_status);
      ++ctx->cnt_coop_launches;
    } else if (use_wg) {
      const size_t lds_wg = sizeof(double) * (size_t)(2 * kWgWaves * 8 + 16 + 8 + 8 * kB + kWgWaves * kB * 8);
      const int ru = (m + kWgThreads - 1) / kWgThreads;
      const dim3 g1((unsigned)nb), b1((unsigned)kWgThreads);
      if (ru <= 1) hipLaunchKernelGGL((k_panel_wg<1, 8>), g1, b1, lds_wg, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0);
      else if (ru == 2) hipLaunchKernelGGL((k_panel_wg<2, 8>), g1, b1, lds_wg, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0);
      else if (ru == 3) hipLaunchKernelGGL((k_panel_wg<3, 4>), g1, b1, lds_wg, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0);
      else hipLaunchKernelGGL((k_panel_wg<4, 2>), g1, b1, lds_wg, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0);
    } else if (use_wg512) {
      constexpr int kW = 512 / 64;
      const size_t lds_wg = sizeof(double) * (size_t)(2 * kW * 8 + 16 + 8 + 8 * kB + kW * kB * 8);
      const int ru = (m + 511) / 512;
      const dim3 g1((unsigned)nb), b1(512u);
      if (ru <= 10) hipLaunchKernelGGL((k_panel_wg<10, 1, 512>), g1, b1, lds_wg, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0);
      else hipLaunchKernelGGL((k_panel_wg<12, 1, 512>), g1, b1, lds_wg, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0);
    } else if (nr == kB && blocked_qr) {
      // blocked panel: inner blocks of 8 columns, their reflectors applied to the rest of the panel at once
      hipLaunchKernelGGL(k_panel_qr, qgrid, dim3(256), lds_qr_blk, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0, 0, nr, kIb);
      for (int c0 = 0; c0 < kB; c0 += kIb) {
        for (int j = c0 + 1; j < c0 + kIb; ++j)
          hipLaunchKernelGGL(k_panel_qr, qgrid, dim3(256), lds_qr_blk, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0, j, nr,
                             c0 + kIb);
        // (their LDS image holds columns c0 .. kB-1 only)
        const size_t cut = sizeof(double) * (size_t)c0 * (kQrRows + 1);
        hipLaunchKernelGGL(k_pqr_blk_a, qgrid, dim3(256), lds_blk_a - cut, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0, c0);
        if (c0 + kIb < kB)
          hipLaunchKernelGGL(k_pqr_blk_b, qgrid, dim3(256), lds_blk_b - cut, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0, c0);
      }
    } else {
      for (int j = 0; j <= nr; ++j)
        hipLaunchKernelGGL(k_panel_qr, qgrid, dim3(256), lds_qr, ps, a_h, stride_a, tri_h, TL, sb_h, SQ, j0, j, nr, kB);
    }
    if (timed) t_qr.stop();
    const GemmDesc* g = d_descs + (size_t)p * kDescKinds * batch;   // [kind][batch]
    if (timed) t_symm.start();
    // X = A22 V: one launch of k_symm3 (X into X1; X2 was zeroed for the whole solve) while the panel has enough tiles
    // for it, else X1 = L V and X2 = strict(L)^T V by two triangular-operand launches of k_gemm2
    if (use_symm3 && symm3_would_take(ctx, symm3_nb_min, m, SL.symm_split, /*aligned16=*/true) &&
        launch_symm3(ctx, g + lo, nb, m, SL.symm_split, /*aligned16=*/true, /*any_size=*/true) == SC_OK) {
      if (SL.symm_split > 1)
        hipLaunchKernelGGL(k_sum_xslices, dim3((unsigned)((m + 255) / 256), kB, (unsigned)nb), dim3(256), 0, ps, sb_h, SL, r0);
    } else {
      SC_TRY(launch_gemm_f64(ctx, g + lo, nb, m, kB, kGemmTile, SL.symm_split, false, true, kGemmAmBk));           // X1 = L V
      SC_TRY(launch_gemm_f64(ctx, g + batch + lo, nb, m, kB, kGemmTile, SL.symm_split, false, true, kGemmAkBk));   // X2 = strict(L)^T V
      if (SL.symm_split > 1)
        hipLaunchKernelGGL(k_sum_xslices, dim3((unsigned)((m + 255) / 256), 2 * kB, (unsigned)nb), dim3(256), 0, ps, sb_h, SL, r0);
    }
    if (rl == 2) {   // the trailing matrix has not seen the first panel's update yet: X1 -= [V1|W1] ([W1|V1]^T V2)
      SC_TRY(launch_gemm_f64(ctx, g + 7 * batch + lo, nb, 2 * kB, kB, kGemmTile, kSmallSplit, false, false, kGemmAkBk));
      hipLaunchKernelGGL(k_sum_p2, dim3((unsigned)(2 * kB * kB / 256), (unsigned)nb), dim3(256), 0, ps, sb_h, SL);
      SC_TRY(launch_gemm_f64(ctx, g + 8 * batch + lo, nb, m, kB, kGemmTile, 1, false, false, kGemmAmBk));
    }
    if (timed) t_symm.stop();
    SC_TRY(launch_gemm_f64(ctx, g + 2 * batch + lo, nb, kB, 3 * kB, kGemmTile, kSmallSplit, false, false, kGemmAkBk));
    hipLaunchKernelGGL(k_sb_small, dim3((unsigned)nb), dim3(1024), lds_small, ps, tri_h, TL, sb_h, SL, j0);
    if (nb == batch) {
      SC_TRY(launch_gemm_f64(ctx, g + 3 * batch, 2 * batch, m, kB, kGemmTile, 1, false, false, kGemmAmBk));
    } else {
      SC_TRY(launch_gemm_f64(ctx, g + 3 * batch + lo, nb, m, kB, kGemmTile, 1, false, false, kGemmAmBk));
      SC_TRY(launch_gemm_f64(ctx, g + 4 * batch + lo, nb, m, kB, kGemmTile, 1, false, false, kGemmAmBk));
    }
    if (timed) t_syr2k.start();
    if (rl == 1)
      SC_TRY(launch_gemm_f64(ctx, g + 6 * batch + lo, nb, m, kB, kGemmTile, 1, false, false, kGemmAmBn));
    else
      // (records of one launch share (m, m, K); operands start at even rows of buffers with even leading dimension n)
      if (launch_gemm3_uniform(ctx, g + 5 * batch + lo, nb, m, m, rl == 2 ? 4 * kB : 2 * kB, kGemmAmBn, /*lower=*/use_symm3 ? 2 : 1, 1.0, 1.0,
                               /*aligned16=*/(n & 1) == 0 && (r0 & 1) == 0) != SC_OK)
        SC_TRY(launch_gemm_f64(ctx, g + 5 * batch + lo, nb, m, m, kGemmTile, 1, false, false, kGemmAmBn, /*lower_grid=*/true));
    if (timed) t_syr2k.stop();
    return SC_OK;
  };
  // Two parts of a large batch on separate streams: the latency-bound panel QR of one part runs beside the GEMMs of the
  // other (618 -> 591 ms per step at 64 matrices).  A profiled solve keeps everything on one stream, because the
  // kernel-group times (panel_qr / symm / syr2k) are event brackets on that stream: its band reduction is therefore
  // about 30 ms longer than in an unprofiled solve.
  static const int env_s1 = [] { const char* e = getenv("SPRINGCRAFT_STAGE1_STREAMS"); return e ? atoi(e) : 0; }();
  const int s1_want = env_s1 > 0 ? env_s1 : ((batch >= 32 && !prof) ? 2 : 1);
  const int s1_parts = std::max(1, std::min(s1_want, std::min(batch, 4)));
  if (s1_parts > 1) {
    struct SideBySide {   // (k_gemm3 leaves CUs to the other parts' kernels while this is set: gemm3_would_take)
      sc_ctx* c;
      explicit SideBySide(sc_ctx* c_) : c(c_) { c->gemm3_side_by_side = true; }
      ~SideBySide() { c->gemm3_side_by_side = false; }
    } side_by_side(ctx);
    SC_TRY(sc_aux_stream(ctx));
    SC_TRY(sc_side_streams(ctx, s1_parts - 1));
    SC_HIP(ctx, hipEventRecord(ctx->aux_fork, st));
    for (int q = 1; q < s1_parts; ++q) SC_HIP(ctx, hipStreamWaitEvent(ctx->side_streams[q - 1], ctx->aux_fork, 0));
    int rc_parts = SC_OK;
    for (int p = 0; p < npanels && rc_parts == SC_OK; ++p)
      for (int q = 0; q < s1_parts && rc_parts == SC_OK; ++q) {
        const int lo = (int)((long long)batch * q / s1_parts), hi = (int)((long long)batch * (q + 1) / s1_parts);
        rc_parts = run_panel(p, lo, hi, q == 0 ? st : ctx->side_streams[q - 1], q == 0);
      }
    // (also after an error: the side streams are joined before anything returns, so that what they still have queued
    // is ordered before whatever the caller enqueues next on the main stream)
    for (int q = 1; q < s1_parts; ++q) {
      SC_HIP(ctx, hipEventRecord(ctx->side_joins[q - 1], ctx->side_streams[q - 1]));
      SC_HIP(ctx, hipStreamWaitEvent(st, ctx->side_joins[q - 1], 0));
    }
    SC_TRY(rc_parts);
  } else {
    for (int p = 0; p < npanels; ++p) SC_TRY(run_panel(p, 0, batch, st, true));
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
    // The launches of one chase are strictly ordered and every one of them ends with a partly filled last round of
    // workgroups.  Parts of the batch on separate streams run the same launches independently of each other, so the
    // tail of one part's launch is filled by the next launch of another part.
    // ---- persistent chase (k_bulge_chase: one launch, sweeps claimed dynamically) while the stage is latency-bound;
    // else, or with SPRINGCRAFT_BULGE_PERSISTENT=0, one launch per wavefront below
    static const int env_persist = [] { const char* e = getenv("SPRINGCRAFT_BULGE_PERSISTENT"); return e ? atoi(e) : 1; }();
    const int persist = ctx->chase_mode >= 0 ? ctx->chase_mode : env_persist;
    bool chased = false;
    // Which form (profiles/r04_bulge_sweep.txt, batch x n: pair / one sweep per workgroup / per-wavefront launches, ms):
    // 16 x 6000: 126 / 115 / 169, 32 x 3000: 63 / 56 / 82, 64 x 3000: 92 / 99 / 124, 32 x 6000: 182 / 223 / 256,
    // 64 x 6000: 335 / 435 / 436.  While the stage is latency-bound (batch * n / 128 <= 1100) one sweep per workgroup has
    // the shorter dependent chain; above that the stage is bound by bytes and the pair form moves half of them.  With
    // fewer matrices than XCDs (a matrix is confined to one XCD) only while the order is moderate: a single n = 24000
    // matrix chases 1.6 x faster with its ~188 tasks per wavefront spread over the whole chip (411 vs 676 ms)
    static const bool use_pair = [] { const char* e = getenv("SPRINGCRAFT_BULGE_PAIR"); return !e || atoi(e) != 0; }();
    static const int force_pair = [] { const char* e = getenv("SPRINGCRAFT_BULGE_PAIR"); return e ? atoi(e) : -1; }();
    // SPRINGCRAFT_PAIR_LOADER = 1: the pair form with its loader waves (k_bulge_pair<1>, 768 threads: measured, slower --
    // profiles/r06_pair_stamps.txt; default: the 512-thread form)
    static const bool pair_loader = [] { const char* e = getenv("SPRINGCRAFT_PAIR_LOADER"); return e && atoi(e) != 0; }();
    // SPRINGCRAFT_PAIR_EARLY = 1: the pair form looks at its predecessor a step ahead (measured: 317-319 against 312-313 ms
    // at 64 x n = 6000, profiles/r06_pair_ab.txt; default: when a step begins, as in round 4)
    static const int pair_early = [] { const char* e = getenv("SPRINGCRAFT_PAIR_EARLY"); return (e && atoi(e) != 0) ? 1 : 0; }();
    if (ctx->pair_attr < 0)   // per device, hence per context: the caller made ctx->device current
      ctx->pair_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bulge_pair<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPairLdsBytes) == hipSuccess &&
                       hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bulge_pair<1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPairLdsBytes) == hipSuccess;
    const bool pair_attr = ctx->pair_attr == 1;
    const long long work = (long long)batch * n / 128;
    // SPRINGCRAFT_BULGE_PAIR = 0: never the pair form, 2: the pair form for every persistent chase (tests), else by size
    static const long long kPairWorkMax = [] { const char* e = getenv("SPRINGCRAFT_BULGE_PAIR_MAX"); return e ? atoll(e) : 6000LL; }();
    // (by size only inside the measured range -- profiles/r04_bulge_sweep.txt up to 64 x 6000, profiles/r05_bulge_sweep_ext.txt
    // up to 96 x 6000 and 256 x 3000, B n / 128 = 6000: the pair form wins everywhere there (622 vs 742 / 775 ms, 354 vs
    // 467 / 558) -- and only while every matrix of an XCD has a pair workgroup of its own (one per CU): 512 x 1026 puts 64
    // matrices on the 32 CUs of an XCD, a workgroup then walks the matrices one after the other, 687 ms against 110 by
    // the per-wavefront launches.  ADVICE round 4.)
    if (ctx->nxcd <= 0) SC_TRY(probe_xcd_count(ctx, st));
    const bool pair_measured = work <= kPairWorkMax && (batch + ctx->nxcd - 1) / ctx->nxcd <= std::max(1, ctx->num_cus / ctx->nxcd);
    const bool pair = ctx->chase_form >= 0 ? (ctx->chase_form == 1 && pair_attr)
                                           : use_pair && pair_attr &&
                                                 (force_pair == 2 || (work > 1100 && batch >= 8 && pair_measured));
    // Round 6, "spread": a few LARGE matrices (fewer than XCDs, n > 4500 -- config C5's single n = 24000 matrix has ~188
    // tasks per wavefront, three times the workgroups one XCD holds) are chased by k_bulge_chase<1> from ALL XCDs, their
    // band handed on by write-through stores; until round 5 the largest took one launch per wavefront.  One matrix on one XCD
    // against all XCDs, bulge chasing in ms (tools/spread_sweep.py, profiles/r06_spread_chase.txt): n = 2100 24.3 / 26.4, 3000
    // 35.2 / 38.2, 4200 54.0 / 53.4, 6000 84.2 / 76.9 (2 and 4 matrices: 86 / 78), 9000 144.6 / 116.3, 12000 195.6 / 156.3, 24000
    // (per-wavefront launches) 418 / 318.  SPRINGCRAFT_BULGE_SPREAD = 0 / 1: never / for every chase with fewer matrices than
    // XCDs (tests).
    static const int env_spread = [] { const char* e = getenv("SPRINGCRAFT_BULGE_SPREAD"); return e ? atoi(e) : -1; }();
    // (debug entry sc_dbg_set_chase: mode 5 forces it, modes 3 / 4 keep it off)
    const bool spread = !pair && batch < ctx->nxcd &&
                        (ctx->chase_form == 2 || (ctx->chase_form < 0 && env_spread != 0 && (env_spread == 1 || n > 4500)));
    const bool want_chase =
        persist == 2 || (persist == 1 && (pair || spread || (work <= 2800 && (batch >= 8 || n <= 6144))));
    // a context whose chase ran into its time-out is not asked again (ctx->chase_ok = 0, counted in chase_timeouts):
    // every further attempt could cost another bound's worth of spinning before the fallback
    // XCDs of this device: the kernels bind matrix b to XCD b mod nxcd (an MI355X in SPX mode has 8 XCDs of 32 CUs; a
    // partition of it has fewer, and a matrix bound to an XCD that is not there would never be claimed)
    // (counted on the device, not derived from the CU count: ADVICE round 4)
    if (ctx->nxcd <= 0) SC_TRY(probe_xcd_count(ctx, st));
    const int nxcd = spread ? 1 : ctx->nxcd;   // (spread: every workgroup counts as "XCD 0" and serves every matrix)
    if (want_chase && ctx->num_cus > 0 && (ctx->chase_ok != 0 || persist == 2)) {
      int per_cu = 0;
      if (pair) {
        const hipError_t oe = pair_loader ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bulge_pair<1>, 768, kPairLdsBytes)
                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bulge_pair<0>, 512, kPairLdsBytes);
        if (oe != hipSuccess) per_cu = 0;
      } else {
        const hipError_t oe = spread ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bulge_chase<1>, 256, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bulge_chase<0>, 256, 0);
        if (oe != hipSuccess) per_cu = 0;
      }
      const int slots_per_xcd = per_cu * ctx->num_cus / nxcd;
      const int mpx = (batch + nxcd - 1) / nxcd;   // matrices per XCD (XCD 0 has the most)
      // workgroups per matrix: all the XCD's slots divided by its matrices, but no more than sweeps can be in flight
      // (every sweep trails its predecessor by two tasks; a pair its predecessor pair by three steps)
      const int useful = pair ? std::max(4, chase_len(n, 0) / 3 + 2) : std::max(8, chase_len(n, 0) / 2 + 1);
      const int W = slots_per_xcd > 0 ? std::max(1, std::min(useful, slots_per_xcd / mpx)) : 0;
      if (W >= 1) {
        // progress counters (batch x n) | claim counters (batch) | ctl
        const size_t prog_bytes = align_up((size_t)batch * n * sizeof(int), 256);
        const size_t next_bytes = align_up((size_t)batch * sizeof(int), 256);
        const size_t ctl_bytes = align_up(kChaseCtlInts * sizeof(int), 256);
        // early hand-off records of k_bulge_chase (16 B x kEarlyRing per sweep; zeroed with the counters: a stale tag of
        // the previous solve would pass for this one's)
        static const bool no_early = getenv("SPRINGCRAFT_BULGE_NO_EARLY") != nullptr;
        const size_t early_bytes = (pair || no_early) ? 0 : align_up((size_t)batch * n * kEarlyRing * sizeof(v4i), 256);
        SC_TRY(sc_reserve_dc_aux(ctx, prog_bytes + next_bytes + ctl_bytes + early_bytes));
        int* d_prog = reinterpret_cast<int*>(ctx->dc_aux);
        int* d_next = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->dc_aux) + prog_bytes);
        int* d_ctl = reinterpret_cast<int*>(reinterpret_cast<char*>(ctx->dc_aux) + prog_bytes + next_bytes);
        v4i* d_early = early_bytes ? reinterpret_cast<v4i*>(reinterpret_cast<char*>(ctx->dc_aux) + prog_bytes + next_bytes + ctl_bytes)
                                   : nullptr;
        SC_HIP(ctx, hipMemsetAsync(ctx->dc_aux, 0, prog_bytes + next_bytes + ctl_bytes + early_bytes, st));
        // the dispatcher deals workgroups round-robin over the XCDs, so 8 x (workgroups one XCD needs) gives every XCD
        // its share; the kernel does not rely on it (a short-changed XCD is just slower, see the kernel's header)
        const int grid = nxcd * std::min(slots_per_xcd, mpx * W);
        t_bulge.start();
        if (pair && pair_loader)
          hipLaunchKernelGGL(k_bulge_pair<1>, dim3((unsigned)grid), dim3(768), kPairLdsBytes, st, d_sb_ws,
                             PairLayout{SL.n, SL.slab, SL.ab, SL.vd, SL.tau2}, batch, W, nxcd, d_prog, d_next, d_ctl,
                             ctx->chase_give_up, pair_early);
        else if (pair)
          hipLaunchKernelGGL(k_bulge_pair<0>, dim3((unsigned)grid), dim3(512), kPairLdsBytes, st, d_sb_ws,
                             PairLayout{SL.n, SL.slab, SL.ab, SL.vd, SL.tau2}, batch, W, nxcd, d_prog, d_next, d_ctl,
                             ctx->chase_give_up, pair_early);
        else
          if (spread)
            hipLaunchKernelGGL(k_bulge_chase<1>, dim3((unsigned)grid), dim3(256), 0, st, d_sb_ws, SL, batch, W, nxcd, d_prog,
                               d_next, d_ctl, ctx->chase_give_up, d_early);
          else
            hipLaunchKernelGGL(k_bulge_chase<0>, dim3((unsigned)grid), dim3(256), 0, st, d_sb_ws, SL, batch, W, nxcd, d_prog,
                               d_next, d_ctl, ctx->chase_give_up, d_early);
        const hipError_t le = hipGetLastError();
        t_bulge.stop();
        if (le != hipSuccess && pair) {
          // a refused pair launch (e.g. the 157 KB of dynamic LDS on a device that does not grant them): this solve is
          // finished by the per-wavefront launches below, later solves of the context leave the pair form alone
          ctx->pair_attr = 0;
          ++ctx->cnt_pair_fallbacks;
        }
        if (le == hipSuccess) {
          // (round 6: no look at the outcome from the host -- the take-over is a launch of its own that returns at once
          // when the chase completed, and the control block's counts reach the context's counters at the next
          // synchronising call: sc_collect_events)
          hipLaunchKernelGGL(k_chase_finish, dim3((unsigned)batch), dim3(256), 0, st, d_sb_ws, SL, batch, nxcd, d_prog,
                             (const int*)d_ctl, ctx->d_status);
          SC_HIP(ctx, hipGetLastError());
          ++ctx->cnt_chase_launches;
          if (pair) ++ctx->cnt_pair_launches;
          ctx->last_chase_ctl = d_ctl;
          chased = true;
        }
      }
    }
    if (!chased) ++ctx->cnt_stepwise_chases;
    static const int env_streams = [] { const char* e = getenv("SPRINGCRAFT_BULGE_STREAMS"); return e ? atoi(e) : 0; }();
    // (three parts from 48 matrices on: 477 -> 431 ms at 64; four and more exceed the ~11 us per launch one host thread
    // needs -- the stage then takes 48 k launches x 11.4 us --, and replaying the launches as a captured hipGraph is slower
    // still: 569 ms with three parts)
    const int nparts = std::max(1, std::min(env_streams > 0 ? env_streams : (batch >= 48 ? 3 : (batch >= 16 ? 2 : 1)),
                                            std::min(batch, 8)));
    if (!chased) t_bulge.start();
    if (chased) {
    } else if (nparts > 1) {
      SC_TRY(sc_aux_stream(ctx));            // (its fork event)
      SC_TRY(sc_side_streams(ctx, nparts - 1));
      SC_HIP(ctx, hipEventRecord(ctx->aux_fork, st));
      for (int p = 1; p < nparts; ++p) SC_HIP(ctx, hipStreamWaitEvent(ctx->side_streams[p - 1], ctx->aux_fork, 0));
      // launches go out in bursts of 32 wavefronts per stream rather than alternating streams launch by launch: the host
      // needs ~11-13 us per launch and three parts x 12 k launches are close to the stage's own time (451 -> 434 ms with
      // bursts of 16; round 3, same-box A/B on two boxes: 8 / 16 / 32 / 64 -> 465 / 464 / 421 / 463 and 437-467 / 430-434)
      static const int burst = [] { const char* e = getenv("SPRINGCRAFT_BULGE_BURST"); return e ? std::max(1, atoi(e)) : 32; }();
      for (int t0 = 0; t0 <= t_max; t0 += burst)
        for (int p = 0; p < nparts; ++p) {
          const int lo = (int)((long long)batch * p / nparts), hi = (int)((long long)batch * (p + 1) / nparts);
          for (int t = t0; t <= std::min(t_max, t0 + burst - 1); ++t)
            hipLaunchKernelGGL(k_bulge_step, dim3((unsigned)gx, (unsigned)(hi - lo)), dim3(256), 0,
                               p == 0 ? st : ctx->side_streams[p - 1], d_sb_ws + (size_t)lo * SL.slab, SL, t);
        }
      for (int p = 1; p < nparts; ++p) {
        SC_HIP(ctx, hipEventRecord(ctx->side_joins[p - 1], ctx->side_streams[p - 1]));
        SC_HIP(ctx, hipStreamWaitEvent(st, ctx->side_joins[p - 1], 0));
      }
    } else {
      for (int t = 0; t <= t_max; ++t)
        hipLaunchKernelGGL(k_bulge_step, dim3((unsigned)gx, (unsigned)batch), dim3(256), 0, st, d_sb_ws, SL, t);
    }
    if (!chased) t_bulge.stop();
  }
  hipLaunchKernelGGL(k_band_to_tri, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, st, d_sb_ws, SL,
                     d_tri_ws, TL);
  SC_HIP(ctx, hipGetLastError());
  if (prof) {
    SC_HIP(ctx, hipEventRecord(ev[2], st));
    SC_HIP(ctx, hipEventSynchronize(ev[2]));
    SC_HIP(ctx, hipEventElapsedTime(ms_stage1, ev[0], ev[1]));
    SC_HIP(ctx, hipEventElapsedTime(ms_stage2, ev[1], ev[2]));
  }
  t_qr.finish(); t_symm.finish(); t_syr2k.finish(); t_bulge.finish();
  return SC_OK;
}

// ================================================================================================================
// T factors and V T of all diamonds, on `st` (they only depend on the bulge chase, not on Z).
int bt2_prepare(sc_ctx* ctx, int n, int batch, double* d_sb_ws, const SbLayout& SL, hipStream_t st) {
  if (n < 3 || SL.ndia == 0) return SC_OK;
  for (auto& ph : ctx->phases)
    if (ph.first == "dia_tfactor") ph.second = 0.0;
  for (long long d0 = 0; d0 < SL.ndia; d0 += 32768) {
    const unsigned cnt = (unsigned)std::min<long long>(32768, SL.ndia - d0);
    hipLaunchKernelGGL(k_dia_tfactor2, dim3(cnt, (unsigned)batch), dim3(256), 0, st, d_sb_ws, SL, (int)d0);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// Z <- Q2 Z (after bt2_prepare).  d_z: (batch) ncols columns of length n (ld n).
int bt2_batched(sc_ctx* ctx, int n, int batch, double* d_sb_ws, const SbLayout& SL, const int* d_dia_off, double* d_z,
                long long stride_z, int ncols, float* ms_fused) {
  hipStream_t st = ctx->stream;
  if (n < 3 || SL.ndia == 0 || ncols <= 0) return SC_OK;
  ScopedEvents<2> ev;
  const bool prof = ctx->profiling && ms_fused;
  if (prof) {
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    SC_HIP(ctx, hipEventRecord(ev[0], st));
  }
  // few columns (partial spectrum): one launch per wavefront of independent diamonds, see k_bt2_wave
  static const int env_wave = [] { const char* e = getenv("SPRINGCRAFT_BT2_WAVE"); return e ? atoi(e) : -1; }();
  const long long col_wgs = (long long)((ncols + 63) / 64) * batch;
  const bool wave_path = env_wave >= 0 ? env_wave != 0 : col_wgs <= 16;
  if (wave_path) {
    int nk0 = 0;
    {
      const std::vector<int> doff = dia_offsets(n);
      for (size_t S = 0; S + 1 < doff.size(); ++S) nk0 = std::max(nk0, doff[S + 1] - doff[S]);
    }
    const int t_last = 3 * (SL.ngroups - 1) + nk0 - 1;
    for (int t = 0; t <= t_last; ++t) {
      const int gx = std::min(t / 3, SL.ngroups - 1) + 1;
      hipLaunchKernelGGL(k_bt2_wave, dim3((unsigned)gx, (unsigned)((ncols + 63) / 64), (unsigned)batch), dim3(256), 0, st,
                         d_sb_ws, SL, d_dia_off, d_z, stride_z, ncols, t);
    }
  } else {
    // the role-split form (k_bt2_role: 64 columns per workgroup, MFMA / fragment / window-row waves): SPRINGCRAFT_BT2_ROLE = 1
    static const int env_role = [] { const char* e = getenv("SPRINGCRAFT_BT2_ROLE"); return e ? atoi(e) : 0; }();
    if (env_role != 0 && (n & 1) == 0 && batch >= 8 && n >= 256) {
      // (per device, with the answer checked: sc_raise_dyn_lds)
      const bool role_attr = sc_raise_dyn_lds(reinterpret_cast<const void*>(&k_bt2_role), kRoleLds);
      if (role_attr) {
        const int nchunk64 = (ncols + 63) / 64;
        hipLaunchKernelGGL(k_bt2_role, dim3((unsigned)(8 * ((batch + 7) / 8) * nchunk64)), dim3(768), kRoleLds, st, d_sb_ws, SL,
                           d_dia_off, d_z, stride_z, ncols, batch);
        SC_HIP(ctx, hipGetLastError());
        if (prof) {
          SC_HIP(ctx, hipEventRecord(ev[1], st));
          SC_HIP(ctx, hipEventSynchronize(ev[1]));
          SC_HIP(ctx, hipEventElapsedTime(ms_fused, ev[0], ev[1]));
        }
        return SC_OK;
      }
    }
    // 128 columns per workgroup (8 waves) when that still gives every CU a workgroup, else 64 (4 waves)
    // ring of three half-diamond fragment buffers + one 16 x 18 transposition tile per wave
    constexpr size_t lds = sizeof(double) * (3 * kHalfDoubles + 8 * 16 * 18);
    // (per device, with the answer checked -- ADVICE round 5: a process-wide flag left a context on a second GPU with a
    // refused 138 KB launch)
    if (!sc_raise_dyn_lds(reinterpret_cast<const void*>(&k_bt2_apply<8>), (int)lds) ||
        !sc_raise_dyn_lds(reinterpret_cast<const void*>(&k_bt2_apply<4>), (int)lds))
      return sc_set_error(ctx, SC_ERR_HIP, "k_bt2_apply: the device refuses %zu bytes of dynamic LDS", lds);
    static const int force_nw = [] { const char* e = getenv("SPRINGCRAFT_BT2_NW"); return e ? atoi(e) : 0; }();
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    int nw = ((long long)((ncols + 127) / 128) * batch >= cus) ? 8 : 4;
    if (force_nw == 4 || force_nw == 8) nw = force_nw;
    const int nchunk = (ncols + 16 * nw - 1) / (16 * nw);
    const bool xcd = batch >= 8;
    const dim3 grid = xcd ? dim3((unsigned)(8 * ((batch + 7) / 8) * nchunk)) : dim3((unsigned)nchunk, (unsigned)batch);
    if (nw == 8)
      hipLaunchKernelGGL(k_bt2_apply<8>, grid, dim3(512), lds, st, d_sb_ws, SL, d_dia_off, d_z, stride_z, ncols, batch,
                         xcd ? 1 : 0);
    else
      hipLaunchKernelGGL(k_bt2_apply<4>, grid, dim3(256), lds, st, d_sb_ws, SL, d_dia_off, d_z, stride_z, ncols, batch,
                         xcd ? 1 : 0);
  }
  SC_HIP(ctx, hipGetLastError());
  if (prof) {
    SC_HIP(ctx, hipEventRecord(ev[1], st));
    SC_HIP(ctx, hipEventSynchronize(ev[1]));
    SC_HIP(ctx, hipEventElapsedTime(ms_fused, ev[0], ev[1]));
  }
  return SC_OK;
}

int sb_band_width() { return kB; }

// ---- diagnostic build only (-DBT2_TRACE): time stamps in front of every MFMA of one diamond, [wave][half][81]
extern "C" int sc_dbg_bt2_trace(unsigned long long* out) {
#ifdef BT2_TRACE
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bt2_trace), sizeof(unsigned long long) * 8 * 2 * 81) == hipSuccess ? 0 : 5;
#else
  (void)out;
  return 1;
#endif
}

// ---- diagnostic build only: per-wave segment sums of k_bt2_apply (first 64 workgroups x 8 waves x (8 sums + count))
extern "C" int sc_dbg_chase_stamps(unsigned long long* out16) {
#ifdef CHASE_STAMPS
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_chase_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 5;
  static const unsigned long long zeros[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_chase_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : 5;
#else
  (void)out16;
  return 1;
#endif
}

extern "C" int sc_dbg_bt2_clock(unsigned long long* out2) {
#ifdef BT2_CLOCK
  return hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_bt2_clk), sizeof(unsigned long long) * 2) == hipSuccess ? 0 : 5;
#else
  (void)out2;
  return 1;
#endif
}

extern "C" int sc_dbg_bt2_stamps(unsigned long long* out) {
#ifdef BT2_STAMPS
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bt2_stamps), sizeof(unsigned long long) * 64 * 8 * 17) == hipSuccess ? 0 : 5;
#else
  (void)out;
  return 1;
#endif
}

// ---- diagnostic build only: {sum t(E in LDS), sum t(E stored) [tasks with k > 0], sum t(D in LDS), sum t(end), tasks,
// tasks with k > 0}, cycles since the task's start; reset after the read
extern "C" int sc_dbg_pair_stamps(unsigned long long* out16) {
#ifdef PAIR_STAMPS
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pair_stamps), 384) != hipSuccess) return 5;   // (48 entries)
  const unsigned long long z[48] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stamps), z, 384) == hipSuccess ? 0 : 5;
#else
  (void)out16;
  return 1;
#endif
}

extern "C" int sc_dbg_bulge_stamps(unsigned long long* out6) {
#ifdef BULGE_STAMPS
  if (hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_bulge_stamps), 48) != hipSuccess) return 5;
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_bulge_stamps), z, 64) == hipSuccess ? 0 : 5;
#else
  (void)out6;
  return 1;
#endif
}

// ---- debugging entry point (not part of the public C ABI): band after stage 1 (128 x n, AB(i,j) at [(i-j) + 128 j])
// and the tridiagonal after stage 2 of ONE host matrix (NumPy layout, lower triangle read).
extern "C" int sc_dbg_two_stage(sc_ctx* ctx, const double* a, int n, double* band_out, double* d_out, double* e_out) {
  if (!ctx || !a || n < 4 * kB) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  TriLayout TL{};
  TL.n = n; TL.nb = kB;
  TL.d = 0; TL.e = n; TL.tau = 2 * (long long)n; TL.slab = 3 * (long long)n + 64;
  SbLayout SL;
  const size_t sbd = sb_slab_doubles(n, 1, &SL);
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
