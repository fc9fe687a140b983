// Batched Householder tridiagonalisation  A = Q_H T Q_H^T  of symmetric float64 matrices
// (lower triangle, column-major) — stage K5 of the eigensolver that replaces the LAPACK dsytrd
// step inside np.linalg.eigh (reference call site: nma.py:61).
//
// Blocked one-stage algorithm (panel of NB reflectors, trailing update deferred):
//   per column c      k_col_update : a = A[:,c] - V W[c,:]^T - W V[c,:]^T      (applies the panel lazily)
//                     k_symv_tiles : y = A22 v on 64x64 lower tiles, each tile read ONCE and used for
//                                    both  y_i += T x_j  and  y_j += T^T x_i  (HBM-bound: 8 B per lower element)
//                     k_w_reduce   : w~ = tau (y - V (W^T v) - W (V^T v)),  partial  w~^T v
//   per panel         k_w_fix + SYR2K:  A22 -= V W^T + W V^T  as ONE f64-MFMA GEMM with K = 2 NB
// The scalar  alpha2 = -1/2 tau (w~^T v)  needs a grid-wide reduction; it is applied lazily by the
// next k_col_update (w = w~ + alpha2 v), so a column costs three launches.
//
// All kernels take blockIdx.y = matrix index of the batch.
#include <algorithm>

#include "eigh_internal.h"
#include "host_logic.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return v;
}

// result valid in every thread
__device__ __forceinline__ double block_sum_bcast(double v, double* red /*[5]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

struct HH {
  double beta, tau, scale;
};

// LAPACK dlarfg scalars for alpha = a[c+1], xn2 = ||a[c+2:]||^2
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

// Sum of `cnt` partial results written by the previous kernel: coalesced loads, fixed reduction tree
// (deterministic), result in every thread.  Needs all 256 threads of the block; contains barriers.
__device__ __forceinline__ double sum_partials(const double* p, int cnt, double* sh /*[blockDim/64]*/) {
  double v = 0.0;
  for (int q = threadIdx.x; q < cnt; q += (int)blockDim.x) v += p[q];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  __syncthreads();
  return s;
}

// Sum over a workgroup of any size (a multiple of 64 threads, <= 1024): result in every thread; contains barriers.
__device__ __forceinline__ double block_sum_any(double v, double* sh /*[16]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;
}

// Rows are tiled in ABSOLUTE blocks of 64 (block B = rows 64B .. 64B+63), not relative to the
// current column: every 64-row segment then starts on an even row, so with an even matrix order the
// SYMV tiles can be fetched with 16-byte loads, and the partial-result tables keep the same indexing
// from one column to the next.  For column c the active rows are c+1 .. n-1, i.e. blocks
// b0 = (c+1)/64 .. nba-1 with nba = ceil(n/64); rows of block b0 below c+1 are masked.

__device__ __forceinline__ int first_block(int row) { return row >> 6; }

// ---- column update -------------------------------------------------------------------------------------
// a = A[:,c] - V W[c,:]^T - W V[c,:]^T for rows c .. n-1 (i = c - j0 pending reflectors).
// Block = 64 absolute rows x 4 groups that split the reflector range; group sums meet in LDS.
__global__ __launch_bounds__(1024) void k_col_update(double* __restrict__ a_all, long long stride_a,
                                                     double* __restrict__ ws_all, TriLayout L, int c,
                                                     int j0) {
  constexpr int NG = 16;
  __shared__ double rowV[128], rowW[128], part[NG][64], sh[16], arow[64];
  const int n = L.n, nb = L.nb, i = c - j0;
  const int nba = (n + 63) >> 6;
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  double* VW = ws + L.vw;
  double* WV = ws + L.wv;
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const int r = ((c >> 6) + blockIdx.x) * 64 + lane;
  const bool valid = r >= c && r < n;

  double alpha2 = 0.0;
  if (i > 0) {  // block-uniform
    const double tau_prev = ws[L.tau + c - 1];
    const int cnt = nba - first_block(c);  // blocks of k_w_reduce(c-1): rows c .. n-1
    const double s = sum_partials(ws + L.wvpart, cnt, sh);
    if (tau_prev != 0.0) alpha2 = -0.5 * tau_prev * s;
  }
  if (tid < i) {
    const int p = tid;
    const double vcp = VW[(size_t)p * n + c];
    double wcp = VW[(size_t)(nb + p) * n + c];
    if (p == i - 1) wcp += alpha2 * vcp;
    rowV[p] = vcp;
    rowW[p] = wcp;
  }
  __syncthreads();

  // each group owns reflectors p = g, g+16, ..., g+112 (nb <= 128): keep their V / W entries for the dots below
  constexpr int NU = 8;
  double vk[NU], wk[NU];
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int p = g + NG * u;
    vk[u] = 0.0;
    wk[u] = 0.0;
    if (valid && p < i) {
      const double v = VW[(size_t)p * n + r];
      double w = VW[(size_t)(nb + p) * n + r];
      if (p == i - 1 && alpha2 != 0.0) {
        w += alpha2 * v;
        // Row c itself is never read again (later columns and the SYR2K only touch rows > c), and every
        // block reads the UNFIXED W[c, i-1] into rowW above: do not store it (would race with those reads).
        if (r > c) {
          VW[(size_t)(nb + p) * n + r] = w;   // W column of [V|W]
          WV[(size_t)p * n + r] = w;          // W column of [W|V]
        }
      }
      acc += v * rowW[p] + w * rowV[p];
      vk[u] = v;
      wk[u] = w;
    }
  }
  part[g][lane] = acc;
  __syncthreads();
  if (g == 0) {
    double sq = 0.0, a = 0.0;
    if (valid) {
      double corr = 0.0;
#pragma unroll
      for (int q = 0; q < NG; ++q) corr += part[q][lane];
      a = A[(size_t)c * n + r] - corr;
      ws[L.xraw + r] = a;
      if (r == c) ws[L.d + c] = a;
      if (r == c + 1 && c == n - 2) {  // last sub-diagonal element: no reflector (dsytd2: tau = 0)
        ws[L.e + c] = a;
        ws[L.tau + c] = 0.0;
      }
      if (r >= c + 2) sq = a * a;
    }
    arow[lane] = (valid && r >= c + 2) ? a : 0.0;   // rows below the reflector's unit entry
    sq = wave_sum(sq);
    if (lane == 0) ws[L.npart + blockIdx.x] = sq;
  }
  __syncthreads();
  // Partial dots for the panel corrections of THIS column's reflector v = (1, scale * a[c+2:]):
  //   V_p^T v = scale * sum_{r >= c+2} V[r,p] a[r] + V[c+1,p]   (same for W); the sums are taken here, where the
  // V / W entries are already in registers; k_w_reduce adds the scale and the unit-entry term.
  const double ar = arow[lane];
  const int blk = (c >> 6) + blockIdx.x;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int p = g + NG * u;
    if (p < i) {  // wave-uniform
      const double sv = wave_sum(vk[u] * ar);
      const double sw = wave_sum(wk[u] * ar);
      if (lane == 0) {
        ws[L.dpart + (size_t)blk * 2 * nb + p] = sv;
        ws[L.dpart + (size_t)blk * 2 * nb + nb + p] = sw;
      }
    }
  }
}

// ---- symmetric matrix-vector product on lower 64x64 tiles ---------------------------------------------------
// Tiles (bi >= bj) over the absolute row blocks b0 .. nba-1 of the trailing matrix; ntiles = nt (nt+1)/2.
// Persistent blocks: block g walks tiles g, g+G, g+2G, ...; as soon as the current tile has been parked in LDS
// the NEXT tile's 16 doubles per thread (and its x segments) are requested into the same registers and stay in
// flight while the current tile is multiplied, so every resident block always has 32 KB of HBM reads outstanding.  Barriers are raw s_barrier + lgkmcnt(0): they only order
// LDS traffic and must not drain the prefetch (a __syncthreads() would wait for vmcnt(0)).
// Each tile is fetched once and used for both  y_i += T x_j  and  y_j += T^T x_i.
// VEC2: 16-byte loads (needs an even matrix order).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void tile_coords(int t, int& ti, int& tj) {
  ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (ti * (ti + 1) / 2 > t) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  tj = t - ti * (ti + 1) / 2;
}

template <bool VEC2>
__global__ __launch_bounds__(256, 4) void k_symv_tiles(double* __restrict__ a_all, long long stride_a,
                                                    double* __restrict__ ws_all, TriLayout L, int c,
                                                    int j0, int ntiles) {
  constexpr int TS = 64, LD = 66;
  __shared__ __attribute__((aligned(16))) double T[TS * LD];
  __shared__ double xi[TS], xj[TS];
  __shared__ double red[4][TS];
  __shared__ double sh[4];
  const int n = L.n, nb = L.nb, i = c - j0;
  const int nba = (n + 63) >> 6, b0 = first_block(c + 1);
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  double* VW = ws + L.vw;
  double* WV = ws + L.wv;
  const int tid = threadIdx.x;
  const int first = c + 1;
  const int lr = tid & 31, kq = tid >> 5;
  const int lane = tid & 63, q = tid >> 6;

  // loads of one tile: thread (lr, kq) owns rows r0+2lr, r0+2lr+1 of columns k0 + kq + 8p; threads < 128 also
  // fetch one raw x entry each (xi for tid < 64, xj for 64 <= tid < 128)
  // All loads are UNCONDITIONAL (addresses clamped into the matrix): a predicated load merged with a zero
  // makes hipcc wait for each load right after issuing it.  Masking happens when the tile goes to LDS.
  auto issue = [&](int t, double (&v0)[8], double (&v1)[8], double& xr) {
    int ti, tj;
    tile_coords(t, ti, tj);
    const int r0 = (b0 + ti) * TS, k0 = (b0 + tj) * TS;
    const int row = min(r0 + 2 * lr, VEC2 ? n - 2 : n - 1);
    const double* base = A + row;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int col = min(k0 + kq + 8 * p, n - 1);
      const double* ptr = base + (size_t)col * n;
      if (VEC2) {
        const double2 d2 = *reinterpret_cast<const double2*>(ptr);
        v0[p] = d2.x;
        v1[p] = d2.y;
      } else {
        v0[p] = ptr[0];
        v1[p] = ptr[row + 1 < n ? 1 : 0];
      }
    }
    const int rr = min(((tid & 64) ? k0 : r0) + (tid & 63), n - 1);
    xr = ws[L.xraw + rr];
  };

  double c0[8], c1[8], cx;
  int t = blockIdx.x;
  issue(t, c0, c1, cx);

  // Householder scalars (every block recomputes them from the same partials, in the same order)
  const double xn2 = sum_partials(ws + L.npart, nba - first_block(c), sh);
  const double alpha = ws[L.xraw + c + 1];
  const HH h = householder(alpha, xn2);

  while (t < ntiles) {
    int ti, tj;
    tile_coords(t, ti, tj);
    const int bi = b0 + ti, bj = b0 + tj;
    const bool diag = (bi == bj);
    const int r0 = bi * TS, k0 = bj * TS;
    const int row = r0 + 2 * lr;

    // current tile: registers -> LDS (the only place that waits for its loads)
    if (tid < 2 * TS) {
      const int rr = (tid < TS ? r0 : k0) + (tid & 63);
      const double xv = (rr >= first && rr < n) ? (rr == first ? 1.0 : cx * h.scale) : 0.0;
      if (tid < TS) xi[tid] = xv; else xj[tid - TS] = xv;
    }
    {
      const bool r0ok = row >= first && row < n, r1ok = row + 1 >= first && row + 1 < n;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int kk = kq + 8 * p;
        const int col = k0 + kk;
        const bool cok = col >= first && col < n;
        double2 d2;
        d2.x = (r0ok && cok) ? c0[p] : 0.0;
        d2.y = (r1ok && cok) ? c1[p] : 0.0;
        *reinterpret_cast<double2*>(&T[2 * lr + kk * LD]) = d2;
      }
    }
    // next tile: its loads go into the registers just freed and fly during the multiplications below
    // (always issued, tile index clamped, so no predicated-load merge; hipcc waits for them at the top of
    // the next iteration, where nothing else is outstanding)
    const int tn = t + (int)gridDim.x;
    issue(min(tn, ntiles - 1), c0, c1, cx);
    lds_barrier();

    if (diag) {
      // y[r] = sum_k Tsym[r,k] x[k], Tsym from the lower triangle only
      double acc = 0.0;
#pragma unroll 4
      for (int kk = q * 16; kk < q * 16 + 16; ++kk) {
        const double tv = (lane >= kk) ? T[lane + kk * LD] : T[kk + lane * LD];
        acc += tv * xj[kk];
      }
      red[q][lane] = acc;
      lds_barrier();
      if (tid < TS && r0 + tid < n)
        ws[L.ypart + (size_t)bj * n + r0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    } else {
      double acc = 0.0;
#pragma unroll 4
      for (int kk = q * 16; kk < q * 16 + 16; ++kk) acc += T[lane + kk * LD] * xj[kk];
      red[q][lane] = acc;
      lds_barrier();
      if (tid < TS && r0 + tid < n)
        ws[L.ypart + (size_t)bj * n + r0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      lds_barrier();
      // transposed product: y_j[k] += sum_r T[r,k] x_i[r]
      acc = 0.0;
#pragma unroll 4
      for (int rr = q * 16; rr < q * 16 + 16; ++rr) acc += T[rr + lane * LD] * xi[rr];
      red[q][lane] = acc;
      lds_barrier();
      if (tid < TS && k0 + tid < n)
        ws[L.ypart + (size_t)bi * n + k0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }

    if (diag) {
      // store v (explicit leading 1) for the back-transformation and into both panel copies
      const int arow = r0 + lane;
      const bool rv = arow >= first && arow < n;
      if (tid < TS && rv) {
        const double xv = xi[tid];
        A[(size_t)c * n + arow] = xv;
        VW[(size_t)i * n + arow] = xv;
        WV[(size_t)(nb + i) * n + arow] = xv;
      }
      if (t == 0 && tid == 0) {
        ws[L.e + c] = h.beta;
        ws[L.tau + c] = h.tau;
        ws[L.hscale] = h.scale;
      }
    }
    lds_barrier();  // T / x / red are rewritten by the next tile
    t = tn;
  }
}

// ---- reduce partial products, apply panel corrections, scale by tau ---------------------------------------
// Block = 64 absolute rows x 4 groups; the groups split the partner-block range (y) and the reflector
// range (corrections) and meet in LDS, so the launch has ~n/64 blocks instead of n/256.
__global__ __launch_bounds__(1024) void k_w_reduce(double* __restrict__ a_all, long long stride_a,
                                                   double* __restrict__ ws_all, TriLayout L, int c,
                                                   int j0) {
  constexpr int NG = 16;
  __shared__ double dots[4][256], part[NG][64];
  const int n = L.n, nb = L.nb, i = c - j0;
  const int nba = (n + 63) >> 6, b0 = first_block(c + 1);
  const double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  double* VW = ws + L.vw;
  double* WV = ws + L.wv;
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  {
    const int col = tid & 255, slice = tid >> 8;   // 4 slices of the block range per dot column (2 nb <= 256)
    const int p = col < nb ? col : col - nb;
    double s = 0.0;
    if (col < 2 * nb && p < i)
      for (int b = first_block(c) + slice; b < nba; b += 4) s += ws[L.dpart + (size_t)b * 2 * nb + col];
    dots[slice][col] = s;  // partial sums over rows >= c+2 of V[r,p] a[r] / W[r,p] a[r] (k_col_update)
  }
  __syncthreads();
  if (tid < 2 * nb) {
    const int p = tid < nb ? tid : tid - nb;
    const double s = (dots[0][tid] + dots[1][tid]) + (dots[2][tid] + dots[3][tid]);
    // v = (1, scale * a[c+2:]):  [p] = V_p^T v, [nb+p] = W_p^T v
    dots[0][tid] = p < i ? ws[L.hscale] * s + VW[(size_t)tid * n + c + 1] : 0.0;
  }
  __syncthreads();

  const int r = (b0 + blockIdx.x) * 64 + lane;
  const bool valid = r >= c + 1 && r < n;
  double acc = 0.0;
  if (valid) {
    for (int o = b0 + g; o < nba; o += NG) acc += ws[L.ypart + (size_t)o * n + r];
    for (int p = g; p < i; p += NG) {
      const double v = VW[(size_t)p * n + r];
      const double w = VW[(size_t)(nb + p) * n + r];
      acc -= v * dots[0][nb + p] + w * dots[0][p];
    }
  }
  part[g][lane] = acc;
  __syncthreads();
  if (g == 0) {
    double wv = 0.0;
    if (valid) {
      double y = 0.0;
#pragma unroll
      for (int q = 0; q < NG; ++q) y += part[q][lane];
      const double wt = ws[L.tau + c] * y;
      VW[(size_t)(nb + i) * n + r] = wt;
      WV[(size_t)i * n + r] = wt;
      wv = wt * A[(size_t)c * n + r];  // v was stored in A[:, c] by k_symv_tiles
    }
    wv = wave_sum(wv);
    if (lane == 0) ws[L.wvpart + blockIdx.x] = wv;
  }
}

// ---- end of panel: apply the pending alpha2 of the panel's last reflector --------------------------------------
__global__ __launch_bounds__(256) void k_w_fix(double* __restrict__ ws_all, TriLayout L, int c, int j0) {
  const int n = L.n, nb = L.nb, i = c - j0;
  const int nba = (n + 63) >> 6;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  double* VW = ws + L.vw;
  double* WV = ws + L.wv;
  __shared__ double sh[4];
  const double tau = ws[L.tau + c];
  if (tau == 0.0) return;  // block-uniform
  const double alpha2 = -0.5 * tau * sum_partials(ws + L.wvpart, nba - first_block(c + 1), sh);
  const int r = c + 1 + blockIdx.x * 256 + threadIdx.x;
  if (r < n) {
    const double w = VW[(size_t)(nb + i) * n + r] + alpha2 * VW[(size_t)i * n + r];
    VW[(size_t)(nb + i) * n + r] = w;
    WV[(size_t)i * n + r] = w;
  }
}

__global__ void k_zero_panels(double* __restrict__ ws_all, TriLayout L) {
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  const size_t total = (size_t)L.n * 2 * L.nb;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    ws[L.vw + idx] = 0.0;
    ws[L.wv + idx] = 0.0;
  }
}

// X(r,c) <- X(c,r) for r > c: makes the column-major lower triangle equal to the lower triangle of the
// row-major (NumPy) matrix the caller handed over, i.e. exactly what eigh(UPLO='L') reads (nma.py:61).
__global__ void k_mirror_lower(double* __restrict__ a_all, long long stride_a, int n) {
  __shared__ double tile[32][33];
  double* A = a_all + (size_t)blockIdx.z * stride_a;
  const int bx = blockIdx.x, by = blockIdx.y;  // tile (row-block by, col-block bx) of the destination, by >= bx
  if (by < bx) return;
  const int tx = threadIdx.x, ty = threadIdx.y;
  // read source tile (rows of block bx, cols of block by) = upper part, coalesced along rows
  for (int k = ty; k < 32; k += 8) {
    const int sr = bx * 32 + tx, sc = by * 32 + k;
    tile[k][tx] = (sr < n && sc < n) ? A[(size_t)sc * n + sr] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int dr = by * 32 + tx, dc = bx * 32 + k;
    if (dr < n && dc < n && dr > dc) A[(size_t)dc * n + dr] = tile[tx][k];
  }
}

// ---- scaling of badly scaled matrices (what LAPACK dsyevd does with dlansy / dlascl) -------------------------------
// The Householder norms square the entries, so a matrix whose largest |entry| is outside [1e-100, 1e100] is multiplied
// by a power of two that brings it to ~1 (exact), and the eigenvalues are divided by it afterwards.  Everything is
// decided on the device: amax -> factor -> conditional in-place scaling, no host synchronisation.
__global__ __launch_bounds__(256) void k_absmax_lower(const double* __restrict__ a_all, long long stride_a, int n,
                                                      double* __restrict__ ws_all, TriLayout L,
                                                      unsigned long long* __restrict__ status) {
  const double* A = a_all + (size_t)blockIdx.y * stride_a;
  unsigned long long* slot = reinterpret_cast<unsigned long long*>(ws_all + (size_t)blockIdx.y * L.slab + L.hscale + 1);
  double m = 0.0;
  int bad = 0;
  const size_t total = (size_t)n * n;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx / n), r = (int)(idx - (size_t)c * n);
    if (r >= c) {
      const double x = fabs(A[idx]);
      m = fmax(m, x);                       // (fmax drops NaNs)
      bad |= !(x <= 1.7976931348623157e308);   // NaN or Inf
    }
  }
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  // the bit patterns of non-negative doubles are ordered like the values
  if ((threadIdx.x & 63) == 0 && m > 0.0) atomicMax(slot, (unsigned long long)__double_as_longlong(m));
  // a matrix with a NaN / Inf entry must not reach the solver (comparisons with NaN would drive its index arithmetic):
  // a flag in its own slab makes k_scale_lower replace it by the zero matrix and k_unscale_values return NaN for its
  // eigenvalues; its number + 1 goes to the context's deferred status word (sc_deferred_status)
  if (__any(bad) && (threadIdx.x & 63) == 0) {
    *reinterpret_cast<unsigned long long*>(ws_all + (size_t)blockIdx.y * L.slab + L.hscale + 3) = 1ull;
    atomicMax(status, (unsigned long long)blockIdx.y + 1ull);
  }
}

__device__ __forceinline__ double matrix_scale_factor(double amax) {
  if (!(amax > 0.0) || amax > 1.7e308) return 1.0;
  if (amax >= 1e-100 && amax <= 1e100) return 1.0;
  return ldexp(1.0, -ilogb(amax));
}

__global__ __launch_bounds__(256) void k_scale_lower(double* __restrict__ a_all, long long stride_a, int n,
                                                     double* __restrict__ ws_all, TriLayout L) {
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  const double amax = __longlong_as_double((long long)*reinterpret_cast<const unsigned long long*>(ws + L.hscale + 1));
  const bool bad = *reinterpret_cast<const unsigned long long*>(ws + L.hscale + 3) != 0ull;
  const double f = bad ? 0.0 : matrix_scale_factor(amax);
  if (blockIdx.x == 0 && threadIdx.x == 0) ws[L.hscale + 2] = bad ? 1.0 : f;
  if (f == 1.0) return;
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  const size_t total = (size_t)n * n;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx / n), r = (int)(idx - (size_t)c * n);
    // (r | 1: also the first super-diagonal entry of every even row, which the band reduction keeps equal to its mirror
    // image for k_symm3 -- symm3.hip)
    if ((r | 1) >= c) A[idx] = bad ? 0.0 : A[idx] * f;   // (a matrix with non-finite entries is solved as the zero matrix)
  }
}

__global__ __launch_bounds__(256) void k_unscale_values(double* __restrict__ w_all, long long stride_w, int m,
                                                        const double* __restrict__ ws_all, TriLayout L) {
  const double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  const double f = ws[L.hscale + 2];
  const bool bad = *reinterpret_cast<const unsigned long long*>(ws + L.hscale + 3) != 0ull;
  if (f == 1.0 && !bad) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) {
    double& w = w_all[(size_t)blockIdx.y * stride_w + i];
    w = bad ? __builtin_nan("") : w / f;
  }
}


// ---- one matrix RESIDENT on the chip: the whole tridiagonalisation in one launch (round 6) -------------------------------
// The reference's own call is ONE structure at a time (anm.py:150-167, proteins of 100 - 1500 residues), and for one
// matrix the launches above are all latency: 3 dependent launches of 5 - 9 us per column, 25 ms at n = 1536, whatever
// the kernels inside them do.  Here P <= 256 workgroups of ONE launch keep the matrix on the chip for the whole
// reduction -- workgroup k holds the FULL rows k, k + P, k + 2P, ... (both triangles): in LDS for orders <= 2048 (<= 8
// rows of <= 2048 doubles: <= 128 KB), in registers for orders <= 3072 (<= 12 rows, thread t the entries t, t + 256, ...
// of each) -- and run LAPACK's unblocked dsytd2 recurrence with ONE exchange between the workgroups per column:
//   * every workgroup knows the current reflector v_c (registers: thread t holds the entries j = t, t + 256, ...) and
//     has y_r = A[r, :] v_c for its own rows from the pass below;
//   * it publishes, for each own row r, the 16-byte record { tau y_r , A[r, c+1] } (sc1 store: agent scope, visible to
//     the other XCDs without a fence) and polls the records of ALL rows (sc1 loads; a record is "empty" while it holds
//     the all-ones pattern, which no finite or NaN result of an arithmetic instruction has);
//   * from the records EVERY workgroup computes, redundantly and identically, w = w~ - (tau/2)(w~.v) v, the next column
//     a = A[:, c+1] - v w[c+1] - w (by symmetry A[r, c+1] of row r's owner IS the column entry; v[c+1] = 1), its norm,
//     the next reflector and tau: two workgroup-level reductions, no second exchange;
//   * one pass over the own rows applies the rank-2 update A -= v w^T + w v^T and multiplies the updated rows with the
//     NEXT reflector in the same sweep (the y of the next column); a thread only ever touches its own entries of a row.
// Three record buffers take turns: a row's owner empties the buffer of step c - 1 after its own poll of step c (by
// then every workgroup has published step c, i.e. has finished reading step c - 1) and waits for that store before it
// publishes step c + 1, so a reader of step c + 2 cannot see a record of step c - 1.
// Workgroup (c mod P) stores column c's reflector / d / e / tau.  The two copies of an off-diagonal element are updated
// with the operands in a different order and may differ in the last bit: the reduction then is that of a matrix
// A + E with |E| of the order of the rounding errors of the update itself.
// All P workgroups must be resident together.  They check it BEFORE anything is stored (roll call: workgroup 0 counts
// the arrivals within a bound of about 2 s and publishes go / abort in one word that a late comer can only read): after
// an abort the matrix is untouched and k_sytrd_takeover, enqueued behind every launch, reduces it by one workgroup from
// memory; it also restarts a reduction that lost a wait in mid-run when the upper triangle still holds the matrix
// (whole-matrix launches; a trailing-matrix launch reports the failed solve through the status word).  The launches of a
// device are chained by an event (resident_chain below): two of them side by side would each hold half of the chip.
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int kResR = 8;                 // rows per workgroup at most
// (orders: <= 2048 with the rows in LDS, 8 rows of 8 x 256 entries; <= 3072 with the rows in registers, 12 rows of 12 x 256
// entries, 256 workgroups; the launch shape is decided in host_logic.h: resident_shape)
constexpr int kResMaxM = sc_host::kResidentMaxReg;
static_assert(kResR == sc_host::kResidentRowsLds, "rows per workgroup");

struct ResArgs {
  double* a;              // whole matrices, column-major, leading dimension L.n
  long long stride_a;
  double* tri;
  TriLayout L;
  int off, m;             // the trailing matrix reduced here: rows / columns off .. off + m - 1
  int P, logP;            // workgroups per matrix (a power of two)
  int full;               // both triangles of the trailing matrix hold it (off = 0: straight after prepare_matrix_batched)
  int hook;               // tests: 1 = the roll call fails, 2 + c = the exchange of step c fails
  int delay;              // pauses of 64 cycles between a step's publication and its first poll
  unsigned long long* status;
};

__device__ __forceinline__ double res_pair_raw(int lo, int hi) {
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
template <int CTRL>
__device__ __forceinline__ double res_dpp(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
// eight sums over the 64 lanes at the cost of three (as twostage.hip's wave_reduce8): lane l < 8 returns the total of
// v[res_reduce8_index(l)]
__device__ __forceinline__ int res_reduce8_index(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }
__device__ __forceinline__ double res_reduce8(const double (&v)[8]) {
  const int lane = threadIdx.x & 63;
  double k4[4], k2[2];
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double send = b0 ? v[k] : v[k + 4];
    k4[k] = (b0 ? v[k + 4] : v[k]) + res_dpp<0xB1>(send);       // lane ^ 1
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double send = b1 ? k4[k] : k4[k + 2];
    k2[k] = (b1 ? k4[k + 2] : k4[k]) + res_dpp<0x4E>(send);     // lane ^ 2
  }
  const double send = b2 ? k2[0] : k2[1];
  const double from_lo = res_dpp<0x114>(send), from_hi = res_dpp<0x104>(send);   // lane - 4, lane + 4
  double r = (b2 ? k2[1] : k2[0]) + (b2 ? from_lo : from_hi);                    // lane ^ 4
  r += res_dpp<0x128>(r);                                                        // lane ^ 8 inside a row of 16
  r += __shfl_xor(r, 16);
  r += __shfl_xor(r, 32);
  return r;
}
// sum over the 64 lanes on the vector ALU alone (row_shr 1 2 4 8 inside the rows of 16, row_bcast 15 / 31 across them):
// the total is in LANE 63 only.  (wave_sum above goes through the LDS crossbar: twelve ds_bpermute in a row.)
__device__ __forceinline__ double res_wave_sum63(double x) {
  x += res_dpp<0x111>(x);
  x += res_dpp<0x112>(x);
  x += res_dpp<0x114>(x);
  x += res_dpp<0x118>(x);
  {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x142, 0xa, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x142, 0xa, 0xf, false);
    x += res_pair_raw(lo, hi);
  }
  {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x143, 0xc, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x143, 0xc, 0xf, false);
    x += res_pair_raw(lo, hi);
  }
  return x;
}
__device__ __forceinline__ double res_pair(int lo, int hi) {
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ void res_store(v4i* p, double x, double y) {
  const unsigned long long bx = (unsigned long long)__double_as_longlong(x), by = (unsigned long long)__double_as_longlong(y);
  const v4i r = {(int)(unsigned)bx, (int)(unsigned)(bx >> 32), (int)(unsigned)by, (int)(unsigned)(by >> 32)};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
}

// The pass of a step over NR own rows (the live ones, consecutive in LDS from `row`): A -= v w^T + w v^T and the products
// with the next reflector, chunk of 256 columns by chunk.  No test per element -- v, w and vn are zero at the columns that
// are done (but for column c + 1, whose entries nobody reads again) and in the padding of the rows --, so the NR reads of a
// chunk are in flight together: with a test per element every read waited for the LDS on its own, 70 cycles per element.
template <int Q, int NR>
__device__ __forceinline__ void res_pass(double* row, const double* vr_l, const double* wr_l, int q0, int qn,
                                         const double (&v)[Q], const double (&w)[Q], const double (&vn)[Q], double (&acc)[8]) {
  constexpr int LDr = 256 * Q;
  double vr[NR], wr[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    vr[u] = vr_l[u];
    wr[u] = wr_l[u];
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if (q >= q0 && q < qn) {   // (uniform)
      double x[NR];
#pragma unroll
      for (int u = 0; u < NR; ++u) x[u] = row[u * LDr + 256 * q];
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        x[u] -= vr[u] * w[q] + wr[u] * v[q];
        row[u * LDr + 256 * q] = x[u];
        acc[u] += x[u] * vn[q];
      }
    }
  }
}

// Diagnostic build (-DRES_STAMPS): cycles (s_memtime) that thread 0 of workgroup 0 spends between six points of a step,
// summed over the steps of a launch and added to g_res_stamps (sc_dbg_resident_stamps, tools/resident_check.py --stamps);
// nothing of it in the normal build.
#ifdef RES_STAMPS
__device__ unsigned long long g_res_stamps[8];
#define RES_STAMP(i)                                                                        \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    st_acc[i] += t_ - st_last;                                                              \
    st_last = t_;                                                                           \
  } while (0)
#else
#define RES_STAMP(i)
#endif

// RR = 0: the own rows in LDS (orders up to 2048).  RR > 0: in REGISTERS, RR rows at most (orders up to 3072 with P = 256:
// thread t keeps the entries t, t + 256, ... of all own rows, <= 12 x 12 doubles of the 512 registers a thread of a
// 256-thread workgroup may have; a pass touches only the thread's own entries, so LDS was never more than storage).
template <int Q, int RR>
__global__ __launch_bounds__(256) void k_sytrd_resident(ResArgs g) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int m = g.m, lp = g.logP, P = g.P, n = g.L.n;
  const int k = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int LDr = 256 * Q;                              // row stride in LDS
  const int RL = (m + 63) & ~63;                            // records of one buffer
  const int Qn = (m + 255) >> 8;                            // chunks of 256 columns with entries
  const int R = k < m ? ((m - 1 - k) >> lp) + 1 : 0;        // own rows k, k + P, ...
  const int Rmax = ((m - 1) >> lp) + 1;
  double* Aw = g.a + (size_t)blockIdx.y * g.stride_a;
  double* A = Aw + (size_t)g.off * n + g.off;               // the trailing matrix (leading dimension n)
  double* tri = g.tri + (size_t)blockIdx.y * g.L.slab;
  int* ctl = reinterpret_cast<int*>(tri + g.L.rctl);        // [0] -1 undecided / 1 go / 2 abort, [1] arrivals - 1, [2] 1 = a wait was lost
  v4i* rec = reinterpret_cast<v4i*>(tri + g.L.rrec);
  double* rows = sm;                                         // [Rmax][LDr]  (RR = 0)
  double* red = RR > 0 ? sm : rows + (size_t)Rmax * LDr;     // [4] [4] : partial sums of the two reductions
  double* red8 = red + 8;                                    // [4][16]
  double* bc = red8 + 64;                                    // [4]   values every thread needs
  double* vrow = bc + 4;                                     // [16]  v_c, w_c at the own rows (RR > 0: pairs {v, w}, [16][2])
  double* wrow = vrow + 16;
  double* colbuf = wrow + 16;                                // [16]  RR > 0: the own rows' entries of the next column
  int* s_flag = reinterpret_cast<int*>(colbuf + 16);         // [0] go, [1] dead
  double xr[RR > 0 ? RR : 1][Q];                             // RR > 0: the own rows

  // ---- the own rows -> LDS.  Straight after prepare_matrix_batched both triangles hold the matrix (the mirror pass
  // copies, it does not move), so a row is contiguous in memory; the scaling pass touches the lower triangle only, and a
  // trailing matrix behind SYR2K updates lives in the lower triangle only: the left part of a row is then gathered.
  {
    const double f = tri[g.L.hscale + 2];
    const bool bad = *reinterpret_cast<const unsigned long long*>(tri + g.L.hscale + 3) != 0ull;
    const bool full = g.full && f == 1.0 && !bad;
    if constexpr (RR > 0) {
#pragma unroll
      for (int i = 0; i < RR; ++i) {
        const int r = k + (i << lp);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int j = tid + 256 * q;
          double x = 0.0;
          if (i < R && j < m) x = (full || j >= r) ? A[(size_t)r * n + j] : A[(size_t)j * n + r];
          xr[i][q] = x;
        }
      }
    } else {
      for (int i = 0; i < R; ++i) {
        const int r = k + (i << lp);
        for (int j = tid; j < LDr; j += 256) {
          double x = 0.0;
          if (j < m) x = (full || j >= r) ? A[(size_t)r * n + j] : A[(size_t)j * n + r];
          rows[(size_t)i * LDr + j] = x;
        }
      }
    }
  }
  double v[Q], w[Q], a[Q], vn[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int j = tid + 256 * q;
    v[q] = 0.0;
    w[q] = 0.0;
    a[q] = j < m ? A[j] : 0.0;          // column 0 (lower triangle)
  }
  // ---- roll call (behind every read of the matrix: workgroup 0 stores column 0's reflector before the first exchange)
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(&ctl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int dec = -1;
    long spins = 0;
    if (k == 0) {
      bool all = false;
      // (bound: ~2 s.  0.12 s was too tight: once in a few thousand launches some workgroups of a launch are dispatched that
      // late on an otherwise idle GPU -- a single-stream stress run and one run of the suite each lost one roll call to it)
      while (!all && spins < (1L << 21)) {
        all = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P - 1;
        ++spins;
        if (!all) __builtin_amdgcn_s_sleep(8);
      }
      if (g.hook == 1) all = false;
      const int want = all ? 1 : 2;
      const int old = atomicCAS(&ctl[0], -1, want);
      dec = old == -1 ? want : old;
    } else {
      while ((dec = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == -1) {
        if (++spins > (1L << 23)) {
          const int old = atomicCAS(&ctl[0], -1, 2);
          dec = old == -1 ? 2 : old;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    s_flag[0] = dec;
    s_flag[1] = 0;
  }
  __syncthreads();
  if (s_flag[0] != 1) return;

  double tau = 0.0, yown = 0.0;
  const int myr = k + (tid << lp);      // the row thread tid < R publishes
  unsigned ownmask = 0;                 // bit q: entry tid + 256 q of a vector belongs to an own row
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int dj = tid + 256 * q - k;
    if (dj >= 0 && (dj & (P - 1)) == 0 && tid + 256 * q < m) ownmask |= 1u << q;
  }
  double* const d_out = tri + g.L.d + g.off;
  double* const e_out = tri + g.L.e + g.off;
  double* const tau_out = tri + g.L.tau + g.off;

#ifdef RES_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
  for (int c = -1; c <= m - 3; ++c) {
    RES_STAMP(5);                                                // [5] y of the own rows summed
    if (c >= 0) {
      v4i* rb = rec + (size_t)(c % 3) * RL;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the emptying store of the previous step, see above)
      if (g.hook == 2 + c && k == 0 && tid == 0) atomicExch(&ctl[2], 1);
      if (tid < R && myr > c) res_store(rb + myr, tau * yown, RR > 0 ? colbuf[tid] : rows[(size_t)tid * LDr + c + 1]);
      // ---- poll the records of the rows j > c (all loads of a round in flight together; no predicate per lane: a
      // predicated load would let the compiler touch the register early.  A lane without a record of its own reads record
      // c + 1: a record per wave for them, spread over the channels, measured no better -- 12.8 vs 12.6 ms at n = 1536)
      bool need[Q];
      unsigned ptr[Q];     // byte offsets from the buffer (a scalar base + a 32-bit lane offset: half the registers of pointers)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int j = tid + 256 * q;
        need[q] = j > c && j < m;
        ptr[q] = 16u * (unsigned)(need[q] ? j : c + 1);
      }
      for (int dly = 0; dly < g.delay; ++dly) __builtin_amdgcn_s_sleep(1);
      v4i r4[Q];
      // (no load at all for the chunks of 256 rows that are done, behind a uniform test: measured slower, 11.0 -> 12.6 ms
      // at n = 1536)
      long spins = 0;
      bool lost = false;
      while (true) {
#pragma unroll
        for (int q = 0; q < Q; ++q) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(r4[q]) : "v"(ptr[q]), "s"(rb) : "memory");
        if constexpr (Q == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r4[0])::"memory");
        if constexpr (Q == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r4[0]), "+v"(r4[1])::"memory");
        if constexpr (Q == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3])::"memory");
        if constexpr (Q == 6)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]), "+v"(r4[4]), "+v"(r4[5])::"memory");
        if constexpr (Q == 8)
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]), "+v"(r4[4]), "+v"(r4[5]), "+v"(r4[6]), "+v"(r4[7])::"memory");
        if constexpr (Q == 10)
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]), "+v"(r4[4]), "+v"(r4[5]), "+v"(r4[6]), "+v"(r4[7]),
                         "+v"(r4[8]), "+v"(r4[9])::"memory");
        if constexpr (Q == 12)
          asm volatile("s_waitcnt vmcnt(0)"
                       : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]), "+v"(r4[4]), "+v"(r4[5]), "+v"(r4[6]), "+v"(r4[7]),
                         "+v"(r4[8]), "+v"(r4[9]), "+v"(r4[10]), "+v"(r4[11])::"memory");
        // (a record that has arrived is read again in the rounds that wait for the others: pointing those loads at one
        // record instead made the rounds slower, 14.2 -> 17.4 ms at n = 1536)
        bool all = true;
#pragma unroll
        for (int q = 0; q < Q; ++q) all = all && ((r4[q].x & r4[q].y) != -1);   // (a lane without a record reads a needed one)
        if (__all(all)) break;
        ++spins;
        if ((spins & 63) == 0 && __hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) { lost = true; break; }
        if (spins > (1L << 22)) {
          if (lane == 0 && atomicExch(&ctl[2], 1) != 1) { ctl[3] = c; ctl[4] = k; ctl[5] = wave; }
          lost = true;
          break;
        }
      }
      RES_STAMP(0);                                              // [0] publish + poll
      if (lost && lane == 0) s_flag[1] = 1;
      // the buffer of step c - 1 is free: every workgroup has published step c, so it has read step c - 1
      if (tid < R && myr > c) res_store(rec + (size_t)((c + 2) % 3) * RL + myr, res_pair(-1, -1), res_pair(-1, -1));
      double dp = 0.0;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        w[q] = need[q] ? res_pair(r4[q].x, r4[q].y) : 0.0;     // w~ = tau y
        a[q] = need[q] ? res_pair(r4[q].z, r4[q].w) : 0.0;     // A[j, c+1] before this step's update
        dp += w[q] * v[q];
      }
      if (tid == ((c + 1) & 255)) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (q == ((c + 1) >> 8)) bc[0] = w[q];
      }
      dp = res_wave_sum63(dp);
      if (lane == 63) red[wave] = dp;
      lds_barrier();
      if (s_flag[1]) return;
      const double alpha2 = -0.5 * tau * ((red[0] + red[1]) + (red[2] + red[3]));
      const double wc1 = bc[0] + alpha2;                       // w[c+1]  (v[c+1] = 1)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (need[q]) {
          w[q] += alpha2 * v[q];
          a[q] = (a[q] - v[q] * wc1) - w[q];
        }
      }
    }
    RES_STAMP(1);                                                // [1] w~.v, w, a
    // ---- column c + 1 of the updated matrix is a[c+1 ..]: its diagonal entry, the pivot a[c+2], the norm below it
    double np = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = tid + 256 * q;
      if (j >= c + 3) np += a[q] * a[q];                        // (a is zero from row m on)
    }
    if (tid == ((c + 2) & 255)) {
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (q == ((c + 2) >> 8)) bc[1] = a[q];
    }
    if (tid == ((c + 1) & 255)) {
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (q == ((c + 1) >> 8)) bc[2] = a[q];
    }
    if (ownmask) {                                              // own rows: this step's v and w at them, for the update below
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (ownmask >> q & 1) {
          const int i = (tid + 256 * q - k) >> lp;
          if constexpr (RR > 0) {
            vrow[2 * i] = v[q];
            vrow[2 * i + 1] = w[q];
          } else {
            vrow[i] = v[q];
            wrow[i] = w[q];
          }
        }
    }
    np = res_wave_sum63(np);
    if (lane == 63) red[4 + wave] = np;
    lds_barrier();
    const double xn2 = (red[4] + red[5]) + (red[6] + red[7]);
    const HH h = householder(bc[1], xn2);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = tid + 256 * q;
      vn[q] = j > c + 2 ? h.scale * a[q] : (j == c + 2 ? 1.0 : 0.0);
    }
    if (k == ((c + 1) & (P - 1))) {
      if (c + 1 <= m - 3) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int j = tid + 256 * q;
          if (j >= c + 2 && j < m) A[(size_t)(c + 1) * n + j] = vn[q];
        }
      }
      if (tid == 0) {
        d_out[c + 1] = bc[2];
        e_out[c + 1] = h.beta;      // (column m - 2: no entries below the pivot, householder returns beta = the pivot, tau = 0)
        tau_out[c + 1] = h.tau;
      }
    }
    RES_STAMP(2);                                                // [2] norm, next reflector, its stores
    // ---- own rows >= c + 2, columns >= c + 2: A -= v w^T + w v^T, y = A vn
    double yp[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};    // [u]: own row i0 + u  (RR > 0: own row u)
    double yp2[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // RR > 0: own row 8 + u
    const int i0 = c + 2 > k ? (c + 2 - k + P - 1) >> lp : 0;   // the own rows from i0 on are live
    if constexpr (RR > 0) {
      const int q0 = (c + 2) >> 8;
      double vr[RR], wr[RR];
#pragma unroll
      for (int i = 0; i < RR; ++i) {
        vr[i] = vrow[2 * i];
        wr[i] = vrow[2 * i + 1];
      }
      // (rows that are done are skipped; inside a live row only the lower half of the chunks is tested: v, w and vn are
      // zero at the columns that are done, and a test per chunk cost more than the products it saved)
#pragma unroll
      for (int i = 0; i < RR; ++i) {
        if (i >= i0 && i < R) {     // (uniform)
          double acc = 0.0;
          if (q0 < Q / 2) {
#pragma unroll
            for (int q = 0; q < Q / 2; ++q) {
              xr[i][q] -= vr[i] * w[q] + wr[i] * v[q];
              acc += xr[i][q] * vn[q];
            }
          }
#pragma unroll
          for (int q = Q / 2; q < Q; ++q) {
            xr[i][q] -= vr[i] * w[q] + wr[i] * v[q];
            acc += xr[i][q] * vn[q];
          }
          if (i < 8) yp[i & 7] = acc; else yp2[i & 7] = acc;
        }
      }
      if (tid == ((c + 2) & 255)) {   // the own rows' entries of column c + 2: what the next step's records carry
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (q == q0) {
#pragma unroll
            for (int i = 0; i < RR; ++i) colbuf[i] = xr[i][q];
          }
      }
    } else {
      double* row = rows + (size_t)i0 * LDr + tid;
      const int q0 = (c + 2) >> 8;
      switch (R - i0) {
        case 1: res_pass<Q, 1>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        case 2: res_pass<Q, 2>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        case 3: res_pass<Q, 3>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        case 4: res_pass<Q, 4>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        case 5: res_pass<Q, 5>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        case 6: res_pass<Q, 6>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        case 7: res_pass<Q, 7>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        case 8: res_pass<Q, 8>(row, vrow + i0, wrow + i0, q0, Qn, v, w, vn, yp); break;
        default: break;   // (no live row left)
      }
    }
    RES_STAMP(3);                                                // [3] the pass over the own rows
    if constexpr (RR > 0) {
      const double ys = res_reduce8(yp), ys2 = res_reduce8(yp2);
      if (lane < 8) {
        red8[wave * 16 + res_reduce8_index(lane)] = ys;
        red8[wave * 16 + 8 + res_reduce8_index(lane)] = ys2;
      }
      lds_barrier();
      if (tid < 16) yown = tid < R ? (red8[tid] + red8[16 + tid]) + (red8[32 + tid] + red8[48 + tid]) : 0.0;
    } else {
      const double ys = res_reduce8(yp);
      if (lane < 8) red8[wave * 8 + res_reduce8_index(lane)] = ys;
      lds_barrier();
      if (tid < 8) {
        const int u = tid - i0;
        yown = (u >= 0 && tid < R) ? (red8[u] + red8[8 + u]) + (red8[16 + u] + red8[24 + u]) : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = vn[q];
    tau = h.tau;
  }
#ifdef RES_STAMPS
  if (k == 0 && tid == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_res_stamps[i], st_acc[i]);
    atomicAdd(&g_res_stamps[7], (unsigned long long)(m - 1));
  }
#endif
  // the last diagonal entry: with the row's owner, written by the thread that reads it here
  {
    const int jl = m - 1, dj = jl - k;
    if (dj >= 0 && (dj & (P - 1)) == 0 && tid == (jl & 255)) {
      if constexpr (RR > 0) {
#pragma unroll
        for (int i = 0; i < RR; ++i)
#pragma unroll
          for (int q = 0; q < Q; ++q)
            if (i == (dj >> lp) && q == (jl >> 8)) d_out[jl] = xr[i][q];
      } else {
        d_out[jl] = rows[(size_t)(dj >> lp) * LDr + jl];
      }
    }
  }
}

// The take-over of k_sytrd_resident, enqueued behind every one of its launches: ONE workgroup that returns at once unless
// the roll call was aborted (matrix untouched) or a wait was lost in mid-run (whole-matrix launches: the upper triangle
// still holds the matrix; a trailing-matrix launch cannot be restarted and fails the solve through the status word).
// Then LAPACK's dsytd2 from memory on the full symmetric matrix (the other triangle is filled in first), a thread per row:
// slow -- the matrix is streamed twice per column by one CU -- and never expected.
__global__ __launch_bounds__(1024) void k_sytrd_takeover(ResArgs g) {
  __shared__ double vs[kResMaxM], wsv[kResMaxM], sh[16];   // (48 KB)
  const int m = g.m, n = g.L.n, tid = threadIdx.x;
  double* Aw = g.a + (size_t)blockIdx.x * g.stride_a;
  double* A = Aw + (size_t)g.off * n + g.off;
  double* tri = g.tri + (size_t)blockIdx.x * g.L.slab;
  const int* ctl = reinterpret_cast<const int*>(tri + g.L.rctl);
  const bool lost = ctl[2] == 1;
  if (ctl[0] == 1 && !lost) return;
  if (lost && !g.full) {
    if (tid == 0) atomicMax(g.status + 1, 1ull);
    return;
  }
  if (tid == 0) {
    atomicAdd(g.status + 7, 1ull);
    atomicAdd(g.status + (lost ? 9 : 8), 1ull);   // why: [8] the roll call was aborted, [9] a wait was lost in mid-run
    // where: (step, workgroup) of the lost wait; the arrivals counted when the roll call was aborted
    g.status[10] = lost ? (((unsigned long long)(unsigned)ctl[3] << 32) | (unsigned)ctl[4]) : (unsigned long long)(ctl[1] + 1);
  }
  const double f = tri[g.L.hscale + 2];
  const bool bad = *reinterpret_cast<const unsigned long long*>(tri + g.L.hscale + 3) != 0ull;
  for (size_t idx = tid; idx < (size_t)m * m; idx += 1024) {
    const int j = (int)(idx / m), i = (int)(idx - (size_t)j * m);    // row i, column j
    if (i <= j) continue;
    if (lost) {   // lower <- upper (g.off = 0); the scaling pass has seen the upper entries with (row | 1) >= column only
      const double x = bad ? 0.0 : A[(size_t)i * n + j] * (((j | 1) >= i || f == 1.0) ? 1.0 : f);
      A[(size_t)j * n + i] = x;
      A[(size_t)i * n + j] = x;
    } else {
      A[(size_t)i * n + j] = A[(size_t)j * n + i];
    }
  }
  __syncthreads();
  double* d_out = tri + g.L.d + g.off;
  double* e_out = tri + g.L.e + g.off;
  double* tau_out = tri + g.L.tau + g.off;
  for (int c = 0; c <= m - 3; ++c) {
    double np = 0.0;
    for (int j = c + 2 + tid; j < m; j += 1024) { const double x = A[(size_t)c * n + j]; np += x * x; }
    const double xn2 = block_sum_any(np, sh);
    const HH h = householder(A[(size_t)c * n + c + 1], xn2);
    for (int j = c + 1 + tid; j < m; j += 1024) {
      const double x = j == c + 1 ? 1.0 : h.scale * A[(size_t)c * n + j];
      vs[j] = x;
    }
    __syncthreads();
    for (int j = c + 1 + tid; j < m; j += 1024) A[(size_t)c * n + j] = vs[j];
    if (tid == 0) { d_out[c] = A[(size_t)c * n + c]; e_out[c] = h.beta; tau_out[c] = h.tau; }
    if (h.tau != 0.0) {   // (block-uniform)
      double dp = 0.0;
      for (int i = c + 1 + tid; i < m; i += 1024) {
        double y = 0.0;
        for (int j = c + 1; j < m; ++j) y += A[(size_t)j * n + i] * vs[j];
        y *= h.tau;
        wsv[i] = y;
        dp += y * vs[i];
      }
      const double alpha2 = -0.5 * h.tau * block_sum_any(dp, sh);
      for (int i = c + 1 + tid; i < m; i += 1024) wsv[i] += alpha2 * vs[i];
      __syncthreads();
      for (int i = c + 1 + tid; i < m; i += 1024) {
        const double vi = vs[i], wi = wsv[i];
        for (int j = c + 1; j < m; ++j) A[(size_t)j * n + i] -= vi * wsv[j] + wi * vs[j];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    d_out[m - 2] = A[(size_t)(m - 2) * n + m - 2];
    e_out[m - 2] = A[(size_t)(m - 2) * n + m - 1];
    tau_out[m - 2] = 0.0;
    d_out[m - 1] = A[(size_t)(m - 1) * n + m - 1];
  }
}

}  // namespace

int mirror_lower_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch) {
  dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(k_mirror_lower, grid, dim3(32, 8), 0, ctx->stream, d_a, stride_a, n);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// NumPy lower triangle -> column-major lower triangle, then the scaling described above (factor kept in the tri slab).
int prepare_matrix_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, double* d_tri_ws,
                           const TriLayout& L) {
  SC_TRY(mirror_lower_batched(ctx, d_a, stride_a, n, batch));
  hipStream_t st = ctx->stream;
  for (int b = 0; b < batch; ++b)
    SC_HIP(ctx, hipMemsetAsync(d_tri_ws + (size_t)b * L.slab + L.hscale + 1, 0, 3 * sizeof(double), st));
  const unsigned gx = (unsigned)std::min<size_t>(1024, ((size_t)n * n + 255) / 256);
  hipLaunchKernelGGL(k_absmax_lower, dim3(gx, (unsigned)batch), dim3(256), 0, st, d_a, stride_a, n, d_tri_ws, L,
                     ctx->d_status);
  hipLaunchKernelGGL(k_scale_lower, dim3(gx, (unsigned)batch), dim3(256), 0, st, d_a, stride_a, n, d_tri_ws, L);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// eigenvalues of the scaled matrix -> eigenvalues of the caller's matrix; d_w: (batch) m values each
int unscale_values_batched(sc_ctx* ctx, double* d_w, long long stride_w, int m, int batch, const double* d_tri_ws,
                           const TriLayout& L) {
  hipLaunchKernelGGL(k_unscale_values, dim3((unsigned)((m + 255) / 256), (unsigned)batch), dim3(256), 0, ctx->stream, d_w,
                     stride_w, m, d_tri_ws, L);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}


// Whether (and from which column) k_sytrd_resident takes the reduction of this solve: one matrix (a batch is better
// served by launches that all its matrices share), the trailing matrix from the first panel boundary at which its order
// is <= 2048 -- the whole matrix for n <= 2048.  SPRINGCRAFT_RESIDENT = 0 switches it off; SPRINGCRAFT_RESIDENT_WGS = the
// workgroups (a power of two; at least order / 8, at most the CUs of the device).
struct ResPlan {
  int off, m, P, logP, Q;
  bool reg;       // rows in registers
  size_t lds;
};
bool resident_enabled(const sc_ctx* ctx) {
  static const int env_mode = [] { const char* e = getenv("SPRINGCRAFT_RESIDENT"); return e ? atoi(e) : -1; }();
  const int mode = (ctx && ctx->resident_mode >= 0) ? ctx->resident_mode : env_mode;
  return mode != 0 && !(ctx && ctx->resident_ok == 0);
}
static bool resident_plan(sc_ctx* ctx, int n, int batch, int nb, ResPlan* out) {
  static const int env_wgs = [] { const char* e = getenv("SPRINGCRAFT_RESIDENT_WGS"); return e ? atoi(e) : 0; }();
  static const int env_max = [] { const char* e = getenv("SPRINGCRAFT_RESIDENT_MAX"); return e ? atoi(e) : 0; }();
  if (!resident_enabled(ctx) || batch != 1) return false;
  sc_host::ResidentShape S{};
  const int want = ctx->resident_wgs > 0 ? ctx->resident_wgs : env_wgs;
  if (!sc_host::resident_shape(n, nb, ctx->num_cus > 0 ? ctx->num_cus : 256, env_max, want, &S)) return false;
  ResPlan R{};
  R.off = S.off; R.m = S.m; R.P = S.P; R.logP = S.logP; R.Q = S.Q; R.reg = S.reg; R.lds = S.lds_bytes;
  *out = R;
  return true;
}

template <int Q, int RR>
static int launch_resident_q(sc_ctx* ctx, const ResArgs& g, const ResPlan& R, int batch) {
  if (!sc_raise_dyn_lds(reinterpret_cast<const void*>(&k_sytrd_resident<Q, RR>), 160 * 1024)) return 1;
  hipLaunchKernelGGL((k_sytrd_resident<Q, RR>), dim3((unsigned)R.P, (unsigned)batch), dim3(256), R.lds, ctx->stream, g);
  return SC_OK;
}

// Two launches of k_sytrd_resident must not run side by side: each wants every CU, both would sit in their roll calls with
// half of their workgroups until one gives up.  Within a process the launches on a device are chained by an event: a
// launch waits (on the device, in its stream) for the previous launch on that device, whatever context and stream it
// came from.  (Another process on the same GPU is not seen: the roll call and the take-over are for that.)
// The wait, the launch and the record of one solve are one critical section (g_resident_mu, taken by launch_resident):
// two host threads with their own contexts would otherwise both wait for the same predecessor and then run side by side.
static std::mutex g_resident_mu;
static int resident_chain(sc_ctx* ctx, hipStream_t st, bool before) {
  static std::map<int, hipEvent_t> last;
  static const bool off = getenv("SPRINGCRAFT_RESIDENT_NO_CHAIN") != nullptr;   // (diagnostic: what the chain is for)
  if (off) return SC_OK;
  auto it = last.find(ctx->device);
  if (before) {
    if (it != last.end()) SC_HIP(ctx, hipStreamWaitEvent(st, it->second, 0));
    return SC_OK;
  }
  if (it == last.end()) {
    hipEvent_t e = nullptr;
    SC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    it = last.emplace(ctx->device, e).first;
  }
  SC_HIP(ctx, hipEventRecord(it->second, st));
  return SC_OK;
}

// SC_OK: enqueued (with its take-over); 1: not available on this device (the caller goes on with the launches per column)
static int launch_resident(sc_ctx* ctx, double* d_a, long long stride_a, int batch, double* d_ws, const TriLayout& L,
                           const ResPlan& R) {
  hipStream_t st = ctx->stream;
  ResArgs g{};
  g.a = d_a; g.stride_a = stride_a; g.tri = d_ws; g.L = L;
  g.off = R.off; g.m = R.m; g.P = R.P; g.logP = R.logP;
  g.full = R.off == 0 ? 1 : 0;
  g.hook = ctx->resident_hook;
  // (the first poll a microsecond or so after the publication: polls that come back empty slow the stores they wait
  // for -- all workgroups read all records --; n = 1536: 14.3 ms without the pause, 11.7 with 24 x 64 cycles, 13.8 with 64)
  static const int env_delay = [] { const char* e = getenv("SPRINGCRAFT_RESIDENT_DELAY"); return e ? atoi(e) : 24; }();
  g.delay = env_delay;
  g.status = ctx->d_status;
  for (int b = 0; b < batch; ++b)
    SC_HIP(ctx, hipMemsetAsync(d_ws + (size_t)b * L.slab + L.rctl, 0xFF, tri_resident_bytes(L.n), st));
  std::lock_guard<std::mutex> chain_lock(g_resident_mu);
  SC_TRY(resident_chain(ctx, st, true));
  int rc = 1;
  switch (R.Q) {
    case 1: rc = launch_resident_q<1, 0>(ctx, g, R, batch); break;
    case 2: rc = launch_resident_q<2, 0>(ctx, g, R, batch); break;
    case 4: rc = launch_resident_q<4, 0>(ctx, g, R, batch); break;
    case 6: rc = launch_resident_q<6, 0>(ctx, g, R, batch); break;
    case 8: rc = launch_resident_q<8, 0>(ctx, g, R, batch); break;
    case 10: rc = launch_resident_q<10, 10>(ctx, g, R, batch); break;
    default: rc = launch_resident_q<12, 12>(ctx, g, R, batch); break;
  }
  if (rc != SC_OK) return rc;
  hipLaunchKernelGGL(k_sytrd_takeover, dim3((unsigned)batch), dim3(1024), 0, st, g);
  SC_HIP(ctx, hipGetLastError());
  SC_TRY(resident_chain(ctx, st, false));
  ++ctx->cnt_resident_launches;
  return SC_OK;
}

// ---- diagnostic build only (-DRES_STAMPS, else returns 1): out8[0..5] = cycles of workgroup 0 in the six segments of a
// step of k_sytrd_resident (see RES_STAMP), out8[7] = steps; summed since the last call, reset by it
extern "C" int sc_dbg_resident_stamps(unsigned long long* out8) {
#ifdef RES_STAMPS
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_res_stamps), 64) != hipSuccess) return 5;
  const unsigned long long z[8] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_res_stamps), z, 64) == hipSuccess ? 0 : 5;
#else
  (void)out8;
  return 1;
#endif
}

int tridiag_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, double* d_ws,
                    const TriLayout& L, const GemmDesc* d_syr2k_descs, float* ms_symv,
                    float* ms_syr2k) {
  hipStream_t st = ctx->stream;
  const int nb = L.nb;
  ResPlan RP{};
  bool use_res = resident_plan(ctx, n, batch, nb, &RP);
  // persistent SYMV grid: 4 resident blocks per CU (LDS-limited), shared by the matrices of the batch
  const int symv_blocks = 4 * (ctx->num_cus > 0 ? ctx->num_cus : 256);
  ScopedEvents<4> ev;
  const bool prof = ctx->profiling && ms_symv && ms_syr2k;
  if (prof) {
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    *ms_symv = 0.f;
    *ms_syr2k = 0.f;
  }
  int panel = 0;
  for (int j0 = 0; j0 < n; j0 += nb, ++panel) {
    if (use_res && j0 == RP.off) {
      PhaseTimer t_res(ctx, "resident_tridiag", st);
      t_res.start();
      const int rc = launch_resident(ctx, d_a, stride_a, batch, d_ws, L, RP);
      t_res.stop();
      if (rc == SC_OK) {
        t_res.finish();
        break;
      }
      if (rc != 1) return rc;
      use_res = false;   // (the LDS size was refused: the launches per column)
    }
    const int pend = j0 + nb < n ? j0 + nb : n;
    hipLaunchKernelGGL(k_zero_panels, dim3(64, (unsigned)batch), dim3(256), 0, st, d_ws, L);
    for (int c = j0; c < pend; ++c) {
      const int nba = (n + 63) / 64;
      hipLaunchKernelGGL(k_col_update, dim3((unsigned)(nba - c / 64), (unsigned)batch), dim3(1024),
                         0, st, d_a, stride_a, d_ws, L, c, j0);
      if (c <= n - 3) {
        const int nt = nba - (c + 1) / 64;
        if (prof) SC_HIP(ctx, hipEventRecord(ev[0], st));
        const int ntiles = nt * (nt + 1) / 2;
        const int gx = std::min(ntiles, std::max(256, symv_blocks / batch));
        if (n % 2 == 0)
          hipLaunchKernelGGL(k_symv_tiles<true>, dim3((unsigned)gx, (unsigned)batch), dim3(256), 0, st, d_a,
                             stride_a, d_ws, L, c, j0, ntiles);
        else
          hipLaunchKernelGGL(k_symv_tiles<false>, dim3((unsigned)gx, (unsigned)batch), dim3(256), 0, st, d_a,
                             stride_a, d_ws, L, c, j0, ntiles);
        if (prof) {
          SC_HIP(ctx, hipEventRecord(ev[1], st));
          SC_HIP(ctx, hipEventSynchronize(ev[1]));
          float ms = 0.f;
          SC_HIP(ctx, hipEventElapsedTime(&ms, ev[0], ev[1]));
          *ms_symv += ms;
        }
        hipLaunchKernelGGL(k_w_reduce, dim3((unsigned)nt, (unsigned)batch), dim3(1024), 0,
                           st, d_a, stride_a, d_ws, L, c, j0);
      }
    }
    if (pend < n) {
      // The panel's last reflector still has its alpha2 pending, unless a later k_col_update of the same
      // panel (columns n-2, n-1 carry no reflector) has already applied it.
      const int cl = pend - 1;
      if (cl <= n - 3) {
        const int m = n - cl - 1;
        hipLaunchKernelGGL(k_w_fix, dim3((unsigned)((m + 255) / 256), (unsigned)batch), dim3(256), 0, st,
                           d_ws, L, cl, j0);
      }
      const int mt = n - pend;
      if (prof) SC_HIP(ctx, hipEventRecord(ev[2], st));
      SC_TRY(launch_gemm_f64(ctx, d_syr2k_descs + (size_t)panel * batch, batch, mt, mt, kGemmTile, 1, false, false,
                             kGemmAmBn));
      if (prof) {
        SC_HIP(ctx, hipEventRecord(ev[3], st));
        SC_HIP(ctx, hipEventSynchronize(ev[3]));
        float ms = 0.f;
        SC_HIP(ctx, hipEventElapsedTime(&ms, ev[2], ev[3]));
        *ms_syr2k += ms;
      }
    }
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}
