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
#include "eigh_internal.h"

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
  if (xn2 == 0.0) {
    h.beta = alpha; h.tau = 0.0; h.scale = 0.0;
    return h;
  }
  h.beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
  h.tau = (h.beta - alpha) / h.beta;
  h.scale = 1.0 / (alpha - h.beta);
  return h;
}

__device__ __forceinline__ double sum_partials(const double* p, int cnt) {
  double s = 0.0;
  for (int q = 0; q < cnt; ++q) s += p[q];
  return s;
}

// ---- column update -------------------------------------------------------------------------------------
// rows r = c .. n-1, one thread per row.  i = c - j0 reflectors of the current panel are pending.
__global__ __launch_bounds__(256) void k_col_update(double* __restrict__ a_all, long long stride_a,
                                                    double* __restrict__ ws_all, TriLayout L, int c,
                                                    int j0) {
  __shared__ double rowV[64], rowW[64], red[5];
  const int n = L.n, nb = L.nb, i = c - j0;
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  double* VW = ws + L.vw;
  double* WV = ws + L.wv;
  const int tid = threadIdx.x;
  const int r = c + blockIdx.x * 256 + tid;

  double alpha2 = 0.0;
  if (i > 0) {
    const double tau_prev = ws[L.tau + c - 1];
    if (tau_prev != 0.0) {
      const int cnt = (n - c + 255) / 256;  // blocks of k_w_reduce(c-1): m_prev = n - c rows
      alpha2 = -0.5 * tau_prev * sum_partials(ws + L.wvpart, cnt);
    }
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

  double sq = 0.0;
  if (r < n) {
    double a = A[(size_t)c * n + r];
    for (int p = 0; p < i; ++p) {
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
      a -= v * rowW[p] + w * rowV[p];
    }
    ws[L.xraw + r] = a;
    if (r == c) ws[L.d + c] = a;
    if (r == c + 1 && c == n - 2) {  // last sub-diagonal element: no reflector (dsytd2: tau = 0)
      ws[L.e + c] = a;
      ws[L.tau + c] = 0.0;
    }
    if (r >= c + 2) sq = a * a;
  }
  const double s = block_sum_bcast(sq, red);
  if (tid == 0) ws[L.npart + blockIdx.x] = s;
}

// ---- symmetric matrix-vector product on lower 64x64 tiles ---------------------------------------------------
// grid.x = nt (nt+1)/2 tiles (bi >= bj) of the trailing matrix A22 = A[c+1:, c+1:], m = n-c-1.
__global__ __launch_bounds__(256) void k_symv_tiles(double* __restrict__ a_all, long long stride_a,
                                                    double* __restrict__ ws_all, TriLayout L, int c,
                                                    int j0) {
  constexpr int TS = 64, LD = 65;
  __shared__ double T[TS * LD];
  __shared__ double xi[TS], xj[TS];
  __shared__ double red[4][TS];
  const int n = L.n, nb = L.nb, i = c - j0;
  const int m = n - c - 1;
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  const int tid = threadIdx.x;

  // Householder scalars (every block recomputes them from the same partials, in the same order)
  const double xn2 = sum_partials(ws + L.npart, (n - c + 255) / 256);
  const double alpha = ws[L.xraw + c + 1];
  const HH h = householder(alpha, xn2);

  // tile coordinates
  const int t = blockIdx.x;
  int bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1) / 2 > t) --bi;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  const int bj = t - bi * (bi + 1) / 2;
  const bool diag = (bi == bj);
  const int r0 = bi * TS, k0 = bj * TS;

  if (tid < TS) {
    const int rr = r0 + tid;
    xi[tid] = rr < m ? (rr == 0 ? 1.0 : ws[L.xraw + c + 1 + rr] * h.scale) : 0.0;
  } else if (tid < 2 * TS) {
    const int kk = k0 + tid - TS;
    xj[tid - TS] = kk < m ? (kk == 0 ? 1.0 : ws[L.xraw + c + 1 + kk] * h.scale) : 0.0;
  }
  // load tile: element (rr, kk) = A22[r0+rr, k0+kk] = A[(c+1+r0+rr) + (c+1+k0+kk) n]
  {
    const int rr = tid & 63;
    const int kq = tid >> 6;
    const int gr = r0 + rr;
    const double* base = A + (size_t)(c + 1 + k0) * n + (c + 1 + gr);
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
      const int kk = kq + 4 * q;
      const int gk = k0 + kk;
      double v = 0.0;
      if (gr < m && gk < m) v = base[(size_t)kk * n];
      T[rr + kk * LD] = v;
    }
  }
  __syncthreads();

  const int lane = tid & 63, q = tid >> 6;
  if (diag) {
    // y[r] = sum_k Tsym[r,k] x[k], Tsym from the lower triangle only
    double acc = 0.0;
#pragma unroll 4
    for (int kk = q * 16; kk < q * 16 + 16; ++kk) {
      const double tv = (lane >= kk) ? T[lane + kk * LD] : T[kk + lane * LD];
      acc += tv * xj[kk];
    }
    red[q][lane] = acc;
    __syncthreads();
    if (tid < TS && r0 + tid < m)
      ws[L.ypart + (size_t)bj * n + r0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  } else {
    double acc = 0.0;
#pragma unroll 4
    for (int kk = q * 16; kk < q * 16 + 16; ++kk) acc += T[lane + kk * LD] * xj[kk];
    red[q][lane] = acc;
    __syncthreads();
    if (tid < TS && r0 + tid < m)
      ws[L.ypart + (size_t)bj * n + r0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    __syncthreads();
    // transposed product: y_j[k] += sum_r T[r,k] x_i[r]
    acc = 0.0;
#pragma unroll 4
    for (int rr = q * 16; rr < q * 16 + 16; ++rr) acc += T[rr + lane * LD] * xi[rr];
    red[q][lane] = acc;
    __syncthreads();
    if (tid < TS && k0 + tid < m)
      ws[L.ypart + (size_t)bi * n + k0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }

  if (diag) {
    // store v (explicit leading 1) for the back-transformation and into both panel copies
    double* VW = ws + L.vw;
    double* WV = ws + L.wv;
    if (tid < TS && r0 + tid < m) {
      const double xv = xi[tid];
      const size_t row = (size_t)c + 1 + r0 + tid;
      A[(size_t)c * n + row] = xv;
      VW[(size_t)i * n + row] = xv;
      WV[(size_t)(nb + i) * n + row] = xv;
    }
    // partial dots for the panel corrections: dpart[bi][p] = sum_rows V[r,p] x[r]  (p < i),
    //                                         dpart[bi][nb+p] = sum_rows W[r,p] x[r]
    const int row = c + 1 + r0 + lane;
    const bool rv = (r0 + lane) < m;
    const double xv = rv ? xi[lane] : 0.0;
    for (int p = q; p < 2 * i; p += 4) {
      const int col = p < i ? p : nb + (p - i);
      double v = rv ? VW[(size_t)col * n + row] * xv : 0.0;
      v = wave_sum(v);
      if (lane == 0) ws[L.dpart + (size_t)bi * 2 * nb + col] = v;
    }
    if (t == 0 && tid == 0) {
      ws[L.e + c] = h.beta;
      ws[L.tau + c] = h.tau;
    }
  }
}

// ---- reduce partial products, apply panel corrections, scale by tau ---------------------------------------
// one thread per row r' = 0..m-1 of v / w
__global__ __launch_bounds__(256) void k_w_reduce(double* __restrict__ a_all, long long stride_a,
                                                  double* __restrict__ ws_all, TriLayout L, int c,
                                                  int j0) {
  __shared__ double dots[128], red[5];
  const int n = L.n, nb = L.nb, i = c - j0;
  const int m = n - c - 1;
  const int nt = (m + 63) / 64;
  const double* A = a_all + (size_t)blockIdx.y * stride_a;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  double* VW = ws + L.vw;
  double* WV = ws + L.wv;
  const int tid = threadIdx.x;

  if (tid < 2 * nb) {
    const int p = tid < nb ? tid : tid - nb;
    double s = 0.0;
    if (p < i)
      for (int b = 0; b < nt; ++b) s += ws[L.dpart + (size_t)b * 2 * nb + tid];
    dots[tid] = s;  // dots[p] = V_p^T v, dots[nb+p] = W_p^T v
  }
  __syncthreads();

  const double tau = ws[L.tau + c];
  const int rr = blockIdx.x * 256 + tid;
  double wv = 0.0;
  if (rr < m) {
    const size_t row = (size_t)c + 1 + rr;
    double y = 0.0;
    for (int o = 0; o < nt; ++o) y += ws[L.ypart + (size_t)o * n + rr];
    for (int p = 0; p < i; ++p) {
      const double v = VW[(size_t)p * n + row];
      const double w = VW[(size_t)(nb + p) * n + row];
      y -= v * dots[nb + p] + w * dots[p];
    }
    const double wt = tau * y;
    VW[(size_t)(nb + i) * n + row] = wt;
    WV[(size_t)i * n + row] = wt;
    wv = wt * A[(size_t)c * n + row];  // v was stored in A[:, c] by k_symv_tiles
  }
  const double s = block_sum_bcast(wv, red);
  if (tid == 0) ws[L.wvpart + blockIdx.x] = s;
}

// ---- end of panel: apply the pending alpha2 of the panel's last reflector --------------------------------------
__global__ __launch_bounds__(256) void k_w_fix(double* __restrict__ ws_all, TriLayout L, int c, int j0) {
  const int n = L.n, nb = L.nb, i = c - j0;
  const int m = n - c - 1;
  double* ws = ws_all + (size_t)blockIdx.y * L.slab;
  double* VW = ws + L.vw;
  double* WV = ws + L.wv;
  const double tau = ws[L.tau + c];
  if (tau == 0.0) return;
  const double alpha2 = -0.5 * tau * sum_partials(ws + L.wvpart, (m + 255) / 256);
  const int rr = blockIdx.x * 256 + threadIdx.x;
  if (rr < m) {
    const size_t row = (size_t)c + 1 + rr;
    const double w = VW[(size_t)(nb + i) * n + row] + alpha2 * VW[(size_t)i * n + row];
    VW[(size_t)(nb + i) * n + row] = w;
    WV[(size_t)i * n + row] = w;
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

}  // namespace

int tridiag_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, double* d_ws,
                    const TriLayout& L, const GemmDesc* d_syr2k_descs, float* ms_symv,
                    float* ms_syr2k) {
  hipStream_t st = ctx->stream;
  const int nb = L.nb;
  {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), (unsigned)batch);
    hipLaunchKernelGGL(k_mirror_lower, grid, dim3(32, 8), 0, st, d_a, stride_a, n);
  }
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  const bool prof = ctx->profiling && ms_symv && ms_syr2k;
  if (prof) {
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    *ms_symv = 0.f;
    *ms_syr2k = 0.f;
  }
  int panel = 0;
  for (int j0 = 0; j0 < n; j0 += nb, ++panel) {
    const int pend = j0 + nb < n ? j0 + nb : n;
    hipLaunchKernelGGL(k_zero_panels, dim3(64, (unsigned)batch), dim3(256), 0, st, d_ws, L);
    for (int c = j0; c < pend; ++c) {
      const int rows = n - c;
      hipLaunchKernelGGL(k_col_update, dim3((unsigned)((rows + 255) / 256), (unsigned)batch), dim3(256),
                         0, st, d_a, stride_a, d_ws, L, c, j0);
      if (c <= n - 3) {
        const int m = n - c - 1;
        const int nt = (m + 63) / 64;
        if (prof) SC_HIP(ctx, hipEventRecord(ev[0], st));
        hipLaunchKernelGGL(k_symv_tiles, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)batch),
                           dim3(256), 0, st, d_a, stride_a, d_ws, L, c, j0);
        if (prof) {
          SC_HIP(ctx, hipEventRecord(ev[1], st));
          SC_HIP(ctx, hipEventSynchronize(ev[1]));
          float ms = 0.f;
          SC_HIP(ctx, hipEventElapsedTime(&ms, ev[0], ev[1]));
          *ms_symv += ms;
        }
        hipLaunchKernelGGL(k_w_reduce, dim3((unsigned)((m + 255) / 256), (unsigned)batch), dim3(256), 0,
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
      SC_TRY(launch_gemm_f64(ctx, d_syr2k_descs + (size_t)panel * batch, batch, mt, mt, 0));
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
  if (prof)
    for (auto& e : ev) (void)hipEventDestroy(e);
  return SC_OK;
}
