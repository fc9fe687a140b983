// Partial spectrum of a symmetric tridiagonal matrix (stage K8): eigenvalues il..iu by Sturm bisection and
// their eigenvectors by inverse iteration, followed by a Cholesky-QR (twice) orthonormalisation of the whole
// selected set.  The reference has no partial-spectrum mode (np.linalg.eigh always returns everything,
// nma.py:61); this path serves BASELINE config 5 (N = 8000 C-alpha, lowest 100 non-trivial modes), where the
// full O(n^3) divide & conquer + back-transformation would be wasted work.
//
//   k_sturm_range   one thread per requested eigenvalue: bisection on the Sturm count
//   k_stein         one thread per eigenvector: LU of (T - lambda I) with partial pivoting, 4 inverse
//                   iterations from a fixed pseudo-random start; factors live in global memory in an
//                   [row][vector] layout so the lanes of a wave touch consecutive addresses
//   CholQR2         G = X^T X (split-K MFMA GEMM), k_chol_inv (one workgroup: Cholesky + inverse of R),
//                   X <- X R^-1 (MFMA GEMM); twice.  Inside a numerically degenerate cluster (the six
//                   rigid-body modes of an ANM) this picks an orthonormal basis of the invariant subspace.
#include <algorithm>
#include <vector>

#include "eigh_internal.h"

namespace {

constexpr double kEpsS = 2.220446049250313e-16;

// One WAVE per requested eigenvalue: multisection.  Every round the 64 lanes count the eigenvalues below 64 equally
// spaced shifts inside the current bracket (the Sturm recurrence is sequential in n but independent between shifts, so
// a round costs what one bisection step costs) and the bracket shrinks 65-fold: ~9 rounds instead of ~53 bisection
// steps; a few plain bisection steps then close the bracket to neighbouring floating-point numbers as before.
__global__ __launch_bounds__(64) void k_sturm_range(const double* __restrict__ tri_all, TriLayout TL, int il,
                                                    int m, double* __restrict__ w_all, long long stride_w) {
  const double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  const double* d = tri + TL.d;
  const double* e = tri + TL.e;
  const int n = TL.n;
  const int lane = threadIdx.x;
  // Gershgorin bounds, lanes strided over the rows
  double lo = 1e300, hi = -1e300, emax = 0.0;
  for (int i = lane; i < n; i += 64) {
    const double el = i > 0 ? fabs(e[i - 1]) : 0.0, er = i < n - 1 ? fabs(e[i]) : 0.0;
    lo = fmin(lo, d[i] - el - er);
    hi = fmax(hi, d[i] + el + er);
    emax = fmax(emax, er);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, off));
    hi = fmax(hi, __shfl_xor(hi, off));
    emax = fmax(emax, __shfl_xor(emax, off));
  }
  const double span = fmax(fabs(lo), fabs(hi));
  const double pivmin = fmax(2.2250738585072014e-308 * fmax(1.0, emax * emax), 1e-290);
  lo -= 2.0 * kEpsS * span * n + 2.0 * pivmin;
  hi += 2.0 * kEpsS * span * n + 2.0 * pivmin;
  const int j = blockIdx.x;
  if (j >= m) return;
  const int k = il + j;
  // (the recurrence is a chain of divisions; d and e come in chunks of 16 requested together, so that the chain waits
  // for memory once per chunk instead of once per row)
  auto count_below = [&](double x) {
    int cnt = 0;
    double q = d[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < 0.0;
    for (int i0 = 1; i0 < n; i0 += 16) {
      double dd[16], ee[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = min(i0 + u, n - 1);
        dd[u] = d[i];
        ee[u] = e[i - 1];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (i0 + u < n) {
          q = dd[u] - x - ee[u] * ee[u] / q;
          if (fabs(q) < pivmin) q = -pivmin;
          cnt += q < 0.0;
        }
      }
    }
    return cnt;
  };
  double a = lo, b = hi;   // invariant: count_below(a) <= k < count_below(b)
  for (int round = 0; round < 16; ++round) {
    const double width = b - a;
    const double x = a + width * ((double)(lane + 1) / 65.0);
    if (__any(!(x > a && x < b))) break;      // the bracket is down to a few numbers (decided for the whole wave)
    const int cnt = count_below(x);
    const unsigned long long above = __ballot(cnt > k);
    double nb = b, na = a;
    if (above != 0ull) {
      const int p = __ffsll((long long)above) - 1;   // first shift that has more than k eigenvalues below it
      nb = __shfl(x, p);
      if (p > 0) na = __shfl(x, p - 1);
    } else {
      na = __shfl(x, 63);
    }
    a = na;
    b = nb;
    if (!(b - a < width)) break;
  }
  for (int it = 0; it < 120; ++it) {
    const double mid = 0.5 * (a + b);
    if (mid <= a || mid >= b) break;
    if (count_below(mid) > k) b = mid; else a = mid;
  }
  if (lane == 0) w_all[(size_t)blockIdx.y * stride_w + j] = 0.5 * (a + b);
}

__device__ __forceinline__ double hash_unit(unsigned a, unsigned b) {
  unsigned long long x = ((unsigned long long)a << 32) ^ (b * 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return ((double)(x >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
}

// Workspace per matrix (doubles): u0,u1,u2,lm : 4 x n x m, piv (as double flags): n x m, all [row][vector].
__global__ __launch_bounds__(64) void k_stein(const double* __restrict__ tri_all, TriLayout TL, int m,
                                              const double* __restrict__ w_all, long long stride_w,
                                              double* __restrict__ fac_all, long long stride_fac,
                                              double* __restrict__ x_all, long long stride_x) {
  const double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  const double* d = tri + TL.d;
  const double* e = tri + TL.e;
  const int n = TL.n;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const double* w = w_all + (size_t)blockIdx.y * stride_w;
  double* fac = fac_all + (size_t)blockIdx.y * stride_fac;
  double* u0 = fac;                          // pivots
  double* u1 = fac + (size_t)n * m;          // first super-diagonal of U
  double* u2 = fac + (size_t)2 * n * m;      // second super-diagonal of U
  double* lm = fac + (size_t)3 * n * m;      // multipliers
  double* pv = fac + (size_t)4 * n * m;      // 1.0 = rows k, k+1 were swapped
  double* X = x_all + (size_t)blockIdx.y * stride_x + (size_t)j * n;   // column j of X (n x m, ld n)

  // scale of T and a tiny separation of (numerically) coincident eigenvalues, as dstein does
  double tnorm = 0.0;
  for (int i = 0; i < n; ++i) tnorm = fmax(tnorm, fabs(d[i]) + (i < n - 1 ? fabs(e[i]) : 0.0) + (i > 0 ? fabs(e[i - 1]) : 0.0));
  if (tnorm == 0.0) tnorm = 1.0;
  const double tiny = kEpsS * tnorm;
  double lam = w[j];
  {
    // shift members of a run of (near-)equal eigenvalues apart by multiples of 10 eps |T|
    int run = 0;
    for (int q = j - 1; q >= 0 && fabs(w[q] - w[q + 1]) <= 10.0 * tiny; --q) ++run;
    lam += run * 10.0 * tiny;
  }
#define AT(arr, k) arr[(size_t)(k) * m + j]
  // LU factorisation of T - lam I with partial pivoting (row k against row k+1)
  double p = d[0] - lam, q = n > 1 ? e[0] : 0.0, r = 0.0;
  for (int k0 = 0; k0 < n - 1; k0 += 8) {
    double e8[9], d8[8];   // e[k0 .. k0 + 8], d[k0 + 1 .. k0 + 8], requested together
#pragma unroll
    for (int u = 0; u < 9; ++u) e8[u] = e[min(k0 + u, n - 2)];
#pragma unroll
    for (int u = 0; u < 8; ++u) d8[u] = d[min(k0 + 1 + u, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k < n - 1) {
        const double sub = e8[u];
        const double dn = d8[u] - lam;
        const double en = (k + 2 < n) ? e8[u + 1] : 0.0;
        if (fabs(sub) > fabs(p)) {          // swap: pivot row is (sub, dn, en)
          const double mult = p / sub;
          AT(u0, k) = sub; AT(u1, k) = dn; AT(u2, k) = en; AT(lm, k) = mult; AT(pv, k) = 1.0;
          p = q - mult * dn;
          q = r - mult * en;
          r = 0.0;
        } else {
          if (fabs(p) < tiny) p = copysign(tiny, p == 0.0 ? 1.0 : p);
          const double mult = sub / p;
          AT(u0, k) = p; AT(u1, k) = q; AT(u2, k) = r; AT(lm, k) = mult; AT(pv, k) = 0.0;
          p = dn - mult * q;
          q = en - mult * r;
          r = 0.0;
        }
      }
    }
  }
  if (fabs(p) < tiny) p = copysign(tiny, p == 0.0 ? 1.0 : p);
  AT(u0, n - 1) = p; AT(u1, n - 1) = 0.0; AT(u2, n - 1) = 0.0;

  for (int i = 0; i < n; ++i) X[i] = hash_unit((unsigned)j + 1u, (unsigned)i + 1u);
  for (int iter = 0; iter < 4; ++iter) {
    // forward: apply the row interchanges and multipliers.  The running entry stays in a register, the factors and
    // the next entries of x come in chunks of 8 rows requested together (the loop is a dependent chain: without that it
    // waits for memory in every row)
    {
      double xk = X[0];
      for (int k0 = 0; k0 < n - 1; k0 += 8) {
        double xn8[8], pv8[8], lm8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = min(k0 + u, n - 2);
          xn8[u] = X[k + 1];
          pv8[u] = AT(pv, k);
          lm8[u] = AT(lm, k);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u < n - 1) {
            double xn = xn8[u];
            if (pv8[u] != 0.0) { const double t = xk; xk = xn; xn = t; }
            xn -= lm8[u] * xk;
            X[k0 + u] = xk;
            xk = xn;
          }
        }
      }
      X[n - 1] = xk;
    }
    // back substitution with the three diagonals of U, same chunking
    double x1 = 0.0, x2 = 0.0, nrm = 0.0;
    for (int k0 = n - 1; k0 >= 0; k0 -= 8) {
      double xr[8], a0[8], a1[8], a2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = max(k0 - u, 0);
        xr[u] = X[k];
        a0[u] = AT(u0, k);
        a1[u] = AT(u1, k);
        a2[u] = AT(u2, k);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k0 - u >= 0) {
          const double xv = (xr[u] - a1[u] * x1 - a2[u] * x2) / a0[u];
          X[k0 - u] = xv;
          x2 = x1;
          x1 = xv;
          nrm = fmax(nrm, fabs(xv));
        }
      }
    }
    // rescale (max-norm) to stay in range; final 2-normalisation is done by the QR step
    const double s = nrm > 0.0 ? 1.0 / nrm : 1.0;
    double ss = 0.0;
    for (int k = 0; k < n; ++k) { const double xv = X[k] * s; X[k] = xv; ss += xv * xv; }
    if (iter == 3) {
      const double inv = 1.0 / sqrt(ss);
      for (int k = 0; k < n; ++k) X[k] *= inv;
    }
  }
#undef AT
}

// G (m x m, sum of `splits` slices, summed in place into slice 0) -> upper Cholesky factor R (G = R^T R, in
// place) -> Rinv (upper, m x m, ld m).  One workgroup per matrix, everything in global memory (m is small;
// __syncthreads() orders the global accesses inside the workgroup).
__global__ __launch_bounds__(256) void k_chol_inv(double* __restrict__ g_all, long long stride_g, int m,
                                                  int splits, double* __restrict__ rinv_all, long long stride_r) {
  double* R = g_all + (size_t)blockIdx.x * stride_g;
  double* Ri = rinv_all + (size_t)blockIdx.x * stride_r;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < m * m; idx += blockDim.x) {
    double s = 0.0;
    for (int sl = 0; sl < splits; ++sl) s += R[(size_t)sl * m * m + idx];
    R[idx] = s;
    Ri[idx] = 0.0;
  }
  __syncthreads();
  for (int k = 0; k < m; ++k) {
    if (tid == 0) R[k + k * m] = sqrt(fmax(R[k + k * m], 1e-300));
    __syncthreads();
    const double rkk = R[k + k * m];
    for (int jj = k + 1 + tid; jj < m; jj += blockDim.x) R[k + jj * m] /= rkk;
    __syncthreads();
    const int rem = m - k - 1;
    for (int idx = tid; idx < rem * rem; idx += blockDim.x) {
      const int i = k + 1 + idx % rem, jj = k + 1 + idx / rem;
      if (i <= jj) R[i + jj * m] -= R[k + i * m] * R[k + jj * m];
    }
    __syncthreads();
  }
  // inverse of the upper-triangular R, column by column: Ri[:, c] solves R x = e_c (thread c owns column c)
  for (int c = tid; c < m; c += blockDim.x) {
    for (int i = c; i >= 0; --i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int l = i + 1; l <= c; ++l) s -= R[i + l * m] * Ri[l + c * m];
      Ri[i + c * m] = s / R[i + i * m];
    }
  }
}

}  // namespace

size_t stein_workspace_doubles(int n, int m) {
  // factors 5 n m + second X buffer n m + Gram slices 32 m^2 + Rinv m^2 (+ alignment)
  return (size_t)6 * n * m + (size_t)33 * m * m + 64;
}

// Eigenvalues il..iu (0-based, inclusive) into d_w (batch, m) and eigenvectors of T into d_x (batch, n, m
// column-major = (m, n) rows-are-modes).  d_ws: stein_workspace_doubles(n, m) * batch doubles.
int stein_batched(sc_ctx* ctx, int n, int batch, const double* d_tri_ws, const TriLayout& TL, int il, int iu,
                  double* d_w, long long stride_w, double* d_x, long long stride_x, double* d_ws,
                  GemmDesc* d_descs /* 2 * batch */) {
  hipStream_t st = ctx->stream;
  const int m = iu - il + 1;
  PhaseTimer t_sturm(ctx, "sturm", st), t_stein(ctx, "stein", st), t_qr(ctx, "cholqr", st);
  t_sturm.start();
  hipLaunchKernelGGL(k_sturm_range, dim3((unsigned)m, (unsigned)batch), dim3(64), 0, st, d_tri_ws, TL, il, m, d_w, stride_w);
  t_sturm.stop();
  if (!d_x) {
    t_sturm.finish();
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
  }
  const long long stride_ws = (long long)stein_workspace_doubles(n, m);
  double* fac = d_ws;
  t_stein.start();
  hipLaunchKernelGGL(k_stein, dim3((unsigned)((m + 63) / 64), (unsigned)batch), dim3(64), 0, st, d_tri_ws, TL, m,
                     d_w, stride_w, fac, stride_ws, d_x, stride_x);
  t_stein.stop();
  t_qr.start();
  // CholQR2
  const int splits = std::max(1, std::min(32, n / 512));
  std::vector<GemmDesc> h(2 * (size_t)batch);
  double* x_cur = d_x;
  for (int round = 0; round < 2; ++round) {
    for (int b = 0; b < batch; ++b) {
      double* ws = d_ws + (size_t)b * stride_ws;
      double* x2 = ws + (size_t)5 * n * m;
      double* gram = ws + (size_t)6 * n * m;
      double* rinv = gram + (size_t)32 * m * m;
      const double* xin = (round == 0 ? d_x + (size_t)b * stride_x : x2);
      double* xout = (round == 0 ? x2 : d_x + (size_t)b * stride_x);
      GemmDesc G{};
      G.a = xin; G.sa_i = n; G.sa_k = 1;
      G.b = xin; G.sb_k = 1; G.sb_j = n;
      G.c = gram; G.ldc = m; G.m = m; G.n = m; G.k = n;
      G.alpha = 1.0; G.beta = 0.0; G.split_stride = (long long)m * m;
      h[b] = G;
      GemmDesc U{};
      U.a = xin; U.sa_i = 1; U.sa_k = n;
      U.b = rinv; U.sb_k = 1; U.sb_j = m;
      U.c = xout; U.ldc = n; U.m = n; U.n = m; U.k = m;
      U.alpha = 1.0; U.beta = 0.0;
      h[batch + b] = U;
    }
    SC_TRY(sc_stage_upload(ctx, d_descs, h.data(), h.size() * sizeof(GemmDesc)));
    SC_TRY(launch_gemm_f64(ctx, d_descs, batch, m, m, kGemmTile, splits, false, false, kGemmAkBk));
    {
      double* gram0 = d_ws + (size_t)6 * n * m;
      hipLaunchKernelGGL(k_chol_inv, dim3((unsigned)batch), dim3(256), 0, st, gram0, stride_ws, m, splits,
                         gram0 + (size_t)32 * m * m, stride_ws);
    }
    SC_TRY(launch_gemm_f64(ctx, d_descs + batch, batch, n, m, kGemmTile, 1, false, false, kGemmAmBk));
    (void)x_cur;
  }
  t_qr.stop();
  t_sturm.finish(); t_stein.finish(); t_qr.finish();
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}
