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

// n m rounded up to a multiple of four doubles: the 32-byte backward records start at 2 nm4 and every matrix' share of the
// workspace is a multiple of 32 bytes, so the 16- and 32-byte accesses are aligned whatever the parity of n and m (ADVICE
// round 5: with n m odd the records of batch members >= 1 sat on 8-byte boundaries, which gfx9 tolerates by splitting)
__host__ __device__ inline size_t stein_nm4(int n, int m) { return ((size_t)n * m + 3) / 4 * 4; }

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
  // Sturm count WITHOUT a division in the chain (round 5).  The ratio form q_i = (d_i - x) - e_{i-1}^2 / q_{i-1} that
  // dstebz uses is a chain of n IEEE divisions -- ~150 cycles each on this chip: 4 ms per count at n = 24 000, 68 ms for
  // the 16 counts an eigenvalue needs.  The product form p_i = a_i p_{i-1} - b_i p_{i-2} (a_i = (d_i - x) s,
  // b_i = (e_{i-1} s)^2, q_i = p_i / p_{i-1}) has ONE fused multiply-add in the chain (b_i p_{i-2} is known a step
  // earlier); the eigenvalues below x are the sign changes of the sequence, as Wilkinson's bisection counted them.  The
  // rows are scaled by s = 1 / (Gershgorin radius) so that |a_i| <= 2, b_i <= 1: a sequence grows by at most 3^4 between
  // two re-normalisations (every 4 rows, by the exponent of the larger of the last two members).  An exact zero takes
  // the sign opposite to its predecessor -- dstebz's q = -pivmin.  What the product form gives up against the ratio form:
  // rows whose entries AND distance to the shift are below ~1e-60 of the matrix' norm shrink the sequence into underflow
  // and then count as decoupled zeros -- shifts that small are 44 orders below the eps |T| to which a tridiagonal matrix
  // that came out of an orthogonal reduction is known at all (tools/models/sturm_product_model.py, tests/test_sturm_model.py).
  // The rows reach the chain through LDS: lane l of the wave loads row 64 c + l of a 64-row chunk c (two coalesced loads
  // per chunk and lane instead of 2 x 64 loads of one address each), a group of four chunks ahead of the one in work; a
  // chunk in work is written to LDS once (16 bytes per lane) and every step reads its row as ONE broadcast ds_read_b128.
  // (History of the 68 ms this kernel took at n = 24 000: every lane loading every row in chunks of 16, the count waited for
  // memory 1 500 times -- the division-free chain alone: 64 ms; rows from the lanes by v_readlane, four per step, with the
  // wait states they need and a scalar branch per step for the tail: 42 ms.)
  const double sc = 1.0 / fmax(span + 2.0 * emax, 1e-300);
  const double tiny_p = 0x1p-900;
  const int nchunk = (n + 63) / 64;
  typedef double d2s __attribute__((ext_vector_type(2)));
  __shared__ d2s rows[2][64];
  auto load_rows = [&](int c, double& as, double& bs) {
    const int i = min(c * 64 + lane, n - 1);
    const double es = i > 0 ? e[i - 1] * sc : 0.0;
    as = d[i] * sc;
    bs = es * es;
  };
  auto hi32 = [](double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) >> 32); };
  auto count_below = [&](double x) {
    const double xs = x * sc;
    unsigned cnt = 0;
    double p2 = 1.0, p1 = 1.0;           // p_{-1} (any: b_0 = 0), p_0 = 1
    // one step: row (a, b) -> the next member of the sequence.  An exact zero (rare: a branch for the whole wave) takes a
    // tiny value of the sign opposite to its predecessor, so that a decoupled block behind it (b = 0) starts afresh
#define STURM_STEP(r)                                                                  \
    {                                                                                  \
      double p0 = fma((r)[0] - xs, p1, -((r)[1] * p2));                                \
      if (__builtin_expect(__any(p0 == 0.0), 0))                                       \
        if (p0 == 0.0) p0 = -copysign(tiny_p, p1);                                     \
      cnt += (hi32(p0) ^ hi32(p1)) >> 31;                                              \
      p2 = p1;                                                                         \
      p1 = p0;                                                                         \
    }
#define STURM_RENORM()                                                                 \
    {                                                                                  \
      int ex;                                                                          \
      (void)frexp(fmax(fabs(p1), fabs(p2)), &ex);                                      \
      p1 = ldexp(p1, -ex);                                                             \
      p2 = ldexp(p2, -ex);                                                             \
    }
    // groups of four chunks: the rows of group g + 1 are requested at the top of group g and used a group (256 rows) later
    // -- inside ONE loop body, so that the compiler's wait in front of their first use is the only one
    const int ngroup = (nchunk + 3) / 4;
    double ca[4], cb[4], na[4], nb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) load_rows(min(t, nchunk - 1), ca[t], cb[t]);
    for (int g = 0; g < ngroup; ++g) {
#pragma unroll
      for (int t = 0; t < 4; ++t) load_rows(min(4 * (g + 1) + t, nchunk - 1), na[t], nb[t]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 4 * g + t;
        const int lim = min(64, n - c * 64);      // (<= 0 behind the last chunk)
        if (lim <= 0) break;
        rows[t & 1][lane] = d2s{ca[t], cb[t]};
        __builtin_amdgcn_wave_barrier();
        if (lim == 64) {
          // (the rows of the next eight steps are read while these eight run: one exposed LDS latency per step otherwise)
          d2s r8[8], n8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) r8[q] = rows[t & 1][q];
#pragma unroll
          for (int blk = 0; blk < 8; ++blk) {
            if (blk < 7) {
#pragma unroll
              for (int q = 0; q < 8; ++q) n8[q] = rows[t & 1][8 * (blk + 1) + q];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              STURM_STEP(r8[q])
              if ((q & 3) == 3) STURM_RENORM()
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) r8[q] = n8[q];
          }
        } else {
          for (int u = 0; u < lim; ++u) {
            const d2s r = rows[t & 1][u];
            STURM_STEP(r)
            if ((u & 3) == 3) STURM_RENORM()
          }
          STURM_RENORM()
        }
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) { ca[t] = na[t]; cb[t] = nb[t]; }
    }
#undef STURM_STEP
#undef STURM_RENORM
    return (int)cnt;
  };
  double a = lo, b = hi;   // invariant: count_below(a) <= k < count_below(b)
  // Stop at a bracket of eps |T| (dstebz's abstol): narrower brackets cost counts and, next to an exactly
  // zero eigenvalue of a decoupled block, drive the shift into the range where the product-form count flushes rows to
  // sign changes (tools/models/sturm_product_model.py) -- the value returned is the same to eps |T| either way
  const double tol = 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi));
  for (int round = 0; round < 16; ++round) {
    const double width = b - a;
    if (width <= tol) break;
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
    if (mid <= a || mid >= b || b - a <= tol) break;
    if (count_below(mid) > k) b = mid; else a = mid;
  }
  if (lane == 0) w_all[(size_t)blockIdx.y * stride_w + j] = 0.5 * (a + b);
}

__device__ __forceinline__ double hash_unit(unsigned a, unsigned b) {
  unsigned long long x = ((unsigned long long)a << 32) ^ (b * 0x9E3779B97F4A7C15ull) ^ 0xD1B54A32D192ED03ull;
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return ((double)(x >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
}

// Workspace per matrix: forward records F[row][vector] = {multiplier, 1.0 if rows k, k + 1 were swapped} (16 bytes) and
// backward records B[row][vector] = {1 / pivot, u1 / pivot, u2 / pivot, -} (32 bytes): 6 n m doubles.
// The passes over the factors are bound by memory LATENCY -- 106 vectors are two waves on the whole chip, every row of a
// pass needs its record, and a wave can have 63 loads in flight: the records are requested a chunk of rows at a time and the
// chunk is as long as that limit allows (28 rows forward, 20 backward; round 4: four arrays of doubles in chunks of 8 rows,
// 3 000 exposed latencies per pass instead of 860 / 1 200: 66 -> 30 ms at n = 24 000).  The back substitution multiplies
// by the reciprocal pivot that the factorisation stored: no division in its chain.
typedef double d2v __attribute__((ext_vector_type(2)));
typedef double d4v __attribute__((ext_vector_type(4)));
constexpr int kSteinFw = 28, kSteinBw = 20, kSteinFac = 24;

__global__ __launch_bounds__(64) void k_stein(const double* __restrict__ tri_all, TriLayout TL, int m,
                                              const double* __restrict__ w_all, long long stride_w,
                                              double* __restrict__ fac_all, long long stride_fac,
                                              double* __restrict__ x_all, long long stride_x) {
  const double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  const double* d = tri + TL.d;
  const double* e = tri + TL.e;
  const int n = TL.n;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  // scale of T (the lanes strided over the rows, as in k_sturm_range: one lane walking all n rows waits for memory in
  // every one of them) and a tiny separation of (numerically) coincident eigenvalues, as dstein does
  double tnorm = 0.0;
  for (int i = threadIdx.x; i < n; i += 64)
    tnorm = fmax(tnorm, fabs(d[i]) + (i < n - 1 ? fabs(e[i]) : 0.0) + (i > 0 ? fabs(e[i - 1]) : 0.0));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tnorm = fmax(tnorm, __shfl_xor(tnorm, off));
  if (tnorm == 0.0) tnorm = 1.0;
  if (j >= m) return;
  const double* w = w_all + (size_t)blockIdx.y * stride_w;
  double* fac = fac_all + (size_t)blockIdx.y * stride_fac;
  d2v* F = reinterpret_cast<d2v*>(fac);                              // [n][m]
  d4v* B = reinterpret_cast<d4v*>(fac + 2 * stein_nm4(n, m));          // [n][m]
  double* X = x_all + (size_t)blockIdx.y * stride_x + (size_t)j * n;   // column j of X (n x m, ld n)

  const double tiny = kEpsS * tnorm;
  double lam = w[j];
  {
    // shift members of a run of (near-)equal eigenvalues apart by multiples of 10 eps |T|
    int run = 0;
    for (int q = j - 1; q >= 0 && fabs(w[q] - w[q + 1]) <= 10.0 * tiny; --q) ++run;
    lam += run * 10.0 * tiny;
  }
#define FAT(k) F[(size_t)(k) * m + j]
#define BAT(k) B[(size_t)(k) * m + j]
  // LU factorisation of T - lam I with partial pivoting (row k against row k+1)
  double p = d[0] - lam, q = n > 1 ? e[0] : 0.0, r = 0.0;
  for (int k0 = 0; k0 < n - 1; k0 += kSteinFac) {
    double ec[kSteinFac + 1], dc[kSteinFac];   // e[k0 .. k0 + C], d[k0 + 1 .. k0 + C], requested together
#pragma unroll
    for (int u = 0; u < kSteinFac + 1; ++u) ec[u] = e[min(k0 + u, n - 2)];
#pragma unroll
    for (int u = 0; u < kSteinFac; ++u) dc[u] = d[min(k0 + 1 + u, n - 1)];
#pragma unroll
    for (int u = 0; u < kSteinFac; ++u) {
      const int k = k0 + u;
      if (k < n - 1) {
        const double sub = ec[u];
        const double dn = dc[u] - lam;
        const double en = (k + 2 < n) ? ec[u + 1] : 0.0;
        if (fabs(sub) > fabs(p)) {          // swap: pivot row is (sub, dn, en)
          const double mult = p / sub;
          const double inv = 1.0 / sub;
          FAT(k) = d2v{mult, 1.0};
          BAT(k) = d4v{inv, dn * inv, en * inv, 0.0};
          p = q - mult * dn;
          q = r - mult * en;
          r = 0.0;
        } else {
          if (fabs(p) < tiny) p = copysign(tiny, p == 0.0 ? 1.0 : p);
          const double inv = 1.0 / p;
          const double mult = sub * inv;
          FAT(k) = d2v{mult, 0.0};
          BAT(k) = d4v{inv, q * inv, r * inv, 0.0};
          p = dn - mult * q;
          q = en - mult * r;
          r = 0.0;
        }
      }
    }
  }
  if (fabs(p) < tiny) p = copysign(tiny, p == 0.0 ? 1.0 : p);
  BAT(n - 1) = d4v{1.0 / p, 0.0, 0.0, 0.0};

  for (int i = 0; i < n; ++i) X[i] = hash_unit((unsigned)j + 1u, (unsigned)i + 1u);
  for (int iter = 0; iter < 4; ++iter) {
    // forward: apply the row interchanges and multipliers.  The running entry stays in a register
    {
      double xk = X[0];
      for (int k0 = 0; k0 < n - 1; k0 += kSteinFw) {
        double xn[kSteinFw];
        d2v f[kSteinFw];
#pragma unroll
        for (int u = 0; u < kSteinFw; ++u) {
          const int k = min(k0 + u, n - 2);
          xn[u] = X[k + 1];
          f[u] = FAT(k);
        }
#pragma unroll
        for (int u = 0; u < kSteinFw; ++u) {
          if (k0 + u < n - 1) {
            double xv = xn[u];
            if (f[u][1] != 0.0) { const double t = xk; xk = xv; xv = t; }
            xv -= f[u][0] * xk;
            X[k0 + u] = xk;
            xk = xv;
          }
        }
      }
      X[n - 1] = xk;
    }
    // back substitution with the three (scaled) diagonals of U
    double x1 = 0.0, x2 = 0.0, nrm = 0.0;
    for (int k0 = n - 1; k0 >= 0; k0 -= kSteinBw) {
      double xr[kSteinBw];
      d4v bb[kSteinBw];
#pragma unroll
      for (int u = 0; u < kSteinBw; ++u) {
        const int k = max(k0 - u, 0);
        xr[u] = X[k];
        bb[u] = BAT(k);
      }
#pragma unroll
      for (int u = 0; u < kSteinBw; ++u) {
        if (k0 - u >= 0) {
          const double xv = xr[u] * bb[u][0] - bb[u][1] * x1 - bb[u][2] * x2;
          X[k0 - u] = xv;
          x2 = x1;
          x1 = xv;
          nrm = fmax(nrm, fabs(xv));
        }
      }
    }
    // rescale (max-norm) to stay in range; final 2-normalisation is done by the QR step
    // (the rows in chunks of 32 requested together: a plain loop waits for memory in every row)
    const double s = nrm > 0.0 ? 1.0 / nrm : 1.0;
    double ss = 0.0;
    for (int k0 = 0; k0 < n; k0 += 32) {
      double xc[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) xc[u] = X[min(k0 + u, n - 1)];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        if (k0 + u < n) {
          const double xv = xc[u] * s;
          X[k0 + u] = xv;
          ss += xv * xv;
        }
      }
    }
    if (iter == 3) {
      const double inv = 1.0 / sqrt(ss);
      for (int k0 = 0; k0 < n; k0 += 32) {
        double xc[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) xc[u] = X[min(k0 + u, n - 1)];
#pragma unroll
        for (int u = 0; u < 32; ++u)
          if (k0 + u < n) X[k0 + u] = xc[u] * inv;
      }
    }
  }
#undef FAT
#undef BAT
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
  // factors 6 nm4 (16-byte forward + 32-byte backward records) + second X buffer nm4 + Gram slices 32 m^2 + Rinv m^2
  return (7 * stein_nm4(n, m) + (size_t)33 * m * m + 64 + 3) / 4 * 4;
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
      double* x2 = ws + 6 * stein_nm4(n, m);
      double* gram = ws + 7 * stein_nm4(n, m);
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
      double* gram0 = d_ws + 7 * stein_nm4(n, m);
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
