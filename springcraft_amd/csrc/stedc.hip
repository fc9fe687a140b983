// Batched divide & conquer eigensolver for symmetric tridiagonal matrices (stage K6) — the role LAPACK
// dstedc plays inside np.linalg.eigh (reference call site: nma.py:61).  Cuppen's method with the
// Gu/Eisenstat stabilisation:
//   tear T into 2^D leaves (rank-one modifications at the cuts), solve the leaves by implicit QL,
//   then merge level by level:  D + rho z z^T  ->  deflation, secular equation, eigenvector update.
// Everything runs on the device; sizes that depend on the data (number of non-deflated poles K) stay
// in device memory and are picked up by the next kernel / by the GEMM descriptors.
//
// Per merge (all nodes of a level and all matrices of the batch in the same launches):
//   k_dc_zero_offdiag   make the child eigenvector blocks a clean block-diagonal
//   k_dc_setup          z vector, merge-sort of the two child spectra, deflation scan (dlaed2 logic)
//   k_dc_rotate         apply the deflation Givens rotations to the eigenvector columns
//   k_dc_secular        one wave per root: origin selection + bisection on the bit pattern of the
//                       offset tau (<= 64 evaluations, robust for any pole spacing)
//   k_dc_zhat           Gu/Eisenstat weights from the computed roots (orthogonality to working precision)
//   k_dc_vectors        eigenvectors of the rank-one problem, normalised (K x K matrix U)
//   k_dc_finalize       merged ascending order: destinations of new + deflated columns, eigenvalues
//   k_dc_copy_deflated  move deflated eigenvector columns
//   MFMA GEMM x 2       Q_new[top rows, dest] = Q_old[top rows, src_top] * U[k_top, :]  and the same for the bottom
//                       rows: blockdiag(Q1, Q2) is half zeros, each half-GEMM gathers only the columns that are
//                       non-zero in its rows (gemm_f64.hip, gather on both operands' k axis, scatter on C's columns)
#include <algorithm>
#include <vector>

#include "eigh_internal.h"

namespace {

constexpr double kEps = 2.220446049250313e-16;

__device__ __forceinline__ int* iptr(double* ws, long long off) { return reinterpret_cast<int*>(ws + off); }
__device__ __forceinline__ const int* iptr(const double* ws, long long off) {
  return reinterpret_cast<const int*>(ws + off);
}

__device__ __forceinline__ double wave_sum_all(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_prod_all(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v *= __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_max_all(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

// ---- scaling + tearing -------------------------------------------------------------------------------------
__global__ void k_dc_prepare(const double* __restrict__ tri_all, TriLayout TL, double* __restrict__ dc_all,
                             DcLayout DL, const DcNode* __restrict__ nodes, int n_nodes) {
  __shared__ double red[16];
  const double* tri = tri_all + (size_t)blockIdx.x * TL.slab;
  double* ws = dc_all + (size_t)blockIdx.x * DL.slab;
  const int n = DL.n, tid = threadIdx.x;
  double mx = 0.0;
  for (int i = tid; i < n; i += blockDim.x) {
    mx = fmax(mx, fabs(tri[TL.d + i]));
    if (i < n - 1) mx = fmax(mx, fabs(tri[TL.e + i]));
  }
  mx = wave_max_all(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mx = fmax(mx, red[w]);
  const double nrm = (mx > 0.0 && mx == mx) ? mx : 1.0;
  const double inv = 1.0 / nrm;
  // entries below 1e-150 of the norm are flushed to zero: they cannot change a double-precision result, and kept they
  // would drag the merges into denormal arithmetic (z^2, vector norms) -- a graded tridiagonal matrix of that kind comes
  // out of exactly rank-deficient input such as ones(n, n)
  for (int i = tid; i < n; i += blockDim.x) {
    const double dv = tri[TL.d + i] * inv;
    const double ev = (i < n - 1) ? tri[TL.e + i] * inv : 0.0;
    ws[DL.dd + i] = fabs(dv) < 1e-150 ? 0.0 : dv;
    ws[DL.ee + i] = fabs(ev) < 1e-150 ? 0.0 : ev;
  }
  if (tid == 0) ws[DL.scale] = nrm;
  __syncthreads();
  // rank-one tearing at every cut (dlaed0): both neighbours of the cut lose |e|
  for (int g = tid; g < n_nodes; g += blockDim.x) {
    const int mid = nodes[g].mid;
    const double r = fabs(ws[DL.ee + mid - 1]);
    ws[DL.dd + mid - 1] -= r;
    ws[DL.dd + mid] -= r;
  }
}

// ---- leaves: implicit QL with Wilkinson shift, one wave per leaf -------------------------------------------------
// Lane r owns row r of the leaf's eigenvector matrix; the scalar recurrences are evaluated redundantly by
// every lane (wave-uniform control flow), so no cross-lane traffic is needed.
template <int LEAF>
__global__ __launch_bounds__(64) void k_dc_leaves(double* __restrict__ dc_all, DcLayout DL,
                                                  const DcNode* __restrict__ leaves,
                                                  double* __restrict__ q_all, long long stride_q,
                                                  int* __restrict__ fail_flag) {
  __shared__ double Z[LEAF][LEAF + 1];
  __shared__ double d[LEAF], e[LEAF];
  double* ws = dc_all + (size_t)blockIdx.y * DL.slab;
  double* Q = q_all + (size_t)blockIdx.y * stride_q;
  const int lo = leaves[blockIdx.x].lo, hi = leaves[blockIdx.x].hi;
  const int s = hi - lo, n = DL.n;
  const int lane = threadIdx.x;
  if (lane < s) {
    for (int c = 0; c < s; ++c) Z[lane][c] = (c == lane) ? 1.0 : 0.0;
    d[lane] = ws[DL.dd + lo + lane];
    e[lane] = (lane < s - 1) ? ws[DL.ee + lo + lane] : 0.0;
  }
  __syncthreads();
  for (int l = 0; l < s; ++l) {
    int iter = 0;
    int m;
    do {
      for (m = l; m < s - 1; ++m) {
        const double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= kEps * dd) break;
      }
      if (m != l) {
        if (++iter > 80) {
          if (lane == 0) atomicExch(fail_flag, 1);
          break;
        }
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + copysign(r, g));
        double sn = 1.0, cs = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = sn * e[i];
          const double b = cs * e[i];
          r = hypot(f, g);
          __syncthreads();
          e[i + 1] = r;
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            __syncthreads();
            break;
          }
          sn = f / r;
          cs = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * sn + 2.0 * cs * b;
          p = sn * r;
          d[i + 1] = g + p;
          g = cs * r - b;
          if (lane < s) {
            f = Z[lane][i + 1];
            Z[lane][i + 1] = sn * Z[lane][i] + cs * f;
            Z[lane][i] = cs * Z[lane][i] - sn * f;
          }
          __syncthreads();
        }
        if (r == 0.0 && i >= l) continue;
        __syncthreads();
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
        __syncthreads();
      }
    } while (m != l);
  }
  __syncthreads();
  // ascending selection sort (wave-uniform), swapping eigenvector columns
  for (int a = 0; a < s - 1; ++a) {
    int k = a;
    double best = d[a];
    for (int b = a + 1; b < s; ++b)
      if (d[b] < best) { best = d[b]; k = b; }
    __syncthreads();
    if (k != a) {
      const double tmp = d[a];
      __syncthreads();
      d[k] = tmp;
      d[a] = best;
      if (lane < s) {
        const double t2 = Z[lane][a];
        Z[lane][a] = Z[lane][k];
        Z[lane][k] = t2;
      }
    }
    __syncthreads();
  }
  if (lane < s) {
    ws[DL.w0 + lo + lane] = d[lane];
    for (int c = 0; c < s; ++c) Q[(size_t)(lo + c) * n + lo + lane] = Z[lane][c];
  }
}

// ---- zero the off-diagonal blocks of a node's eigenvector block ---------------------------------------------------
__global__ void k_dc_zero_offdiag(const DcNode* __restrict__ nodes, double* __restrict__ q_all,
                                  long long stride_q, int n) {
  const DcNode nd = nodes[blockIdx.y];
  double* Q = q_all + (size_t)blockIdx.z * stride_q;
  const int n1 = nd.mid - nd.lo, n2 = nd.hi - nd.mid;
  const long long total = 2LL * n1 * n2;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    if (idx < (long long)n1 * n2) {
      // rows [mid,hi) x cols [lo,mid)
      const int r = nd.mid + (int)(idx % n2), c = nd.lo + (int)(idx / n2);
      Q[(size_t)c * n + r] = 0.0;
    } else {
      const long long k = idx - (long long)n1 * n2;
      const int r = nd.lo + (int)(k % n1), c = nd.mid + (int)(k / n1);
      Q[(size_t)c * n + r] = 0.0;
    }
  }
}

// ---- merge setup: z, sort, deflation ---------------------------------------------------------------------------------
// One block per (node, matrix).  Sorted copies live in LDS when they fit (use_lds), else in global scratch.
__global__ __launch_bounds__(1024) void k_dc_setup(double* __restrict__ dc_all, DcLayout DL,
                                                   const DcNode* __restrict__ nodes, int nodes_in_level,
                                                   const double* __restrict__ q_all, long long stride_q,
                                                   long long w_old_off, GemmDesc* __restrict__ descs,
                                                   int use_lds, double* __restrict__ gscratch,
                                                   long long gscratch_stride) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red[16];
  __shared__ double s_tol;
  const int g = blockIdx.x, b = blockIdx.y;
  const DcNode nd = nodes[g];
  double* ws = dc_all + (size_t)b * DL.slab;
  const double* Q = q_all + (size_t)b * stride_q;
  const int n = DL.n, lo = nd.lo, mid = nd.mid, hi = nd.hi;
  const int N = hi - lo, n1 = mid - lo;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const double* wold = ws + w_old_off;

  double* sd;
  double* sz;
  int* scol;
  unsigned char* smix;   // 1: the column has been mixed across the two halves by a deflation rotation
  unsigned char* sflag;  // 1: negligible z component (dlaed2 type-1 deflation)
  if (use_lds) {
    sd = lds;
    sz = lds + N;
    scol = reinterpret_cast<int*>(lds + 2 * N);
    smix = reinterpret_cast<unsigned char*>(scol + N);
  } else {
    double* base = gscratch + (size_t)b * gscratch_stride + 3LL * lo;
    sd = base;
    sz = base + N;
    scol = reinterpret_cast<int*>(base + 2 * N);
    smix = reinterpret_cast<unsigned char*>(scol + N);
  }
  sflag = smix + N;

  const double rho_raw = ws[DL.ee + mid - 1];
  const double sgn = rho_raw < 0.0 ? -1.0 : 1.0;
  const double rho = 2.0 * fabs(rho_raw);
  const double isq2 = 0.7071067811865475244;

  double zmax = 0.0, dmax = 0.0;
  for (int j = tid; j < N; j += nthr) {
    const int col = lo + j;
    const double zj = (j < n1 ? Q[(size_t)col * n + (mid - 1)] : sgn * Q[(size_t)col * n + mid]) * isq2;
    ws[DL.z + col] = zj;
    zmax = fmax(zmax, fabs(zj));
    dmax = fmax(dmax, fabs(wold[col]));
  }
  zmax = wave_max_all(zmax);
  dmax = wave_max_all(dmax);
  if ((tid & 63) == 0) red[tid >> 6] = fmax(zmax, dmax);
  __syncthreads();
  if (tid == 0) {
    double mx = 0.0;
    for (int w = 0; w < (nthr >> 6); ++w) mx = fmax(mx, red[w]);
    s_tol = 8.0 * kEps * mx;
  }
  __syncthreads();
  // stable merge of the two ascending child spectra by rank
  for (int j = tid; j < N; j += nthr) {
    const double v = wold[lo + j];
    int rank;
    if (j < n1) {
      int a = 0, c = N - n1;  // count of second-half elements < v
      while (a < c) {
        const int h = (a + c) >> 1;
        if (wold[mid + h] < v) a = h + 1; else c = h;
      }
      rank = j + a;
    } else {
      int a = 0, c = n1;  // count of first-half elements <= v
      while (a < c) {
        const int h = (a + c) >> 1;
        if (wold[lo + h] <= v) a = h + 1; else c = h;
      }
      rank = (j - n1) + a;
    }
    sd[rank] = v;
    sz[rank] = ws[DL.z + lo + j];
    scol[rank] = lo + j;
    smix[rank] = 0;
  }
  __syncthreads();

  // ---- the deflation scan of dlaed2.  Its order matters only along CHAINS of poles that are rotated into one another
  // (type 2: the rotation changes the pole the next comparison sees).  Round 6: all threads evaluate the type-1 flags and
  // the closeness test of every pair of neighbouring survivors on the values as sorted; the pairs that pass are the seeds
  // of chains, which ONE thread follows in order (rotation, next survivor, test on the changed pole, ... until a test
  // fails) -- work proportional to the rotations, not to the node --; then all threads write the lists from prefix counts.
  // One thread walking the whole node (until round 6) was 0.3 us per element: 0.48 ms for the top level of one n = 1536
  // matrix, 1.8 of the 10.3 ms of a single N = 512 solve.  Deflated entries are listed in index order (the scan listed a
  // rotated-away pole when its partner was reached; k_dc_finalize ranks them, the order is free).
  // sflag: bit 0 = type 1, bit 1 = seed, bit 2 = rotated away (type 2).
  const double tol = s_tol;
  int* src = iptr(ws, DL.src);
  int* rot_a = iptr(ws, DL.rot_a);
  int* rot_b = iptr(ws, DL.rot_b);
  int* cnt = iptr(ws, DL.cnt);
  int* ktop_src = iptr(ws, DL.ktop_src);
  int* ktop_k = iptr(ws, DL.ktop_k);
  int* kbot_src = iptr(ws, DL.kbot_src);
  int* kbot_k = iptr(ws, DL.kbot_k);
  __shared__ int wsum[3][16];
  __shared__ int s_nrot;
  const int L = (N + nthr - 1) / nthr;
  const int j0 = min(tid * L, N), j1 = min(j0 + L, N);

  for (int j = tid; j < N; j += nthr) sflag[j] = !(rho * fabs(sz[j]) > tol) ? 1 : 0;
  if (tid == 0) s_nrot = 0;
  __syncthreads();
  int nseed = 0;
  for (int jj = j0; jj < j1; ++jj) {
    if (sflag[jj] & 1) continue;
    int pj = jj - 1;
    while (pj >= 0 && (sflag[pj] & 1)) --pj;
    if (pj < 0) continue;
    double s = sz[pj], c = sz[jj];
    const double tau = hypot(c, s);
    const double t = sd[jj] - sd[pj];
    c /= tau;
    s /= tau;
    if (fabs(t * c * s) <= tol) { sflag[jj] |= 2; ++nseed; }
  }
  const int any_rot = __syncthreads_or(nseed);
  if (any_rot) {
    // the seeds in ascending order, into the (not yet written) kbot_k list of the node
    int is = nseed;
    for (int off = 1; off < 64; off <<= 1) {
      const int us = __shfl_up(is, off);
      if ((tid & 63) >= off) is += us;
    }
    if ((tid & 63) == 63) wsum[0][tid >> 6] = is;
    __syncthreads();
    int bs = 0, total = 0;
    for (int w = 0; w < (nthr >> 6); ++w) {
      if (w < (tid >> 6)) bs += wsum[0][w];
      total += wsum[0][w];
    }
    int pos = bs + is - nseed;
    for (int jj = j0; jj < j1; ++jj)
      if (sflag[jj] & 2) kbot_k[lo + pos++] = jj;
    __syncthreads();
    if (tid == 0) {
      int nrot = 0, done_upto = -1;
      for (int si = 0; si < total; ++si) {
        int jj = kbot_k[lo + si];
        if (jj <= done_upto) continue;        // reached by the chain of an earlier seed
        int pj = jj - 1;
        while (pj >= 0 && (sflag[pj] & 5)) --pj;   // (the survivor in front of a seed that no chain reached is unchanged)
        while (true) {
          double s = sz[pj], c = sz[jj];
          const double tau = hypot(c, s);
          const double t = sd[jj] - sd[pj];
          c /= tau;
          s /= tau;
          done_upto = jj;
          if (!(fabs(t * c * s) <= tol)) break;    // (never at a seed itself: same values as in the test above)
          // close poles: rotate z_pj into z_jj (type-2 deflation)
          sz[jj] = tau;
          sz[pj] = 0.0;
          rot_a[lo + nrot] = scol[pj];
          rot_b[lo + nrot] = scol[jj];
          ws[DL.rot_c + lo + nrot] = c;
          ws[DL.rot_s + lo + nrot] = s;
          ++nrot;
          if (((scol[pj] < mid) != (scol[jj] < mid)) || smix[pj] || smix[jj]) { smix[jj] = 1; smix[pj] = 1; }
          const double tt = sd[pj] * c * c + sd[jj] * s * s;
          sd[jj] = sd[pj] * s * s + sd[jj] * c * c;
          sd[pj] = tt;
          sflag[pj] |= 4;
          pj = jj;
          ++jj;
          while (jj < N && (sflag[jj] & 1)) ++jj;
          if (jj >= N) break;
        }
      }
      s_nrot = nrot;
    }
    __syncthreads();
  }
  {
    // lists from prefix counts: survivors that stay (status 0), of those the ones with entries in the top / bottom rows
    int ns = 0, nt = 0, nb = 0;
    for (int j = j0; j < j1; ++j)
      if (!(sflag[j] & 5)) {
        const bool top = scol[j] < mid;
        ++ns;
        nt += (top || smix[j]) ? 1 : 0;
        nb += (!top || smix[j]) ? 1 : 0;
      }
    int is = ns, it = nt, ib = nb;   // inclusive sums over the wave, then over the block
    for (int off = 1; off < 64; off <<= 1) {
      const int us = __shfl_up(is, off), ut = __shfl_up(it, off), ub = __shfl_up(ib, off);
      if ((tid & 63) >= off) { is += us; it += ut; ib += ub; }
    }
    __syncthreads();   // (wsum[0] was read above)
    if ((tid & 63) == 63) { wsum[0][tid >> 6] = is; wsum[1][tid >> 6] = it; wsum[2][tid >> 6] = ib; }
    __syncthreads();
    int bs = 0, bt = 0, bb = 0, ts = 0, tt = 0, tb = 0;
    for (int w = 0; w < (nthr >> 6); ++w) {
      if (w < (tid >> 6)) { bs += wsum[0][w]; bt += wsum[1][w]; bb += wsum[2][w]; }
      ts += wsum[0][w];
      tt += wsum[1][w];
      tb += wsum[2][w];
    }
    int K = bs + is - ns, K1 = bt + it - nt, K3 = bb + ib - nb;
    int ndef = j0 - K;
    for (int j = j0; j < j1; ++j) {
      if (sflag[j] & 5) {
        ws[DL.ddef + hi - 1 - ndef] = sd[j];
        src[hi - 1 - ndef] = scol[j];
        ++ndef;
      } else {
        ws[DL.dl + lo + K] = sd[j];
        ws[DL.zz + lo + K] = sz[j];
        src[lo + K] = scol[j];
        // blockdiag(Q1, Q2): a column from the first child is zero in the bottom rows and vice versa, unless a
        // deflation rotation mixed it (dlaed2's column types 1 / 2 / 3): the two half-GEMMs only take what is non-zero
        const bool top = scol[j] < mid;
        if (top || smix[j]) { ktop_src[lo + K1] = scol[j]; ktop_k[lo + K1] = K; ++K1; }
        if (!top || smix[j]) { kbot_src[lo + K3] = scol[j]; kbot_k[lo + K3] = K; ++K3; }
        ++K;
      }
    }
    if (tid == 0) {
      cnt[2 * g] = ts;
      cnt[2 * g + 1] = s_nrot;
      GemmDesc& Dt = descs[((size_t)b * nodes_in_level + g) * 2];      // top rows
      Dt.n = ts;
      Dt.k = tt;
      GemmDesc& Db = descs[((size_t)b * nodes_in_level + g) * 2 + 1];  // bottom rows
      Db.n = ts;
      Db.k = tb;
    }
  }
}

// ---- apply the deflation rotations (rows in parallel, rotations in list order) ----------------------------------------
__global__ void k_dc_rotate(double* __restrict__ dc_all, DcLayout DL, const DcNode* __restrict__ nodes,
                            double* __restrict__ q_all, long long stride_q) {
  const int g = blockIdx.y, b = blockIdx.z;
  const DcNode nd = nodes[g];
  double* ws = dc_all + (size_t)b * DL.slab;
  const int nrot = iptr(ws, DL.cnt)[2 * g + 1];
  if (nrot == 0) return;
  double* Q = q_all + (size_t)b * stride_q;
  const int n = DL.n;
  const int row = nd.lo + blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nd.hi) return;
  const int* rot_a = iptr(ws, DL.rot_a) + nd.lo;
  const int* rot_b = iptr(ws, DL.rot_b) + nd.lo;
  for (int q = 0; q < nrot; ++q) {
    const double c = ws[DL.rot_c + nd.lo + q], s = ws[DL.rot_s + nd.lo + q];
    double* pa = Q + (size_t)rot_a[q] * n + row;
    double* pb = Q + (size_t)rot_b[q] * n + row;
    const double qa = *pa, qb = *pb;
    *pa = c * qa - s * qb;   // q_p' = c q_p - s q_n
    *pb = s * qa + c * qb;   // q_n' = s q_p + c q_n
  }
}

// ---- secular equation: one wave per root --------------------------------------------------------------------------------
// f(lambda) = 1/rho + sum_i zz_i^2 / (dl_i - lambda).  Root j lies in (dl_j, dl_j+1) (last: (dl_K-1, dl_K-1 + rho)).
// lambda is represented as dl_org + tau with org the nearer pole, so that dl_i - lambda = (dl_i - dl_org) - tau
// keeps full relative accuracy.  tau is found by bisection on the IEEE bit pattern of |tau| (monotone map),
// which pins it to one ulp in <= 64 steps regardless of its magnitude.
__global__ __launch_bounds__(1024) void k_dc_secular(double* __restrict__ dc_all, DcLayout DL,
                                                     const DcNode* __restrict__ nodes, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int g = blockIdx.y, b = blockIdx.z;
  const DcNode nd = nodes[g];
  double* ws = dc_all + (size_t)b * DL.slab;
  const int K = iptr(ws, DL.cnt)[2 * g];
  const int waves = blockDim.x >> 6;
  if ((int)blockIdx.x * waves >= K) return;
  const int lo = nd.lo;
  const double* dlg = ws + DL.dl + lo;
  const double* zzg = ws + DL.zz + lo;
  const double* dl;
  const double* z2;
  if (use_lds) {
    double* ldl = lds;
    double* lz2 = lds + K;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
      ldl[i] = dlg[i];
      const double z = zzg[i];
      lz2[i] = z * z;
    }
    __syncthreads();
    dl = ldl;
    z2 = lz2;
  } else {
    dl = dlg;
    z2 = nullptr;
  }
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * waves + (threadIdx.x >> 6);
  if (j >= K) return;
  const double rho = 2.0 * fabs(ws[DL.ee + nd.mid - 1]);
  const double rinv = 1.0 / rho;

  // f and f' = sum zz_i^2 / (dl_i - lambda)^2 in one pass
  auto fsum2 = [&](double dorg, double tau, double& fp) -> double {
    double s = 0.0, sp = 0.0;
    if (z2) {
      for (int i = lane; i < K; i += 64) {
        const double r = 1.0 / ((dl[i] - dorg) - tau);
        const double zr = z2[i] * r;
        s += zr;
        sp += zr * r;
      }
    } else {
      for (int i = lane; i < K; i += 64) {
        const double z = zzg[i];
        const double r = 1.0 / ((dl[i] - dorg) - tau);
        const double zr = z * z * r;
        s += zr;
        sp += zr * r;
      }
    }
    fp = wave_sum_all(sp);
    return rinv + wave_sum_all(s);
  };
  auto fsum = [&](double dorg, double tau) -> double {
    double fp;
    return fsum2(dorg, tau, fp);
  };

  int org;
  bool positive;      // tau > 0 (origin is the lower pole) or tau < 0 (origin is the upper pole)
  double t_hi;        // bracket for |tau|: (0, t_hi]
  double t_start = 0.0;   // first iterate (0: the bit midpoint of the bracket)
  double zj2 = 0.0, zk2 = 0.0;   // squares of the z components of the two poles next to an interior root
  if (j == K - 1) {
    org = K - 1;
    positive = true;
    double s2 = 0.0;
    if (z2) { for (int i = lane; i < K; i += 64) s2 += z2[i]; }
    else { for (int i = lane; i < K; i += 64) s2 += zzg[i] * zzg[i]; }
    s2 = wave_sum_all(s2);
    t_hi = rho * s2 * (1.0 + 8.0 * kEps) + 1e-300;
  } else {
    const double gap = dl[j + 1] - dl[j];
    const double half = 0.5 * gap;
    const double fm = fsum(dl[j], half);
    if (fm > 0.0) { org = j; positive = true; } else { org = j + 1; positive = false; }
    t_hi = half;
    // First iterate as LAPACK's dlaed4 takes it: the two poles next to the root kept exactly, the rest of f frozen at
    // the midpoint (c = f(mid) without the two terms) -- a quadratic in tau.  It only replaces the bit midpoint of the
    // bracket as the starting point (two to three evaluations fewer); every safeguard below is unchanged.
    {
      zj2 = z2 ? z2[j] : zzg[j] * zzg[j];
      zk2 = z2 ? z2[j + 1] : zzg[j + 1] * zzg[j + 1];
      const double c = fm + (zj2 - zk2) / half;
      double t0;
      if (positive) {
        const double a = c * gap + zj2 + zk2, b = zj2 * gap;
        const double disc = sqrt(fabs(a * a - 4.0 * b * c));
        t0 = a > 0.0 ? 2.0 * b / (a + disc) : (a - disc) / (2.0 * c);
      } else {
        const double a = c * gap - zj2 - zk2, b = zk2 * gap;
        const double disc = sqrt(fabs(a * a + 4.0 * b * c));
        t0 = -(a < 0.0 ? 2.0 * b / (a - disc) : -(a + disc) / (2.0 * c));   // |tau|, tau <= 0
      }
      if (t0 == t0 && t0 > 0.0 && t0 < t_hi) t_start = t0;
    }
  }
  const double dorg = dl[org];
  // Root of f in t = |tau| on (0, t_hi], kept in a bracket [lo_b, hi_b] of IEEE bit patterns (monotone map, so the
  // bracket can always be halved whatever the magnitude of the root).  Iteration: Newton in the variable 1/t, i.e.
  // the model f ~ c -+ s / t through the origin pole (exact when that pole dominates, quadratic otherwise),
  // t_new = f' t^2 / (f' t + f) with f' taken with respect to t; a step that leaves the bracket, or is not finite,
  // is replaced by the bit midpoint.  Once the step is below a few ulps the far side is probed so that the bracket
  // closes to neighbouring doubles: same end state as plain bisection, in ~10 evaluations instead of ~62.
  unsigned long long lo_b = 0ull, hi_b = (unsigned long long)__double_as_longlong(t_hi);
  unsigned long long t_b = hi_b - ((hi_b - lo_b) >> 1);
  if (t_start > 0.0) t_b = (unsigned long long)__double_as_longlong(t_start);
  unsigned long long probe_dist = 4ull;
  double t_accept = 0.0;
  for (int it = 0; it < 90 && hi_b - lo_b > 1ull; ++it) {
    const double t = __longlong_as_double((long long)t_b);
    double fp;
    const double f = fsum2(dorg, positive ? t : -t, fp);
    // positive: f increasing in t, f(0+) = -inf;  negative: f decreasing in t, f(0+) = +inf
    const bool go_up = positive ? (f < 0.0) : (f > 0.0);
    if (go_up) lo_b = t_b; else hi_b = t_b;
    if (hi_b - lo_b <= 1ull) break;
    const unsigned long long mid_b = lo_b + ((hi_b - lo_b) >> 1);
    double tn;
    bool model_step = false;
    // (while the origin pole's own term z^2 / t dominates f the two-pole formulas below cancel catastrophically -- LAPACK
    // never gets there because it always starts from the quadratic's root -- and Newton in 1 / t is exact for that term)
    if (j < K - 1 && fabs(f) * t < 0.25 * (positive ? zj2 : zk2)) {
      model_step = true;
      // LAPACK dlaed4's step for an interior root: f modelled by the two poles next to the root, exact in value and
      // slope at the iterate (c + zj^2-ish / (d_j - x) + ... , "middle way"); only f and f' (both summed above) and the
      // two pole terms are needed.  Measured: 21.6 evaluations per root with the Newton step in 1/t alone (and a first
      // iterate that was wrong for roots next to their upper pole), see DESIGN section 7 for the count with this step.
      const double tau = positive ? t : -t;
      const double dj = (dl[j] - dorg) - tau, dk = (dl[j + 1] - dorg) - tau;   // d_j - lambda < 0 < d_{j+1} - lambda
      const double dw = fp;
      double c;
      if (positive) {
        const double q = zj2 / (dj * dj);
        c = f - dk * dw - (dl[j] - dl[j + 1]) * q;
      } else {
        const double q = zk2 / (dk * dk);
        c = f - dj * dw - (dl[j + 1] - dl[j]) * q;
      }
      const double a = (dj + dk) * f - dj * dk * dw;
      const double b = dj * dk * f;
      double eta;
      if (c == 0.0) {
        eta = a != 0.0 ? b / a : -f / dw;
      } else {
        const double disc = sqrt(fabs(a * a - 4.0 * b * c));
        eta = a <= 0.0 ? (a - disc) / (2.0 * c) : 2.0 * b / (a + disc);
      }
      if (f * eta >= 0.0) eta = -f / dw;             // (rounding: the model's root lies on the wrong side: a Newton step)
      const double tau_n = tau + eta;
      tn = positive ? tau_n : -tau_n;
    } else {
      const double fpt = positive ? fp : -fp;         // df/dt
      const double den = fpt * t + f;
      tn = fpt * t * t / den;                         // Newton in 1/t (the last root: no pole above it)
    }
    unsigned long long n_b = mid_b;
    if (tn == tn && tn > 0.0 && tn < 1.7e308) {
      const unsigned long long c_b = (unsigned long long)__double_as_longlong(tn);
      if (c_b > lo_b && c_b < hi_b) {
        n_b = c_b;
        const unsigned long long step = c_b > t_b ? c_b - t_b : t_b - c_b;
#ifndef DC_STRICT_BRACKET
        if (step <= 2ull || (model_step && fabs(tn - t) <= 1e-12 * t)) {
          // the iteration moves by at most two ulps of t: t is the root to that accuracy (the step is quadratically small
          // long before it is that small) -- or the two-pole model moves it by less than 1e-12 t: its next step would be
          // of the order of the square of that times t / (distance to the nearest pole outside the model), i.e. below an
          // ulp even with third poles as close as 1e-10 of the spectrum (round 3 accepted at 1e-9, which poles that are
          // close but not deflated could turn into an error far above an ulp: ADVICE round 3; glued Wilkinson matrices in
          // tests/test_eigh_gpu.py).  Taken as it is -- closing the bracket to neighbouring doubles from the far
          // side and comparing |f| at its two ends cost four more evaluations of ~ nine (LAPACK's dlaed4 also stops on a
          // bound for |f|, not on a closed bracket); the eigenvectors are built from the roots by the Gu-Eisenstat
          // weights, which make them orthogonal for whatever roots they are given.
          t_accept = tn;
          break;
        }
#endif
        if (step <= 4ull) {
          // converged from one side: look a few ulps beyond (4, 32, 256, ...) on the side the bracket is still wide
          const unsigned long long far = go_up ? c_b + probe_dist : (c_b > probe_dist ? c_b - probe_dist : 0ull);
          n_b = (far > lo_b && far < hi_b) ? far : mid_b;
          probe_dist *= 8ull;
        }
      }
    }
    t_b = n_b;
  }
  double t_best = __longlong_as_double((long long)hi_b);
  if (t_accept > 0.0) {
    t_best = t_accept;
  } else if (lo_b > 0ull) {
    const double t_lo = __longlong_as_double((long long)lo_b);
    const double f_lo = fabs(fsum(dorg, positive ? t_lo : -t_lo));
    const double f_hi = fabs(fsum(dorg, positive ? t_best : -t_best));
    if (f_lo < f_hi) t_best = t_lo;
  }
  if (lane == 0) {
    const double tau = positive ? t_best : -t_best;
    iptr(ws, DL.org)[lo + j] = org;
    ws[DL.tauv + lo + j] = tau;
    ws[DL.lam + lo + j] = dorg + tau;
  }
}

// ---- Gu/Eisenstat weights: zhat_i^2 = prod_j (lambda_j - dl_i) / prod_{j != i} (dl_j - dl_i) ---------------------------
__global__ __launch_bounds__(256) void k_dc_zhat(double* __restrict__ dc_all, DcLayout DL,
                                                 const DcNode* __restrict__ nodes) {
  const int g = blockIdx.y, b = blockIdx.z;
  const DcNode nd = nodes[g];
  double* ws = dc_all + (size_t)b * DL.slab;
  const int K = iptr(ws, DL.cnt)[2 * g];
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= K) return;
  const int lane = threadIdx.x & 63, lo = nd.lo;
  const double* dl = ws + DL.dl + lo;
  const int* org = iptr(ws, DL.org) + lo;
  const double* tauv = ws + DL.tauv + lo;
  const double di = dl[i];
  double prod = 1.0;
  for (int j = lane; j < K; j += 64) {
    const double num = (di - dl[org[j]]) - tauv[j];   // dl_i - lambda_j
    prod *= (j == i) ? num : num / (di - dl[j]);
  }
  prod = wave_prod_all(prod);
  if (lane == 0) ws[DL.zhat + lo + i] = copysign(sqrt(fabs(prod)), ws[DL.zz + lo + i]);
}

// ---- eigenvectors of D + rho z z^T: u_j[i] = zhat_i / (dl_i - lambda_j), normalised; block per root --------------------------
__global__ __launch_bounds__(256) void k_dc_vectors(double* __restrict__ dc_all, DcLayout DL,
                                                    const DcNode* __restrict__ nodes,
                                                    double* __restrict__ u_all, long long stride_u) {
  __shared__ double red[4];
  const int g = blockIdx.y, b = blockIdx.z;
  const DcNode nd = nodes[g];
  double* ws = dc_all + (size_t)b * DL.slab;
  const int K = iptr(ws, DL.cnt)[2 * g];
  const int j = blockIdx.x;
  if (j >= K) return;
  const int lo = nd.lo, n = DL.n;
  const double* dl = ws + DL.dl + lo;
  const double* zh = ws + DL.zhat + lo;
  const double dorg = dl[iptr(ws, DL.org)[lo + j]];
  const double tau = ws[DL.tauv + lo + j];
  double* U = u_all + (size_t)b * stride_u + (size_t)(lo + j) * n + lo;
  double ss = 0.0;
  for (int i = threadIdx.x; i < K; i += blockDim.x) {
    const double u = zh[i] / ((dl[i] - dorg) - tau);
    U[i] = u;
    ss += u * u;
  }
  ss = wave_sum_all(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const double inv = 1.0 / sqrt((red[0] + red[1]) + (red[2] + red[3]));
  for (int i = threadIdx.x; i < K; i += blockDim.x) U[i] *= inv;
}

// ---- merged ascending order ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dc_finalize(double* __restrict__ dc_all, DcLayout DL,
                                                     const DcNode* __restrict__ nodes, long long w_new_off) {
  const int g = blockIdx.y, b = blockIdx.z;
  const DcNode nd = nodes[g];
  double* ws = dc_all + (size_t)b * DL.slab;
  const int K = iptr(ws, DL.cnt)[2 * g];
  const int lo = nd.lo, hi = nd.hi, N = hi - lo, ndef = N - K;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N) return;
  const double* lam = ws + DL.lam + lo;
  int* dest = iptr(ws, DL.dest);
  double* wnew = ws + w_new_off;
  auto defval = [&](int e) { return ws[DL.ddef + hi - 1 - e]; };
  if (idx < K) {
    const double v = lam[idx];
    int c = 0;
    for (int e = 0; e < ndef; ++e) c += defval(e) < v;
    dest[lo + idx] = lo + idx + c;
    wnew[lo + idx + c] = v;
  } else {
    const int e = idx - K;
    const double v = defval(e);
    int a = 0, c = K;  // roots <= v
    while (a < c) {
      const int h = (a + c) >> 1;
      if (lam[h] <= v) a = h + 1; else c = h;
    }
    int r = a;
    for (int f = 0; f < ndef; ++f) {
      const double u = defval(f);
      r += (u < v) || (u == v && f < e);
    }
    dest[lo + idx] = lo + r;
    wnew[lo + r] = v;
  }
}

__global__ __launch_bounds__(256) void k_dc_copy_deflated(double* __restrict__ dc_all, DcLayout DL,
                                                          const DcNode* __restrict__ nodes,
                                                          const double* __restrict__ q_old_all,
                                                          double* __restrict__ q_new_all,
                                                          long long stride_q) {
  const int g = blockIdx.y, b = blockIdx.z;
  const DcNode nd = nodes[g];
  double* ws = dc_all + (size_t)b * DL.slab;
  const int K = iptr(ws, DL.cnt)[2 * g];
  const int lo = nd.lo, hi = nd.hi, N = hi - lo;
  const int e = blockIdx.x;
  if (e >= N - K) return;
  const int n = DL.n;
  const int scol = iptr(ws, DL.src)[hi - 1 - e];
  const int dcol = iptr(ws, DL.dest)[lo + K + e];
  const double* qs = q_old_all + (size_t)b * stride_q + (size_t)scol * n + lo;
  double* qd = q_new_all + (size_t)b * stride_q + (size_t)dcol * n + lo;
  for (int r = threadIdx.x; r < N; r += blockDim.x) qd[r] = qs[r];
}

__global__ void k_dc_unscale(const double* __restrict__ dc_all, DcLayout DL, long long w_off,
                             double* __restrict__ w_out, long long stride_w) {
  const double* ws = dc_all + (size_t)blockIdx.y * DL.slab;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < DL.n) w_out[(size_t)blockIdx.y * stride_w + i] = ws[w_off + i] * ws[DL.scale];
}

// host-side tree ------------------------------------------------------------------------------------------------------
struct Tree {
  std::vector<DcNode> leaves;
  std::vector<std::vector<DcNode>> levels;  // levels[0] merges leaves, back() is the root
};

Tree build_tree(int n, int leaf_max) {
  int depth = 0;
  while (((n + (1 << depth) - 1) >> depth) > leaf_max) ++depth;
  // segments after `depth` halvings
  std::vector<std::vector<DcNode>> by_depth(depth + 1);
  by_depth[0].push_back(DcNode{0, n / 2, n});
  for (int dlev = 0; dlev < depth; ++dlev) {
    for (const DcNode& nd : by_depth[dlev]) {
      by_depth[dlev + 1].push_back(DcNode{nd.lo, nd.lo + (nd.mid - nd.lo) / 2, nd.mid});
      by_depth[dlev + 1].push_back(DcNode{nd.mid, nd.mid + (nd.hi - nd.mid) / 2, nd.hi});
    }
  }
  Tree t;
  t.leaves = by_depth[depth];
  for (int dlev = depth - 1; dlev >= 0; --dlev) t.levels.push_back(by_depth[dlev]);
  return t;
}

constexpr int kLeafMax = 32;
constexpr int kLdsCapSetup = 7000;    // 22 B per element
constexpr int kLdsCapSecular = 9800;  // 16 B per pole

}  // namespace

int dc_max_nodes(int n, int leaf_max) {
  Tree t = build_tree(n, leaf_max);
  size_t total = t.leaves.size();
  for (auto& l : t.levels) total += l.size();
  return (int)total;
}

size_t dc_slab_doubles(int n, DcLayout* out) {
  DcLayout L{};
  L.n = n;
  L.leaf_max = kLeafMax;
  long long off = 0;
  const long long nn = ((long long)n + 7) / 8 * 8;
  auto take = [&](long long cnt) { long long o = off; off += (cnt + 7) / 8 * 8; return o; };
  L.dd = take(nn); L.ee = take(nn); L.w0 = take(nn); L.w1 = take(nn); L.z = take(nn);
  L.dl = take(nn); L.zz = take(nn); L.ddef = take(nn); L.lam = take(nn); L.tauv = take(nn);
  L.zhat = take(nn); L.rot_c = take(nn); L.rot_s = take(nn); L.scale = take(8);
  L.src = take(nn); L.org = take(nn); L.dest = take(nn); L.rot_a = take(nn); L.rot_b = take(nn);
  L.ktop_src = take(nn); L.ktop_k = take(nn); L.kbot_src = take(nn); L.kbot_k = take(nn);
  L.cnt = take(2LL * dc_max_nodes(n, kLeafMax) + 8);
  L.slab = off;
  if (out) *out = L;
  return (size_t)off;
}

// 2 m n k summed over the merge records (profiling: the sizes are decided on the device by k_dc_setup)
__global__ __launch_bounds__(1024) void k_dc_desc_flops(const GemmDesc* __restrict__ d, int count, double* __restrict__ out) {
  __shared__ double part[16];
  double a = 0.0;
  for (int r = threadIdx.x; r < count; r += 1024) a += 2.0 * (double)d[r].m * (double)d[r].n * (double)d[r].k;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += part[w];
    *out = t;
  }
}

int stedc_batched(sc_ctx* ctx, int n, int batch, const double* d_tri_ws, const TriLayout& TL,
                  double* d_dc_ws, const DcLayout& DL, double* d_w, long long stride_w, double* d_q_out,
                  double* d_q_tmp, double* d_u, long long stride_q, GemmDesc* d_merge_descs) {
  hipStream_t st = ctx->stream;
  Tree tree = build_tree(n, kLeafMax);
  const int nlev = (int)tree.levels.size();

  // node tables -> device (leaves first, then levels)
  std::vector<DcNode> flat(tree.leaves);
  std::vector<size_t> lev_off(nlev);
  for (int l = 0; l < nlev; ++l) {
    lev_off[l] = flat.size();
    flat.insert(flat.end(), tree.levels[l].begin(), tree.levels[l].end());
  }
  const size_t n_internal = flat.size() - tree.leaves.size();
  // node tables + fail flag + the sorted copies of oversize merges: one cached allocation of the context
  const size_t node_bytes = align_up(flat.size() * sizeof(DcNode), 256);
  const size_t big_bytes = n > kLdsCapSetup ? (size_t)batch * 3 * n * sizeof(double) : 0;
  SC_TRY(sc_reserve_dc_aux(ctx, node_bytes + 256 + big_bytes));
  DcNode* d_nodes = reinterpret_cast<DcNode*>(ctx->dc_aux);
  // (a QL failure goes to the context's deferred status word, read at the next synchronising call)
  int* d_fail = reinterpret_cast<int*>(ctx->d_status + 1);
  double* d_big = big_bytes ? reinterpret_cast<double*>(reinterpret_cast<char*>(d_nodes) + node_bytes + 256) : nullptr;
  SC_TRY(sc_stage_upload(ctx, d_nodes, flat.data(), flat.size() * sizeof(DcNode)));

  // eigenvector ping-pong: level l writes X[l+1]; X[nlev] must be d_q_out
  auto qbuf = [&](int stage) { return ((nlev - stage) % 2 == 0) ? d_q_out : d_q_tmp; };

  // merge GEMM descriptors for all levels (static parts), sizes are filled in by k_dc_setup
  std::vector<GemmDesc> h_descs;
  std::vector<size_t> desc_off(nlev);
  for (int l = 0; l < nlev; ++l) {
    desc_off[l] = h_descs.size();
    const auto& nodes = tree.levels[l];
    const double* q_old = qbuf(l);
    double* q_new = qbuf(l + 1);
    for (int b = 0; b < batch; ++b)
      for (size_t g = 0; g < nodes.size(); ++g) {
        const DcNode& nd = nodes[g];
        auto ip = [&](long long off) { return reinterpret_cast<const int*>(d_dc_ws + (size_t)b * DL.slab + off) + nd.lo; };
        for (int half = 0; half < 2; ++half) {
          const int r0 = half == 0 ? nd.lo : nd.mid;
          GemmDesc D{};
          D.a = q_old + (size_t)b * stride_q + r0;
          D.sa_i = 1; D.sa_k = n;
          D.a_kidx = ip(half == 0 ? DL.ktop_src : DL.kbot_src);
          D.b = d_u + (size_t)b * stride_q + (size_t)nd.lo * n + nd.lo;
          D.sb_k = 1; D.sb_j = n;
          D.b_kidx = ip(half == 0 ? DL.ktop_k : DL.kbot_k);
          D.c = q_new + (size_t)b * stride_q + r0;
          D.ldc = n;
          D.c_jidx = ip(DL.dest);
          D.m = half == 0 ? nd.mid - nd.lo : nd.hi - nd.mid; D.n = 0; D.k = 0;
          D.alpha = 1.0; D.beta = 0.0;
          h_descs.push_back(D);
        }
      }
  }
  if (!h_descs.empty()) SC_TRY(sc_stage_upload(ctx, d_merge_descs, h_descs.data(), h_descs.size() * sizeof(GemmDesc)));

  hipLaunchKernelGGL(k_dc_prepare, dim3((unsigned)batch), dim3(1024), 0, st, d_tri_ws, TL, d_dc_ws, DL,
                     d_nodes + tree.leaves.size(), (int)n_internal);
  hipLaunchKernelGGL((k_dc_leaves<kLeafMax>), dim3((unsigned)tree.leaves.size(), (unsigned)batch), dim3(64),
                     0, st, d_dc_ws, DL, d_nodes, qbuf(0), stride_q, d_fail);

  PhaseTimer t_gemm(ctx, "dc_gemm", st);
  for (int l = 0; l < nlev; ++l) {
    const auto& nodes = tree.levels[l];
    const int G = (int)nodes.size();
    const DcNode* dn = d_nodes + lev_off[l];
    int maxN = 0;
    for (auto& nd : nodes) maxN = std::max(maxN, nd.hi - nd.lo);
    double* q_old = qbuf(l);
    double* q_new = qbuf(l + 1);
    const long long w_old = (l % 2 == 0) ? DL.w0 : DL.w1;
    const long long w_new = (l % 2 == 0) ? DL.w1 : DL.w0;
    GemmDesc* descs = d_merge_descs + desc_off[l];

    hipLaunchKernelGGL(k_dc_zero_offdiag,
                       dim3((unsigned)std::min(1024, (maxN * maxN / 2 + 255) / 256 + 1), (unsigned)G,
                            (unsigned)batch),
                       dim3(256), 0, st, dn, q_old, stride_q, n);
    {
      const int use_lds = maxN <= kLdsCapSetup ? 1 : 0;
      const size_t lds = use_lds ? (size_t)maxN * 22 + 32 : 0;
      const int threads = maxN >= 1024 ? 1024 : (maxN >= 256 ? 256 : 64);
      hipLaunchKernelGGL(k_dc_setup, dim3((unsigned)G, (unsigned)batch), dim3(threads), lds, st, d_dc_ws, DL,
                         dn, G, q_old, stride_q, w_old, descs, use_lds, d_big, (long long)3 * n);
    }
    hipLaunchKernelGGL(k_dc_rotate, dim3((unsigned)((maxN + 255) / 256), (unsigned)G, (unsigned)batch),
                       dim3(256), 0, st, d_dc_ws, DL, dn, q_old, stride_q);
    {
      const int use_lds = maxN <= kLdsCapSecular ? 1 : 0;
      const int threads = maxN >= 2048 ? 1024 : 256;
      const int waves = threads / 64;
      const size_t lds = use_lds ? (size_t)maxN * 16 + 16 : 0;
      hipLaunchKernelGGL(k_dc_secular, dim3((unsigned)((maxN + waves - 1) / waves), (unsigned)G, (unsigned)batch),
                         dim3(threads), lds, st, d_dc_ws, DL, dn, use_lds);
    }
    hipLaunchKernelGGL(k_dc_zhat, dim3((unsigned)((maxN + 3) / 4), (unsigned)G, (unsigned)batch), dim3(256), 0,
                       st, d_dc_ws, DL, dn);
    hipLaunchKernelGGL(k_dc_vectors, dim3((unsigned)maxN, (unsigned)G, (unsigned)batch), dim3(256), 0, st,
                       d_dc_ws, DL, dn, d_u, stride_q);
    hipLaunchKernelGGL(k_dc_finalize, dim3((unsigned)((maxN + 255) / 256), (unsigned)G, (unsigned)batch),
                       dim3(256), 0, st, d_dc_ws, DL, dn, w_new);
    hipLaunchKernelGGL(k_dc_copy_deflated, dim3((unsigned)maxN, (unsigned)G, (unsigned)batch), dim3(256), 0, st,
                       d_dc_ws, DL, dn, q_old, q_new, stride_q);
    t_gemm.start();
    SC_TRY(launch_gemm_f64(ctx, descs, 2 * G * batch, (maxN + 1) / 2, maxN, kGemmTile, 1, /*gather=*/true, false,
                           kGemmAmBk));
    t_gemm.stop();
  }
  const long long w_final = (nlev % 2 == 0) ? DL.w0 : DL.w1;
  hipLaunchKernelGGL(k_dc_unscale, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, st,
                     d_dc_ws, DL, w_final, d_w, stride_w);
  SC_HIP(ctx, hipGetLastError());

  t_gemm.finish();
  if (ctx->profiling && !h_descs.empty()) {
    // flops of the merge GEMMs: their K and N depend on the deflation and only exist in the device-side records
    // (profiling runs only; one small launch and an 8-byte copy)
    double* d_flops = reinterpret_cast<double*>(reinterpret_cast<char*>(d_nodes) + node_bytes);   // (the 256 spare bytes)
    hipLaunchKernelGGL(k_dc_desc_flops, dim3(1), dim3(1024), 0, st, d_merge_descs, (int)h_descs.size(), d_flops);
    double h_flops = 0.0;
    SC_HIP(ctx, hipMemcpyAsync(&h_flops, d_flops, sizeof(double), hipMemcpyDeviceToHost, st));
    SC_HIP(ctx, hipStreamSynchronize(st));
    bool found = false;
    for (auto& ph : ctx->phases)
      if (ph.first == "dc_gemm_gflop") { ph.second = h_flops * 1e-9; found = true; }
    if (!found) ctx->phases.emplace_back("dc_gemm_gflop", h_flops * 1e-9);
  }
  return SC_OK;
}
