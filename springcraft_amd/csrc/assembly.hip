// Contact scan + Kirchhoff / Hessian assembly kernels for gfx950 (MI355X).
//
// Replaces, on device, the reference's  _prepare_values_for_interaction_matrix
// (interaction.py:114-190), compute_kirchhoff (interaction.py:14-54) and compute_hessian
// (interaction.py:57-111).  HBM-write-bound O(N^2) work: 8 B per ordered atom pair for the
// Kirchhoff matrix, 72 B per ordered pair for the Hessian.
//
// This translation unit is compiled with -ffp-contract=off: the contact predicate must
// reproduce the reference's float64 arithmetic bit for bit,
//     sq = (dx*dx + dy*dy) + dz*dz ;  contact = sq <= cutoff**2      (interaction.py:165-166)
// with separately rounded products (no FMA).
#include "common.h"

namespace {

struct FFDev {
  int kind;
  int has_cutoff;
  double cutoff_sq;
  // SC_FF_TABULATED (device pointers)
  int n_bins;
  const double* edges_sq;
  const float* tab;            // [3][20][20][n_bins]: bonded, intra-chain, inter-chain
  const int* atom_type;
  const int* chain;
  const unsigned char* bonded_next;
};

// One structure of a ragged / decorated batch (sc_batch_plan, api.hip): its own size, force field and patches; the
// coordinates / weights / matrices of all structures sit in three buffers the launch gets as base pointers.
struct AsmItem {
  long long atom_off;   // first atom of this structure in the packed coordinate / weight buffers
  int n;                // atoms
  int ld;               // order of the matrix slot (>= dim * n; the slot is padded, see k_pad_fill)
  FFDev ff;
  PatchDev patch;       // empty tables (all zero) when the structure has no patches but another one of the batch has
};

// TabulatedForceField.force_constant (forcefield.py:515-533) from the type tables
__device__ __forceinline__ double tab_gamma(const FFDev& ff, int i, int j, double d2) {
  int bin = 0;
  if (ff.n_bins > 1) {
    int lo = 0, hi = ff.n_bins;   // np.searchsorted(edges**2, d2, side='left'): number of edges^2 < d2
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (ff.edges_sq[mid] < d2) lo = mid + 1; else hi = mid;
    }
    bin = lo < ff.n_bins ? lo : ff.n_bins - 1;
  }
  const int a = i < j ? i : j, b = i < j ? j : i;
  int which;
  if (b == a + 1 && ff.bonded_next[a]) which = 0;
  else which = (ff.chain[i] == ff.chain[j]) ? 1 : 2;
  const int ti = ff.atom_type[i], tj = ff.atom_type[j];
  return (double)ff.tab[((size_t)(which * 20 + ti) * 20 + tj) * ff.n_bins + bin];
}

__device__ __forceinline__ double ff_gamma_dist(int kind, double d2) {
  if (kind == SC_FF_INVARIANT) return 1.0;               // forcefield.py:284-285
  if (kind == SC_FF_PARAMETER_FREE) return 1.0 / d2;     // forcefield.py:361-362
  // Hinsen, forcefield.py:321-326
  double d = sqrt(d2);
  d = fmax(d, 2.9);
  if (d < 4.0) return d * 8.6e2 - 2.39e3;
  const double dd = d * d;
  return 128e4 / (dd * dd * dd);  // d**-6 * 1.28e6, within 4 ulp of numpy's pow()
}

// Evaluates one ordered pair.  ci/cj are the coordinates; returns contact, fills d2, gamma, disp.
template <bool PATCH>
__device__ __forceinline__ bool pair_eval(int i, int j, double cix, double ciy, double ciz,
                                          double cjx, double cjy, double cjz, const FFDev& ff,
                                          const PatchDev& patch, int row_beg, int row_end,
                                          bool shut_i, double& d2, double& gamma, double& dx,
                                          double& dy, double& dz) {
  dx = cjx - cix;
  dy = cjy - ciy;
  dz = cjz - ciz;
  d2 = (dx * dx + dy * dy) + dz * dz;
  const bool within = !ff.has_cutoff || (d2 <= ff.cutoff_sq);
  bool contact = (i != j) && within;
  gamma = 0.0;
  if (PATCH) {
    double gover = __builtin_nan("");
    if (shut_i || patch.shut[j]) contact = false;  // interaction.py:201-203
    for (int e = row_beg; e < row_end; ++e) {      // pair_off / pair_on, interaction.py:204-213
      if (patch.col[e] == j) {
        contact = patch.flag[e] != 0;
        gover = patch.gam[e];
      }
    }
    if (contact) {
      // PatchedForceField.force_constant, forcefield.py:183-226
      gamma = (patch.mask_gamma && !within) ? 0.0
              : (ff.kind == SC_FF_TABULATED ? tab_gamma(ff, i, j, d2) : ff_gamma_dist(ff.kind, d2));
      if (gover == gover) gamma = gover;
    }
  } else {
    if (contact) gamma = ff.kind == SC_FF_TABULATED ? tab_gamma(ff, i, j, d2) : ff_gamma_dist(ff.kind, d2);
  }
  return contact;
}

// Deterministic block-wide sum (256 threads = 4 waves): shuffle tree inside a wave, then a
// fixed-order sum over the 4 wave results.  Result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double* red /*[4]*/) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- Kirchhoff rows (+ contact counts) ---------------------------------------------------------
// One block = TI consecutive atoms i (matrix rows) x all columns; thread = column j in a 256-wide
// tile.  Row coordinates are wave-uniform registers, column coordinates are read once per tile and
// reused for the TI rows.  Stores are 8 B per lane, contiguous along the row.
template <int TI, bool PATCH, bool ITEMS>
__global__ __launch_bounds__(256) void k_kirchhoff(const double* __restrict__ coord_all, int n_arg,
                                                   FFDev ff_arg, PatchDev patch_arg,
                                                   const double* __restrict__ w_all,
                                                   double* __restrict__ k_all,
                                                   long long* __restrict__ counts_all,
                                                   const AsmItem* __restrict__ items) {
  __shared__ double red[4];
  const size_t b = blockIdx.y;
  int n = n_arg;
  size_t ld = (size_t)n_arg;
  FFDev ff = ff_arg;
  PatchDev patch = patch_arg;
  const double* coord;
  const double* w;
  double* K;
  long long* counts = nullptr;
  if (ITEMS) {
    const AsmItem it = items[b];
    n = it.n;
    ld = (size_t)it.ld;
    ff = it.ff;
    patch = it.patch;
    coord = coord_all + (size_t)it.atom_off * 3;
    w = w_all ? w_all + it.atom_off : nullptr;
    K = k_all ? k_all + b * ld * ld : nullptr;
    counts = counts_all ? counts_all + it.atom_off : nullptr;   // contact counts of all structures back to back
  } else {
    coord = coord_all + b * (size_t)n * 3;
    w = w_all ? w_all + b * (size_t)n : nullptr;
    K = k_all ? k_all + b * (size_t)n * n : nullptr;
    counts = counts_all ? counts_all + b * (size_t)n : nullptr;
  }
  const int i0 = blockIdx.x * TI;
  if (i0 >= n) return;   // (ragged batches: the grid covers the largest structure)

  double cix[TI], ciy[TI], ciz[TI], wi[TI];
  int rbeg[TI], rend[TI];
  bool shut[TI];
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    const int i = min(i0 + t, n - 1);
    cix[t] = coord[3 * (size_t)i + 0];
    ciy[t] = coord[3 * (size_t)i + 1];
    ciz[t] = coord[3 * (size_t)i + 2];
    wi[t] = w ? w[i] : 1.0;
    rbeg[t] = rend[t] = 0;
    shut[t] = false;
    if (PATCH) {
      rbeg[t] = patch.row_ptr[i];
      rend[t] = patch.row_ptr[i + 1];
      shut[t] = patch.shut[i] != 0;
    }
  }
  double rsum[TI];
  long long cnt[TI];
#pragma unroll
  for (int t = 0; t < TI; ++t) { rsum[t] = 0.0; cnt[t] = 0; }

  for (int j = threadIdx.x; j < n; j += 256) {
    const double cjx = coord[3 * (size_t)j + 0];
    const double cjy = coord[3 * (size_t)j + 1];
    const double cjz = coord[3 * (size_t)j + 2];
    const double wj = w ? w[j] : 1.0;
#pragma unroll
    for (int t = 0; t < TI; ++t) {
      const int i = i0 + t;
      if (i < n) {
        double d2, g, dx, dy, dz;
        const bool c = pair_eval<PATCH>(i, j, cix[t], ciy[t], ciz[t], cjx, cjy, cjz, ff, patch,
                                        rbeg[t], rend[t], shut[t], d2, g, dx, dy, dz);
        if (c) { rsum[t] += g; cnt[t] += 1; }
        if (K && j != i) {
          double v = c ? -g : 0.0;                     // interaction.py:50
          if (w) v = v * (wi[t] * wj);                 // gnm.py:104-105
          K[(size_t)i * ld + j] = v;
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    const double s = block_sum_256(rsum[t], red);
    const double c = block_sum_256((double)cnt[t], red);  // exact: counts < 2^53
    const int i = i0 + t;
    if (threadIdx.x == 0 && i < n) {
      if (K) {
        double v = s;                                  // -sum(-gamma), interaction.py:52
        if (w) v = v * (wi[t] * wi[t]);
        K[(size_t)i * ld + i] = v;
      }
      if (counts) counts[i] = (long long)c;
    }
  }
}

// ---- Hessian rows ----------------------------------------------------------------------------------
// One block = TI consecutive atoms i (3*TI matrix rows) x all columns.  Per 256-atom column tile each
// thread evaluates ONE pair (i, j) -> its 3x3 block, parks it in LDS as three 768-double row segments
// and the block then streams those segments out with fully contiguous 8-B-per-lane stores (a lane
// writing its own 3 doubles would issue 24-B-strided partial-line stores).  Diagonal blocks are the
// negated row sums, accumulated in registers and reduced in a fixed order (deterministic).
template <int TI, bool PATCH, bool ITEMS>
__global__ __launch_bounds__(256) void k_hessian(const double* __restrict__ coord_all, int n_arg,
                                                 FFDev ff_arg, PatchDev patch_arg,
                                                 const double* __restrict__ w_all,
                                                 double* __restrict__ h_all,
                                                 const AsmItem* __restrict__ items) {
  __shared__ double tile[2][3][768];
  __shared__ double red[4];
  const size_t b = blockIdx.y;
  int n = n_arg;
  FFDev ff = ff_arg;
  PatchDev patch = patch_arg;
  const double* coord;
  const double* w;
  size_t ld;   // row stride of the matrix slot
  if (ITEMS) {
    const AsmItem it = items[b];
    n = it.n;
    ld = (size_t)it.ld;
    ff = it.ff;
    patch = it.patch;
    coord = coord_all + (size_t)it.atom_off * 3;
    w = w_all ? w_all + it.atom_off : nullptr;
  } else {
    ld = (size_t)n * 3;
    coord = coord_all + b * (size_t)n * 3;
    w = w_all ? w_all + b * (size_t)n : nullptr;
  }
  const size_t n3 = (size_t)n * 3;
  double* H = h_all + b * ld * ld;
  const int i0 = blockIdx.x * TI;
  if (i0 >= n) return;   // (ragged batches: the grid covers the largest structure)
  const int tid = threadIdx.x;

  double cix[TI], ciy[TI], ciz[TI], wi[TI];
  int rbeg[TI], rend[TI];
  bool shut[TI];
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    const int i = min(i0 + t, n - 1);
    cix[t] = coord[3 * (size_t)i + 0];
    ciy[t] = coord[3 * (size_t)i + 1];
    ciz[t] = coord[3 * (size_t)i + 2];
    wi[t] = w ? w[i] : 1.0;
    rbeg[t] = rend[t] = 0;
    shut[t] = false;
    if (PATCH) {
      rbeg[t] = patch.row_ptr[i];
      rend[t] = patch.row_ptr[i + 1];
      shut[t] = patch.shut[i] != 0;
    }
  }
  double acc[TI][9];
#pragma unroll
  for (int t = 0; t < TI; ++t)
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[t][q] = 0.0;

  const int ntiles = (n + 255) / 256;
  int buf = 0;
  for (int tl = 0; tl < ntiles; ++tl) {
    const int j = tl * 256 + tid;
    const bool valid = j < n;
    const int jc = valid ? j : n - 1;
    const double cjx = coord[3 * (size_t)jc + 0];
    const double cjy = coord[3 * (size_t)jc + 1];
    const double cjz = coord[3 * (size_t)jc + 2];
    const double wj = w ? w[jc] : 1.0;
#pragma unroll
    for (int t = 0; t < TI; ++t) {
      const int i = i0 + t;
      if (i >= n) break;  // block-uniform
      double blk[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) blk[q] = 0.0;
      if (valid) {
        double d2, g, dx, dy, dz;
        const bool c = pair_eval<PATCH>(i, j, cix[t], ciy[t], ciz[t], cjx, cjy, cjz, ff, patch,
                                        rbeg[t], rend[t], shut[t], d2, g, dx, dy, dz);
        if (c) {
          // interaction.py:96-101, left to right: ((-g / d2) * disp_a) * disp_b
          const double tt = (-g) / d2;
          const double d[3] = {dx, dy, dz};
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double ta = tt * d[a];
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) {
              const double v = ta * d[bb];
              acc[t][a * 3 + bb] += v;
              blk[a * 3 + bb] = w ? v * (wi[t] * wj) : v;   // anm.py:112-113
            }
          }
        }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) tile[buf][a][3 * tid + bb] = blk[a * 3 + bb];
      __syncthreads();
      const size_t col0 = (size_t)tl * 768;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double* row = H + ((size_t)3 * i + a) * ld;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int c = q * 256 + tid;
          const size_t gc = col0 + c;
          if (gc < n3 && (int)(gc / 3) != i) row[gc] = tile[buf][a][c];
        }
      }
      buf ^= 1;
    }
  }
  // diagonal blocks: H_ii = -sum_j B_ij  (interaction.py:103-104; B_ij == B_ji for symmetric gamma)
#pragma unroll
  for (int t = 0; t < TI; ++t) {
    const int i = i0 + t;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const double s = block_sum_256(acc[t][q], red);
      if (tid == 0 && i < n) {
        double v = -s;
        if (w) v = v * (wi[t] * wi[t]);
        H[((size_t)3 * i + q / 3) * ld + (size_t)3 * i + q % 3] = v;
      }
    }
  }
}

// ---- ordered pair list (np.where order) ------------------------------------------------------------------
// One wave per atom i; 64 candidate columns per step; ballot + prefix popcount gives each contact its
// rank inside the row, so pairs come out sorted by (i, j) without atomics.
// ITEMS: structure blockIdx.y of a batch plan; `offsets` then spans the atoms of ALL structures back to back (one
// exclusive scan), so every structure's rows land behind those of the structure before it, with local atom indices.
template <bool PATCH, bool ITEMS = false>
__global__ __launch_bounds__(256) void k_pair_fill(const double* __restrict__ coord, int n, FFDev ff,
                                                   PatchDev patch,
                                                   const long long* __restrict__ offsets,
                                                   long long* __restrict__ pairs,
                                                   double* __restrict__ sqd,
                                                   const AsmItem* __restrict__ items = nullptr) {
  if (ITEMS) {
    const AsmItem it = items[blockIdx.y];
    n = it.n;
    ff = it.ff;
    patch = it.patch;
    coord += (size_t)it.atom_off * 3;
    offsets += it.atom_off;
  }
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const double cix = coord[3 * (size_t)i + 0], ciy = coord[3 * (size_t)i + 1],
               ciz = coord[3 * (size_t)i + 2];
  int rbeg = 0, rend = 0;
  bool shut = false;
  if (PATCH) { rbeg = patch.row_ptr[i]; rend = patch.row_ptr[i + 1]; shut = patch.shut[i] != 0; }
  long long base = offsets[i];
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    bool c = false;
    double d2 = 0, g, dx, dy, dz;
    if (j < n) {
      const double cjx = coord[3 * (size_t)j + 0], cjy = coord[3 * (size_t)j + 1],
                   cjz = coord[3 * (size_t)j + 2];
      c = pair_eval<PATCH>(i, j, cix, ciy, ciz, cjx, cjy, cjz, ff, patch, rbeg, rend, shut, d2, g,
                           dx, dy, dz);
    }
    const unsigned long long m = __ballot(c);
    if (c) {
      const long long r = base + __popcll(m & ((1ull << lane) - 1ull));
      pairs[2 * r] = i;
      pairs[2 * r + 1] = j;
      if (sqd) sqd[r] = d2;
    }
    base += __popcll(m);
  }
}

// ---- host-callback path: matrices from an explicit pair list + gamma[k] -----------------------------------
__global__ void k_kirchhoff_scatter(int n, const long long* __restrict__ pairs, long long k,
                                    const double* __restrict__ gamma, double* __restrict__ K) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= k) return;
  const long long i = pairs[2 * p], j = pairs[2 * p + 1];
  K[i * n + j] = -gamma[p];  // interaction.py:50
}

// diag_j = -sum_i K[i, j], accumulated sequentially over i exactly like np.sum(axis=0) (interaction.py:52)
__global__ void k_kirchhoff_diag(int n, double* __restrict__ K) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += K[(size_t)i * n + j];
  K[(size_t)j * n + j] = -s;
}

__global__ void k_hessian_scatter(const double* __restrict__ coord, int n,
                                  const long long* __restrict__ pairs, long long k,
                                  const double* __restrict__ gamma, double* __restrict__ H) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= k) return;
  const long long i = pairs[2 * p], j = pairs[2 * p + 1];
  const double dx = coord[3 * j + 0] - coord[3 * i + 0];
  const double dy = coord[3 * j + 1] - coord[3 * i + 1];
  const double dz = coord[3 * j + 2] - coord[3 * i + 2];
  const double d2 = (dx * dx + dy * dy) + dz * dz;
  const double tt = (-gamma[p]) / d2;
  const double d[3] = {dx, dy, dz};
  const size_t n3 = (size_t)n * 3;
  for (int a = 0; a < 3; ++a) {
    const double ta = tt * d[a];
    for (int b = 0; b < 3; ++b) H[(3 * i + a) * n3 + 3 * j + b] = ta * d[b];
  }
}

// H4[i,i] = -sum_j' H4[j', i]: thread = (matrix column c = 3i+b, a); sequential over j' (np.sum axis 0).
__global__ void k_hessian_diag(int n, double* __restrict__ H) {
  const size_t n3 = (size_t)n * 3;
  const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (idx >= n3 * 3) return;
  const size_t c = idx / 3;
  const int a = (int)(idx % 3);
  const size_t i = c / 3;
  double s = 0.0;
  for (int jp = 0; jp < n; ++jp) {
    if ((size_t)jp == i) continue;  // the (i,i) block is still zero in the reference at this point
    s += H[((size_t)3 * jp + a) * n3 + c];
  }
  H[(3 * i + a) * n3 + c] = -s;
}

// ---- host-callback path for a whole batch plan: pairs of all structures back to back, gamma from the host ------------
// Pair p belongs to the structure b with pair_off[b] <= p < pair_off[b + 1] (binary search; count + 1 offsets).
__device__ __forceinline__ int item_of_pair(const long long* __restrict__ pair_off, int count, long long p) {
  int lo = 0, hi = count;   // invariant: pair_off[lo] <= p < pair_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pair_off[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

template <int DIM>
__global__ __launch_bounds__(256) void k_items_scatter(const AsmItem* __restrict__ items, int count,
                                                       const long long* __restrict__ pair_off,
                                                       const long long* __restrict__ pairs,
                                                       const double* __restrict__ gamma,
                                                       const double* __restrict__ coord_all, double* __restrict__ m_all) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= pair_off[count]) return;
  const int b = item_of_pair(pair_off, count, p);
  const AsmItem it = items[b];
  const size_t ld = (size_t)it.ld;
  double* M = m_all + (size_t)b * ld * ld;
  const long long i = pairs[2 * p], j = pairs[2 * p + 1];
  if (DIM == 1) {
    M[(size_t)i * ld + j] = -gamma[p];                         // interaction.py:50
  } else {
    const double* coord = coord_all + (size_t)it.atom_off * 3;
    const double dx = coord[3 * j + 0] - coord[3 * i + 0];
    const double dy = coord[3 * j + 1] - coord[3 * i + 1];
    const double dz = coord[3 * j + 2] - coord[3 * i + 2];
    const double d2 = (dx * dx + dy * dy) + dz * dz;
    const double tt = (-gamma[p]) / d2;                        // interaction.py:96-101
    const double d[3] = {dx, dy, dz};
    for (int a = 0; a < 3; ++a) {
      const double ta = tt * d[a];
      for (int c = 0; c < 3; ++c) M[(size_t)(3 * i + a) * ld + 3 * j + c] = ta * d[c];
    }
  }
}

// diagonal (blocks): minus the sums over the FIRST index, sequential as np.sum(axis=0) (interaction.py:52,103-104);
// grid (ceil(DIM * DIM * max_atoms / 256), count)
template <int DIM>
__global__ __launch_bounds__(256) void k_items_diag(const AsmItem* __restrict__ items, double* __restrict__ m_all) {
  const AsmItem it = items[blockIdx.y];
  const size_t ld = (size_t)it.ld;
  double* M = m_all + (size_t)blockIdx.y * ld * ld;
  const size_t idx = blockIdx.x * (size_t)256 + threadIdx.x;
  if (idx >= (size_t)it.n * DIM * DIM) return;
  const size_t c = idx / DIM;            // matrix column
  const int a = (int)(idx % DIM);        // row inside the block
  const size_t i = c / DIM;              // atom of the column
  double s = 0.0;
  for (int jp = 0; jp < it.n; ++jp) {
    if (DIM == 3 && (size_t)jp == i) continue;   // the (i, i) block is still zero in the reference at this point
    s += M[((size_t)DIM * jp + a) * ld + c];
  }
  M[((size_t)DIM * i + a) * ld + c] = -s;
}

// M *= outer(w, w) with w = repeat(1 / sqrt(m), DIM)  (anm.py:89-94,112-113; gnm.py:85-87,104-105)
__global__ __launch_bounds__(256) void k_items_weight(const AsmItem* __restrict__ items, int dim,
                                                      const double* __restrict__ w_all, double* __restrict__ m_all) {
  const AsmItem it = items[blockIdx.y];
  const size_t ld = (size_t)it.ld;
  const size_t m = (size_t)dim * it.n;
  double* M = m_all + (size_t)blockIdx.y * ld * ld;
  const double* w = w_all + it.atom_off;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < m * m; idx += (size_t)gridDim.x * 256) {
    const size_t r = idx / m, c = idx % m;
    M[r * ld + c] *= w[r / dim] * w[c / dim];
  }
}

// ---- padded slots of a ragged batch ----------------------------------------------------------------------------
// Structures of different sizes share ONE batched eigensolve of order ld: slot b holds diag(M_b, D_b) with M_b the
// structure's m x m matrix (m = dim * atoms) and D_b a diagonal block with entries above every eigenvalue of M_b.  The
// two blocks never mix (the reflectors of the tridiagonalisation have exact zeros in the padded rows, the tridiagonal
// matrix splits exactly at m), so the first m eigenpairs of the slot are those of M_b, eigenvectors in the leading m
// components.  The pad values stay within a factor ~2 of ||M_b||: the solver's tolerances scale with the norm of the
// whole slot, so a huge pad value would cost M_b's small eigenvalues their accuracy.
//   k_slot_bound: sigma_b = 2 max_i sum_j |M_b[i, j]| (Gershgorin: every |eigenvalue| <= the largest absolute row sum)
//   k_pad_fill:   zeros beside and below M_b, pad entry p = sigma_b (1 + (p + 1) / (pad count))   (distinct, ascending)
__global__ __launch_bounds__(256) void k_slot_bound(const double* __restrict__ m_all, const AsmItem* __restrict__ items,
                                                    int dim, unsigned long long* __restrict__ bound_bits) {
  const AsmItem it = items[blockIdx.y];
  const size_t ld = (size_t)it.ld;
  const int m = dim * it.n;
  const double* M = m_all + (size_t)blockIdx.y * ld * ld;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;   // wave-uniform
  double s = 0.0;
  for (int j = lane; j < m; j += 64) s += fabs(M[(size_t)row * ld + j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  // bit patterns of non-negative doubles are ordered like the values (a NaN row sum has the largest pattern: the slot
  // bound becomes NaN, the pad entries NaN, and the solver rejects the matrix as it does for any non-finite input)
  if (lane == 0) atomicMax(bound_bits + blockIdx.y, (unsigned long long)__double_as_longlong(s));
}

__global__ __launch_bounds__(256) void k_pad_fill(double* __restrict__ m_all, const AsmItem* __restrict__ items, int dim,
                                                  const unsigned long long* __restrict__ bound_bits) {
  const AsmItem it = items[blockIdx.y];
  const size_t ld = (size_t)it.ld;
  const int m = dim * it.n;
  const int npad = it.ld - m;
  if (npad <= 0) return;
  double* M = m_all + (size_t)blockIdx.y * ld * ld;
  const double bound = __longlong_as_double((long long)bound_bits[blockIdx.y]);
  const double sigma = bound > 0.0 ? 2.0 * bound : (bound == 0.0 ? 1.0 : bound);
  // region 1: rows [0, m) x columns [m, ld); region 2: rows [m, ld) x all columns
  const size_t r1 = (size_t)m * npad, total = r1 + (size_t)npad * ld;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    size_t r, c;
    if (idx < r1) { r = idx / npad; c = m + idx % npad; }
    else { const size_t k = idx - r1; r = m + k / ld; c = k % ld; }
    M[r * ld + c] = (r == c) ? sigma * (1.0 + (double)(r - m + 1) / (double)npad) : 0.0;
  }
}

// ff.tab, when set, has already been replaced by a descriptor holding DEVICE pointers (api.hip:stage_tab)
FFDev make_ff(const sc_ff_desc& ff) {
  FFDev d{};
  d.kind = ff.kind;
  d.has_cutoff = ff.has_cutoff;
  d.cutoff_sq = ff.cutoff_sq;
  if (ff.kind == SC_FF_TABULATED && ff.tab) {
    d.n_bins = ff.tab->n_bins;
    d.edges_sq = ff.tab->edges_sq;
    d.tab = ff.tab->bonded;   // the three tables are uploaded back to back
    d.atom_type = ff.tab->atom_type;
    d.chain = ff.tab->chain;
    d.bonded_next = ff.tab->bonded_next;
  }
  return d;
}

}  // namespace

int launch_kirchhoff(sc_ctx* ctx, const double* d_coord, int64_t n, int64_t batch,
                     const sc_ff_desc& ff, const PatchDev* patch, const double* d_w, double* d_k,
                     int64_t* d_counts) {
  if (n <= 0 || batch <= 0) return SC_OK;
  constexpr int TI = 4;
  dim3 grid((unsigned)((n + TI - 1) / TI), (unsigned)batch);
  PatchDev p{};
  if (patch) {
    p = *patch;
    hipLaunchKernelGGL((k_kirchhoff<TI, true, false>), grid, dim3(256), 0, ctx->stream, d_coord, (int)n,
                       make_ff(ff), p, d_w, d_k, (long long*)d_counts, (const AsmItem*)nullptr);
  } else {
    hipLaunchKernelGGL((k_kirchhoff<TI, false, false>), grid, dim3(256), 0, ctx->stream, d_coord, (int)n,
                       make_ff(ff), p, d_w, d_k, (long long*)d_counts, (const AsmItem*)nullptr);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

int launch_contact_counts(sc_ctx* ctx, const double* d_coord, int64_t n, const sc_ff_desc& ff,
                          const PatchDev* patch, int64_t* d_counts) {
  return launch_kirchhoff(ctx, d_coord, n, 1, ff, patch, nullptr, nullptr, d_counts);
}

int launch_hessian(sc_ctx* ctx, const double* d_coord, int64_t n, int64_t batch,
                   const sc_ff_desc& ff, const PatchDev* patch, const double* d_w, double* d_h) {
  if (n <= 0 || batch <= 0) return SC_OK;
  constexpr int TI = 2;
  dim3 grid((unsigned)((n + TI - 1) / TI), (unsigned)batch);
  PatchDev p{};
  if (patch) {
    p = *patch;
    hipLaunchKernelGGL((k_hessian<TI, true, false>), grid, dim3(256), 0, ctx->stream, d_coord, (int)n,
                       make_ff(ff), p, d_w, d_h, (const AsmItem*)nullptr);
  } else {
    hipLaunchKernelGGL((k_hessian<TI, false, false>), grid, dim3(256), 0, ctx->stream, d_coord, (int)n,
                       make_ff(ff), p, d_w, d_h, (const AsmItem*)nullptr);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// Exclusive prefix sum of n int64 counts into n + 1 offsets (out[0] = 0, out[n] = total): one workgroup, every thread
// owns a contiguous chunk, chunk sums scanned with wave shuffles (the offsets of the ordered pair list, np.where order).
__global__ __launch_bounds__(1024) void k_exclusive_scan_i64(const long long* __restrict__ in, long long n,
                                                             long long* __restrict__ out) {
  __shared__ long long wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long per = (n + 1023) / 1024;
  const long long lo = (long long)tid * per, hi = lo + per < n ? lo + per : n;
  long long s = 0;
  for (long long i = lo; i < hi; ++i) s += in[i];
  long long incl = s;   // inclusive scan of the chunk sums inside the wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const long long t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  long long base = 0;
  for (int i = 0; i < w; ++i) base += wsum[i];
  long long run = base + incl - s;
  for (long long i = lo; i < hi; ++i) {
    out[i] = run;
    run += in[i];
  }
  if (tid == 1023) out[n] = base + incl;
}

int launch_exclusive_scan_i64(sc_ctx* ctx, const int64_t* d_in, int64_t n, int64_t* d_out) {
  hipLaunchKernelGGL(k_exclusive_scan_i64, dim3(1), dim3(1024), 0, ctx->stream, (const long long*)d_in, (long long)n,
                     (long long*)d_out);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

int launch_pair_fill(sc_ctx* ctx, const double* d_coord, int64_t n, const sc_ff_desc& ff,
                     const PatchDev* patch, const int64_t* d_offsets, int64_t* d_pairs,
                     double* d_sqdist) {
  if (n <= 0) return SC_OK;
  dim3 grid((unsigned)((n + 3) / 4));
  PatchDev p{};
  if (patch) {
    p = *patch;
    hipLaunchKernelGGL((k_pair_fill<true>), grid, dim3(256), 0, ctx->stream, d_coord, (int)n,
                       make_ff(ff), p, (const long long*)d_offsets, (long long*)d_pairs, d_sqdist);
  } else {
    hipLaunchKernelGGL((k_pair_fill<false>), grid, dim3(256), 0, ctx->stream, d_coord, (int)n,
                       make_ff(ff), p, (const long long*)d_offsets, (long long*)d_pairs, d_sqdist);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

int launch_kirchhoff_from_pairs(sc_ctx* ctx, int64_t n, const int64_t* d_pairs, int64_t k,
                                const double* d_gamma, double* d_k) {
  SC_HIP(ctx, hipMemsetAsync(d_k, 0, sizeof(double) * (size_t)n * n, ctx->stream));
  if (k > 0)
    hipLaunchKernelGGL(k_kirchhoff_scatter, dim3((unsigned)((k + 255) / 256)), dim3(256), 0,
                       ctx->stream, (int)n, (const long long*)d_pairs, (long long)k, d_gamma, d_k);
  hipLaunchKernelGGL(k_kirchhoff_diag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (int)n, d_k);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

int launch_hessian_from_pairs(sc_ctx* ctx, const double* d_coord, int64_t n, const int64_t* d_pairs,
                              int64_t k, const double* d_gamma, double* d_h) {
  SC_HIP(ctx, hipMemsetAsync(d_h, 0, sizeof(double) * (size_t)n * n * 9, ctx->stream));
  if (k > 0)
    hipLaunchKernelGGL(k_hessian_scatter, dim3((unsigned)((k + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_coord, (int)n, (const long long*)d_pairs, (long long)k,
                       d_gamma, d_h);
  hipLaunchKernelGGL(k_hessian_diag, dim3((unsigned)((9 * n + 255) / 256)), dim3(256), 0,
                     ctx->stream, (int)n, d_h);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// ---- ragged / decorated batches (sc_batch_plan) -----------------------------------------------------------------
size_t asm_item_bytes() { return sizeof(AsmItem); }

void asm_item_fill(void* dst, long long atom_off, int n, int ld, const sc_ff_desc& ff_dev, const PatchDev& patch) {
  AsmItem it{};
  it.atom_off = atom_off;
  it.n = n;
  it.ld = ld;
  it.ff = make_ff(ff_dev);
  it.patch = patch;
  *reinterpret_cast<AsmItem*>(dst) = it;
}

int launch_assemble_items(sc_ctx* ctx, int dim, const void* d_items, int64_t count, int max_atoms, bool any_patch,
                          bool any_pad, const double* d_coord, const double* d_w, double* d_matrix,
                          unsigned long long* d_bound_bits) {
  if (count <= 0 || max_atoms <= 0) return SC_OK;
  const AsmItem* items = reinterpret_cast<const AsmItem*>(d_items);
  hipStream_t st = ctx->stream;
  const FFDev ff0{};
  const PatchDev p0{};
  if (dim == 1) {
    constexpr int TI = 4;
    const dim3 grid((unsigned)((max_atoms + TI - 1) / TI), (unsigned)count);
    if (any_patch)
      hipLaunchKernelGGL((k_kirchhoff<TI, true, true>), grid, dim3(256), 0, st, d_coord, 0, ff0, p0, d_w, d_matrix,
                         (long long*)nullptr, items);
    else
      hipLaunchKernelGGL((k_kirchhoff<TI, false, true>), grid, dim3(256), 0, st, d_coord, 0, ff0, p0, d_w, d_matrix,
                         (long long*)nullptr, items);
  } else {
    constexpr int TI = 2;
    const dim3 grid((unsigned)((max_atoms + TI - 1) / TI), (unsigned)count);
    if (any_patch)
      hipLaunchKernelGGL((k_hessian<TI, true, true>), grid, dim3(256), 0, st, d_coord, 0, ff0, p0, d_w, d_matrix, items);
    else
      hipLaunchKernelGGL((k_hessian<TI, false, true>), grid, dim3(256), 0, st, d_coord, 0, ff0, p0, d_w, d_matrix, items);
  }
  if (any_pad) {
    SC_HIP(ctx, hipMemsetAsync(d_bound_bits, 0, sizeof(unsigned long long) * (size_t)count, st));
    hipLaunchKernelGGL(k_slot_bound, dim3((unsigned)((dim * max_atoms + 3) / 4), (unsigned)count), dim3(256), 0, st,
                       d_matrix, items, dim, d_bound_bits);
    hipLaunchKernelGGL(k_pad_fill, dim3(64, (unsigned)count), dim3(256), 0, st, d_matrix, items, dim, d_bound_bits);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// Contact counts of every structure of a plan (counts of all atoms back to back), one launch.
int launch_items_counts(sc_ctx* ctx, const void* d_items, int64_t count, int max_atoms, bool any_patch,
                        const double* d_coord, int64_t* d_counts) {
  if (count <= 0 || max_atoms <= 0) return SC_OK;
  const AsmItem* items = reinterpret_cast<const AsmItem*>(d_items);
  constexpr int TI = 4;
  const dim3 grid((unsigned)((max_atoms + TI - 1) / TI), (unsigned)count);
  const FFDev ff0{};
  const PatchDev p0{};
  if (any_patch)
    hipLaunchKernelGGL((k_kirchhoff<TI, true, true>), grid, dim3(256), 0, ctx->stream, d_coord, 0, ff0, p0,
                       (const double*)nullptr, (double*)nullptr, (long long*)d_counts, items);
  else
    hipLaunchKernelGGL((k_kirchhoff<TI, false, true>), grid, dim3(256), 0, ctx->stream, d_coord, 0, ff0, p0,
                       (const double*)nullptr, (double*)nullptr, (long long*)d_counts, items);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// Ordered pair lists (np.where order, local atom indices) of every structure, back to back; d_offsets: exclusive scan of
// the counts over all atoms.
int launch_items_pair_fill(sc_ctx* ctx, const void* d_items, int64_t count, int max_atoms, bool any_patch,
                           const double* d_coord, const int64_t* d_offsets, int64_t* d_pairs, double* d_sqdist) {
  if (count <= 0 || max_atoms <= 0) return SC_OK;
  const AsmItem* items = reinterpret_cast<const AsmItem*>(d_items);
  const dim3 grid((unsigned)((max_atoms + 3) / 4), (unsigned)count);
  const FFDev ff0{};
  const PatchDev p0{};
  if (any_patch)
    hipLaunchKernelGGL((k_pair_fill<true, true>), grid, dim3(256), 0, ctx->stream, d_coord, 0, ff0, p0,
                       (const long long*)d_offsets, (long long*)d_pairs, d_sqdist, items);
  else
    hipLaunchKernelGGL((k_pair_fill<false, true>), grid, dim3(256), 0, ctx->stream, d_coord, 0, ff0, p0,
                       (const long long*)d_offsets, (long long*)d_pairs, d_sqdist, items);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// All slots of a plan from explicit pair lists + gamma (the host-callback path of compute_kirchhoff / compute_hessian,
// interaction.py:49-52 / :94-104, for a whole batch): zero, scatter, diagonal, mass weights, pads.
int launch_items_from_pairs(sc_ctx* ctx, int dim, const void* d_items, int64_t count, int max_atoms, int64_t order,
                            bool any_pad, const double* d_coord, const int64_t* d_pair_off, int64_t k_total,
                            const int64_t* d_pairs, const double* d_gamma, const double* d_w, double* d_matrix,
                            unsigned long long* d_bound_bits) {
  if (count <= 0 || max_atoms <= 0) return SC_OK;
  const AsmItem* items = reinterpret_cast<const AsmItem*>(d_items);
  hipStream_t st = ctx->stream;
  SC_HIP(ctx, hipMemsetAsync(d_matrix, 0, sizeof(double) * (size_t)count * order * order, st));
  const dim3 gp((unsigned)((k_total + 255) / 256));
  const dim3 gd((unsigned)(((size_t)dim * dim * max_atoms + 255) / 256), (unsigned)count);
  if (dim == 1) {
    if (k_total > 0)
      hipLaunchKernelGGL((k_items_scatter<1>), gp, dim3(256), 0, st, items, (int)count, (const long long*)d_pair_off,
                         (const long long*)d_pairs, d_gamma, d_coord, d_matrix);
    hipLaunchKernelGGL((k_items_diag<1>), gd, dim3(256), 0, st, items, d_matrix);
  } else {
    if (k_total > 0)
      hipLaunchKernelGGL((k_items_scatter<3>), gp, dim3(256), 0, st, items, (int)count, (const long long*)d_pair_off,
                         (const long long*)d_pairs, d_gamma, d_coord, d_matrix);
    hipLaunchKernelGGL((k_items_diag<3>), gd, dim3(256), 0, st, items, d_matrix);
  }
  if (d_w) hipLaunchKernelGGL(k_items_weight, dim3(256, (unsigned)count), dim3(256), 0, st, items, dim, d_w, d_matrix);
  if (any_pad) {
    SC_HIP(ctx, hipMemsetAsync(d_bound_bits, 0, sizeof(unsigned long long) * (size_t)count, st));
    hipLaunchKernelGGL(k_slot_bound, dim3((unsigned)((dim * max_atoms + 3) / 4), (unsigned)count), dim3(256), 0, st,
                       d_matrix, items, dim, d_bound_bits);
    hipLaunchKernelGGL(k_pad_fill, dim3(64, (unsigned)count), dim3(256), 0, st, d_matrix, items, dim, d_bound_bits);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}
