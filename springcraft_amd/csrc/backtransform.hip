// Back-transformation  Z <- Q_H Z  (stage K7): applies the Householder reflectors of the
// tridiagonalisation, NBT at a time, as compact-WY block reflectors  I - V T V^T  so that all O(n^3)
// work is f64-MFMA GEMM (the role of LAPACK dormtr inside np.linalg.eigh, reference call site nma.py:61).
//
// Per block of reflectors (last block first):
//   k_bt_extract   clean copy of V (zeros above the unit diagonal; A's storage there holds other data)
//   GEMM (split-K) G  = V^T V                         (NBT x NBT)
//   k_bt_tfactor   T  from tau and G  (dlarft recurrence, forward / columnwise)
//   GEMM (split-K) W1 = V^T Z[rows]                   (NBT x n, K = rows: split so the grid fills the chip)
//   k_bt_apply_t   W2 = T * sum_slices(W1)
//   GEMM           Z[rows] -= V W2
#include <vector>

#include "eigh_internal.h"

namespace {

__global__ void k_bt_extract(const double* __restrict__ a_all, long long stride_a, double* __restrict__ bt_all,
                             BtLayout BL, int cs, int kk) {
  const int n = BL.n;
  const double* A = a_all + (size_t)blockIdx.z * stride_a;
  double* vc = bt_all + (size_t)blockIdx.z * BL.slab + BL.vc;
  const int q = blockIdx.y;
  const int mrow = n - cs - 1;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= mrow) return;
  double v = 0.0;
  if (q < kk && r >= q) v = A[(size_t)(cs + q) * n + (cs + 1 + r)];
  vc[(size_t)q * n + r] = v;
}

// T (upper triangular, nbt x nbt, column-major ld nbt) from the Gram matrix slices and tau.
__global__ __launch_bounds__(256) void k_bt_tfactor(const double* __restrict__ tri_all, TriLayout TL,
                                                    double* __restrict__ bt_all, BtLayout BL, int cs, int kk) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int nbt = BL.nbt;
  double* G = sm;               // nbt x nbt
  double* T = sm + nbt * nbt;   // nbt x nbt
  const double* tri = tri_all + (size_t)blockIdx.x * TL.slab;
  double* bt = bt_all + (size_t)blockIdx.x * BL.slab;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nbt * nbt; idx += blockDim.x) {
    double s = 0.0;
    for (int sl = 0; sl < BL.splits_g; ++sl) s += bt[BL.gram + (size_t)sl * nbt * nbt + idx];
    G[idx] = s;
    T[idx] = 0.0;
  }
  __syncthreads();
  for (int q = 0; q < kk; ++q) {
    const double tau = tri[TL.tau + cs + q];
    // T[0:q, q] = -tau * T[0:q, 0:q] * G[0:q, q]
    if (tid < q) {
      double s = 0.0;
      for (int l = tid; l < q; ++l) s += T[tid + l * nbt] * G[l + q * nbt];
      T[tid + q * nbt] = -tau * s;
    }
    if (tid == q) T[q + q * nbt] = tau;
    __syncthreads();
  }
  for (int idx = tid; idx < nbt * nbt; idx += blockDim.x) bt[BL.t + idx] = T[idx];
}

// W2[:, j] = T * sum_s W1_s[:, j]
__global__ __launch_bounds__(256) void k_bt_apply_t(double* __restrict__ bt_all, BtLayout BL, int splits) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int nbt = BL.nbt, n = BL.n;
  double* T = sm;                 // nbt x nbt
  double* X = sm + nbt * nbt;     // nbt x cols_per_block
  double* bt = bt_all + (size_t)blockIdx.y * BL.slab;
  const int cols_per_block = blockDim.x / nbt;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nbt * nbt; idx += blockDim.x) T[idx] = bt[BL.t + idx];
  const int i = tid % nbt, cl = tid / nbt;
  const int j = blockIdx.x * cols_per_block + cl;
  double x = 0.0;
  if (j < n)
    for (int s = 0; s < splits; ++s) x += bt[BL.w1 + (size_t)s * nbt * n + (size_t)j * nbt + i];
  X[cl * nbt + i] = x;
  __syncthreads();
  if (j < n) {
    double acc = 0.0;
    for (int l = i; l < nbt; ++l) acc += T[i + l * nbt] * X[cl * nbt + l];
    bt[BL.w2 + (size_t)j * nbt + i] = acc;
  }
}

constexpr int kNbt = 64;

int splits_for(int n, int mrow) {
  const int col_tiles = (n + 127) / 128;
  int s = (640 + col_tiles - 1) / col_tiles;
  const int cap = mrow / 128 > 1 ? mrow / 128 : 1;
  if (s > cap) s = cap;
  if (s > 32) s = 32;
  return s < 1 ? 1 : s;
}

}  // namespace

size_t bt_slab_doubles(int n, BtLayout* out) {
  BtLayout L{};
  L.n = n;
  L.nbt = kNbt;
  L.splits = splits_for(n, n);
  L.splits_g = 32;
  long long off = 0;
  auto take = [&](long long cnt) { long long o = off; off += (cnt + 7) / 8 * 8; return o; };
  L.vc = take((long long)n * kNbt);
  L.gram = take((long long)L.splits_g * kNbt * kNbt);
  L.t = take((long long)kNbt * kNbt);
  L.w1 = take((long long)L.splits * kNbt * n);
  L.w2 = take((long long)kNbt * n);
  L.slab = off;
  if (out) *out = L;
  return (size_t)off;
}

int backtransform_batched(sc_ctx* ctx, const double* d_a, long long stride_a, int n, int batch,
                          const double* d_tri_ws, const TriLayout& TL, double* d_bt_ws, const BtLayout& BL,
                          double* d_z, long long stride_z, GemmDesc* d_descs) {
  hipStream_t st = ctx->stream;
  const int nref = n - 2;  // reflector columns 0 .. n-3
  if (nref <= 0) return SC_OK;
  const int nbt = BL.nbt;
  const int npanels = (nref + nbt - 1) / nbt;

  // descriptors: per panel [gram | w1 | update] x batch
  std::vector<GemmDesc> h((size_t)npanels * 3 * batch);
  std::vector<int> w1_splits(npanels), g_splits(npanels);
  for (int p = 0; p < npanels; ++p) {
    const int cs = p * nbt;
    const int mrow = n - cs - 1;
    w1_splits[p] = std::min(BL.splits, splits_for(n, mrow));
    g_splits[p] = std::max(1, std::min(BL.splits_g, mrow / 128));
    for (int b = 0; b < batch; ++b) {
      double* bt = d_bt_ws + (size_t)b * BL.slab;
      double* vc = bt + BL.vc;
      GemmDesc G{};
      G.a = vc; G.sa_i = n; G.sa_k = 1;
      G.b = vc; G.sb_k = 1; G.sb_j = n;
      G.c = bt + BL.gram; G.ldc = nbt;
      G.m = nbt; G.n = nbt; G.k = mrow;
      G.alpha = 1.0; G.beta = 0.0;
      G.split_stride = (long long)nbt * nbt;
      h[((size_t)p * 3 + 0) * batch + b] = G;
      GemmDesc W{};
      W.a = vc; W.sa_i = n; W.sa_k = 1;
      W.b = d_z + (size_t)b * stride_z + cs + 1; W.sb_k = 1; W.sb_j = n;
      W.c = bt + BL.w1; W.ldc = nbt;
      W.m = nbt; W.n = n; W.k = mrow;
      W.alpha = 1.0; W.beta = 0.0;
      W.split_stride = (long long)nbt * n;
      h[((size_t)p * 3 + 1) * batch + b] = W;
      GemmDesc U{};
      U.a = vc; U.sa_i = 1; U.sa_k = n;
      U.b = bt + BL.w2; U.sb_k = 1; U.sb_j = nbt;
      U.c = d_z + (size_t)b * stride_z + cs + 1; U.ldc = n;
      U.m = mrow; U.n = n; U.k = nbt;
      U.alpha = -1.0; U.beta = 1.0;
      h[((size_t)p * 3 + 2) * batch + b] = U;
    }
  }
  SC_HIP(ctx, hipMemcpyAsync(d_descs, h.data(), h.size() * sizeof(GemmDesc), hipMemcpyHostToDevice, st));

  for (int p = npanels - 1; p >= 0; --p) {
    const int cs = p * nbt;
    const int kk = std::min(nbt, nref - cs);
    const int mrow = n - cs - 1;
    hipLaunchKernelGGL(k_bt_extract, dim3((unsigned)((mrow + 255) / 256), (unsigned)nbt, (unsigned)batch),
                       dim3(256), 0, st, d_a, stride_a, d_bt_ws, BL, cs, kk);
    const GemmDesc* dp = d_descs + (size_t)p * 3 * batch;
    SC_TRY(launch_gemm_f64(ctx, dp, batch, nbt, nbt, 1, g_splits[p]));
    {
      BtLayout B2 = BL;
      B2.splits_g = g_splits[p];
      hipLaunchKernelGGL(k_bt_tfactor, dim3((unsigned)batch), dim3(256), sizeof(double) * 2 * nbt * nbt, st,
                         d_tri_ws, TL, d_bt_ws, B2, cs, kk);
    }
    SC_TRY(launch_gemm_f64(ctx, dp + batch, batch, nbt, n, 1, w1_splits[p]));
    {
      const int cols_per_block = 256 / nbt;
      hipLaunchKernelGGL(k_bt_apply_t, dim3((unsigned)((n + cols_per_block - 1) / cols_per_block), (unsigned)batch),
                         dim3(256), sizeof(double) * (nbt * nbt + 256), st, d_bt_ws, BL, w1_splits[p]);
    }
    SC_TRY(launch_gemm_f64(ctx, dp + 2 * (size_t)batch, batch, mrow, n, 0));
  }
  SC_HIP(ctx, hipGetLastError());
  SC_HIP(ctx, hipStreamSynchronize(st));  // `h` must outlive the descriptor upload
  return SC_OK;
}
