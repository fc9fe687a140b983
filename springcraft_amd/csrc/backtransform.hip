// Back-transformation  Z <- Q_H Z  (stage K7): applies the Householder reflectors of the
// tridiagonalisation, NBT = 256 at a time, as compact-WY block reflectors  I - V T V^T  so that all O(n^3)
// work is f64-MFMA GEMM (the role of LAPACK dormtr inside np.linalg.eigh, reference call site nma.py:61).
//
// Everything that does not depend on Z is done once, for ALL blocks, in four launches:
//   k_bt_clean      zero, in place, the entries of A above each reflector's unit entry inside its block (A is
//                   scratch by now), so V_p is a plain sub-matrix view of A
//   GEMM (grouped, split-K)   G_p = V_p^T V_p
//   k_bt_tfactor    the two 128 x 128 diagonal blocks T1, T2 of T_p from tau and G_p (dlarft recurrence, forward /
//                   columnwise, one workgroup per diagonal block: 128 x 128 is what fits in LDS)
//   2 GEMMs         the off-diagonal block  T12 = -T1 (V1^T V2) T2  (larft's blocked form)
//   GEMM (grouped)  VT_p = V_p T_p
// Blocks of 256 instead of 128 reflectors halve the passes over Z (each block reads Z once for W1 and reads and
// writes it once for the update: at K = 128 that traffic, not the MFMAs, set the pace: 866 + 433 GB per step).
// and then, last block first, three launches per block:
//   GEMM (split-K)  W1 = V_p^T Z[rows]        (NBT x n, K = rows: split so the grid fills the chip)
//   k_bt_sum        W  = sum of the K slices
//   GEMM            Z[rows] -= VT_p W
#include <algorithm>
#include <vector>

#include "eigh_internal.h"

namespace {

constexpr int kNbt = 256;    // reflectors per block
constexpr int kNbtT = 128;   // diagonal blocks of T computed by the recurrence
constexpr int kGramSplits = 4;

__global__ __launch_bounds__(256) void k_bt_clean(double* __restrict__ a_all, long long stride_a, int n,
                                                  int nbt, int nref, int off) {
  double* A = a_all + (size_t)blockIdx.y * stride_a;
  const int cs = blockIdx.x * nbt;
  // (a) triangle above the unit entries: rows cs+off .. c+off-1 of column c, for c in the block
  for (int idx = threadIdx.x; idx < nbt * nbt; idx += blockDim.x) {
    const int q = idx / nbt, rr = idx % nbt;  // column cs+q, row cs+off+rr
    const int c = cs + q, r = cs + off + rr;
    if (c < n && r < n && r < c + off) A[(size_t)c * n + r] = 0.0;
  }
  // (b) columns without a reflector (c >= nref): zero rows cs+off ..
  for (int c = std::max(cs, nref); c < std::min(cs + nbt, n); ++c)
    for (int r = cs + off + threadIdx.x; r < n; r += blockDim.x) A[(size_t)c * n + r] = 0.0;
}

// Diagonal 128 x 128 blocks of T (upper triangular, nbt x nbt, column-major) from the Gram slices and tau; grid (2 x block,
// matrix).  The first workgroup of a block also sums the Gram slices of the off-diagonal block G12 = V1^T V2 into g12 and
// zeroes the lower-left block of T.
__global__ __launch_bounds__(256) void k_bt_tfactor(const double* __restrict__ tri_all, TriLayout TL,
                                                    double* __restrict__ bt_all, BtLayout BL, int nref) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int nt = kNbtT;
  const int nbt = BL.nbt;
  double* T = sm;               // nt x nt
  double* gcol = sm + nt * nt;
  const int p = blockIdx.x >> 1, h = blockIdx.x & 1;
  const double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  double* bt = bt_all + (size_t)blockIdx.y * BL.slab;
  const double* gram_p = bt + BL.gram + (size_t)p * BL.splits_g * nbt * nbt;
  const double* gram = gram_p + (size_t)(h * nt) * nbt + h * nt;   // diagonal block h
  double* tout = bt + BL.t + (size_t)p * nbt * nbt;
  const int cs = p * nbt + h * nt;
  const int kk = std::max(0, std::min(nt, nref - cs));
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nt * nt; idx += blockDim.x) T[idx] = 0.0;
  if (h == 0) {
    double* g12 = bt + BL.g12 + (size_t)p * nt * nt;
    const int pc = std::min(nbt, BL.n - p * nbt);   // columns of this block that exist: the Gram product wrote pc x pc
    for (int idx = tid; idx < nt * nt; idx += blockDim.x) {
      const int i = idx % nt, j = idx / nt;
      double s2 = 0.0;
      if (i < pc && nt + j < pc)
        for (int sl = 0; sl < BL.splits_g; ++sl) s2 += gram_p[(size_t)sl * nbt * nbt + (size_t)(nt + j) * nbt + i];
      g12[idx] = s2;
      tout[(size_t)j * nbt + nt + i] = 0.0;   // lower-left block
    }
  }
  __syncthreads();
  for (int q = 0; q < kk; ++q) {
    if (tid < q) {
      double s = 0.0;
      for (int sl = 0; sl < BL.splits_g; ++sl) s += gram[(size_t)sl * nbt * nbt + (size_t)q * nbt + tid];
      gcol[tid] = s;   // G[tid, q] = v_tid . v_q
    }
    __syncthreads();
    const double tau = tri[TL.tau + cs + q];
    if (tid < q) {
      // T[0:q, q] = -tau * T[0:q, 0:q] * G[0:q, q]
      double s = 0.0;
      for (int l = tid; l < q; ++l) s += T[tid + l * nt] * gcol[l];
      T[tid + q * nt] = -tau * s;
    }
    if (tid == q) T[q + q * nt] = tau;
    __syncthreads();
  }
  for (int idx = tid; idx < nt * nt; idx += blockDim.x) {
    const int i = idx % nt, j = idx / nt;
    tout[(size_t)(h * nt + j) * nbt + h * nt + i] = T[idx];
  }
}

__global__ __launch_bounds__(256) void k_bt_sum(double* __restrict__ bt_all, BtLayout BL, int splits, int ncols) {
  double* bt = bt_all + (size_t)blockIdx.y * BL.slab;
  const size_t total = (size_t)BL.nbt * ncols;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int sl = 0; sl < splits; ++sl) s += bt[BL.w1 + (size_t)sl * total + idx];
    bt[BL.w2 + idx] = s;
  }
}

int w1_splits_for(int ncols, int batch) {
  // W1 is only NBT rows tall: 2 x ceil(ncols/64) tiles of 64x64 per matrix; split K until ~3000 workgroups exist
  const int tiles = 2 * ((ncols + 63) / 64) * batch;
  const int s = (3000 + tiles - 1) / tiles;
  return std::max(1, std::min(8, s));
}

}  // namespace

size_t bt_slab_doubles(int n, BtLayout* out) {
  BtLayout L{};
  L.n = n;
  L.nbt = kNbt;
  L.splits = 8;
  L.splits_g = kGramSplits;
  const long long npanels = std::max(1, (std::max(n - 2, 0) + kNbt - 1) / kNbt);
  long long off = 0;
  auto take = [&](long long cnt) { long long o = off; off += (cnt + 7) / 8 * 8; return o; };
  L.vc = 0;
  L.gram = take(npanels * L.splits_g * kNbt * kNbt);
  L.t = take(npanels * kNbt * kNbt);
  L.g12 = take(npanels * kNbtT * kNbtT);
  L.tx = take(npanels * kNbtT * kNbtT);
  L.w1 = take((long long)L.splits * kNbt * n);
  L.w2 = take((long long)kNbt * n);
  L.slab = off;
  if (out) *out = L;
  return (size_t)off;
}

int bt_desc_count(int n, int batch) {
  const int nref = std::max(n - 2, 0);
  const int npanels = (nref + kNbt - 1) / kNbt;
  return npanels * 6 * batch;
}

// d_a is modified (cleaned); d_vt: (batch, n, n) scratch that receives V T.
int backtransform_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, const double* d_tri_ws,
                          const TriLayout& TL, double* d_bt_ws, const BtLayout& BL, double* d_z,
                          long long stride_z, int ncols, double* d_vt, GemmDesc* d_descs, int off, int phase) {
  // phase: 0 everything (on ctx->stream); or in three calls, so that the part that neither needs Z nor the scratch the
  // D&C also uses can run on a second stream beside the D&C: 1 = descriptor upload, 2 = clean V, Gram products, T factors
  // (launched on whatever ctx->stream is at the call), 3 = V T and the block applications
  hipStream_t st = ctx->stream;
  const int nref = n - 1 - off;  // reflector columns 0 .. nref-1 (a reflector needs two rows)
  if (nref <= 0) return SC_OK;
  const int nbt = BL.nbt;
  const int npanels = (nref + nbt - 1) / nbt;
  int w1s = std::min(BL.splits, w1_splits_for(ncols, batch));
  // W = V^T Z on k_gemm3 (the role-split kernel; its K loop runs at 0.84 of the MFMA peak against k_gemm2's 0.74) when the
  // batch gives every CU a few of the 2 x ceil(ncols / 64) tiles per matrix without K slices; decided here because the
  // records say where W goes (a single slice writes straight to where the update reads it)
  static const bool env_w3 = [] { const char* e = getenv("SPRINGCRAFT_GEMM3_W"); return !e || atoi(e) != 0; }();
  const bool al16 = (n & 1) == 0 && (off & 1) == 0 && (stride_z & 1) == 0 && (stride_a & 1) == 0 && (nbt & 1) == 0;
  const bool w_gemm3 = env_w3 && al16 && gemm3_would_take(ctx, batch, nbt, ncols, n - off, kGemmAkBk, false, 1.0, 0.0, al16);
  if (w_gemm3) w1s = 1;

  // descriptor table: [gram | vt | w1 | update | x = G12 T2 | T12 = -T1 x] x npanels x batch  (each group contiguous
  // for one launch)
  const size_t grp = (size_t)npanels * batch;
  std::vector<GemmDesc> h(6 * grp);
  for (int p = 0; p < npanels; ++p) {
    const int cs = p * nbt;
    const int mrow = n - cs - off;
    const int pc = std::min(nbt, n - cs);   // columns of this block that exist in the matrix
    for (int b = 0; b < batch; ++b) {
      double* bt = d_bt_ws + (size_t)b * BL.slab;
      const double* vp = d_a + (size_t)b * stride_a + (size_t)cs * n + cs + off;
      double* vtp = d_vt + (size_t)b * stride_a + (size_t)cs * n + cs + off;
      double* tp = bt + BL.t + (size_t)p * nbt * nbt;
      GemmDesc G{};   // G_p = V^T V, split-K slices
      G.a = vp; G.sa_i = n; G.sa_k = 1;
      G.b = vp; G.sb_k = 1; G.sb_j = n;
      G.c = bt + BL.gram + (size_t)p * BL.splits_g * nbt * nbt; G.ldc = nbt;
      G.m = pc; G.n = pc; G.k = mrow;
      G.alpha = 1.0; G.beta = 0.0;
      G.split_stride = (long long)nbt * nbt;
      h[0 * grp + (size_t)p * batch + b] = G;
      GemmDesc V{};   // VT_p = -V T
      V.a = vp; V.sa_i = 1; V.sa_k = n;
      V.b = tp; V.sb_k = 1; V.sb_j = nbt;
      V.c = vtp; V.ldc = n;
      V.m = mrow; V.n = pc; V.k = pc;
      V.alpha = -1.0; V.beta = 0.0;   // -(V T): the update below then ADDS (alpha = 1, what k_gemm3 takes); exact, bit for bit
      h[1 * grp + (size_t)p * batch + b] = V;
      GemmDesc W{};   // W1 = V^T Z[rows], split-K slices
      W.a = vp; W.sa_i = n; W.sa_k = 1;
      W.b = d_z + (size_t)b * stride_z + cs + off; W.sb_k = 1; W.sb_j = n;
      W.c = bt + (w1s > 1 ? BL.w1 : BL.w2); W.ldc = nbt;   // a single K slice goes straight to where the update reads it
      W.m = pc; W.n = ncols; W.k = mrow;
      W.alpha = 1.0; W.beta = 0.0;
      W.split_stride = (long long)nbt * ncols;
      h[2 * grp + (size_t)p * batch + b] = W;
      GemmDesc U{};   // Z[rows] += (-V T) W
      U.a = vtp; U.sa_i = 1; U.sa_k = n;
      U.b = bt + BL.w2; U.sb_k = 1; U.sb_j = nbt;
      U.c = d_z + (size_t)b * stride_z + cs + off; U.ldc = n;
      U.m = mrow; U.n = ncols; U.k = pc;
      U.alpha = 1.0; U.beta = 1.0;
      h[3 * grp + (size_t)p * batch + b] = U;
      double* g12 = bt + BL.g12 + (size_t)p * kNbtT * kNbtT;
      double* tx = bt + BL.tx + (size_t)p * kNbtT * kNbtT;
      GemmDesc X{};   // x = G12 T2
      X.a = g12; X.sa_i = 1; X.sa_k = kNbtT;
      X.b = tp + (size_t)kNbtT * nbt + kNbtT; X.sb_k = 1; X.sb_j = nbt;
      X.c = tx; X.ldc = kNbtT;
      X.m = kNbtT; X.n = kNbtT; X.k = kNbtT;
      X.alpha = 1.0; X.beta = 0.0;
      h[4 * grp + (size_t)p * batch + b] = X;
      GemmDesc Y{};   // T12 = -T1 x
      Y.a = tp; Y.sa_i = 1; Y.sa_k = nbt;
      Y.b = tx; Y.sb_k = 1; Y.sb_j = kNbtT;
      Y.c = tp + (size_t)kNbtT * nbt; Y.ldc = nbt;
      Y.m = kNbtT; Y.n = kNbtT; Y.k = kNbtT;
      Y.alpha = -1.0; Y.beta = 0.0;
      h[5 * grp + (size_t)p * batch + b] = Y;
    }
  }
  if (phase == 0 || phase == 1) SC_TRY(sc_stage_upload(ctx, d_descs, h.data(), h.size() * sizeof(GemmDesc)));
  if (phase == 1) return SC_OK;

  if (phase == 0 || phase == 2) {
  hipLaunchKernelGGL(k_bt_clean, dim3((unsigned)npanels, (unsigned)batch), dim3(256), 0, st, d_a, stride_a, n, nbt,
                     nref, off);
  SC_TRY(launch_gemm_f64(ctx, d_descs, (int)grp, nbt, nbt, kGemmTile, BL.splits_g, false, false, kGemmAkBk));
  hipLaunchKernelGGL(k_bt_tfactor, dim3((unsigned)(2 * npanels), (unsigned)batch), dim3(256),
                     sizeof(double) * (kNbtT * kNbtT + kNbtT), st, d_tri_ws, TL, d_bt_ws, BL, nref);
  SC_TRY(launch_gemm_f64(ctx, d_descs + 4 * grp, (int)grp, kNbtT, kNbtT, kGemmTile, 1, false, false, kGemmAmBk));
  SC_TRY(launch_gemm_f64(ctx, d_descs + 5 * grp, (int)grp, kNbtT, kNbtT, kGemmTile, 1, false, false, kGemmAmBk));
  }
  if (phase == 2) {
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
  }
  SC_TRY(launch_gemm_f64(ctx, d_descs + grp, (int)grp, n, nbt, kGemmTile, 1, false, false, kGemmAmBk));

  PhaseTimer t_w(ctx, "bt1_w", st), t_u(ctx, "bt1_update", st);
  for (int p = npanels - 1; p >= 0; --p) {
    const int mrow = n - p * nbt - off;
    t_w.start();
    {
      const int pc = std::min(nbt, n - p * nbt);
      const bool al = al16 && ((p * nbt) & 1) == 0;
      if (!w_gemm3 || launch_gemm3_uniform(ctx, d_descs + 2 * grp + (size_t)p * batch, batch, pc, ncols, mrow, kGemmAkBk, false,
                                           1.0, 0.0, al) != SC_OK)
        SC_TRY(launch_gemm_f64(ctx, d_descs + 2 * grp + (size_t)p * batch, batch, nbt, ncols, kGemmTile, w1s, false, false,
                               kGemmAkBk));
    }
    if (w1s > 1) hipLaunchKernelGGL(k_bt_sum, dim3(256, (unsigned)batch), dim3(256), 0, st, d_bt_ws, BL, w1s, ncols);
    t_w.stop();
    t_u.start();
    {
      const int pc = std::min(nbt, n - p * nbt);
      const bool al = (n & 1) == 0 && ((p * nbt + off) & 1) == 0 && (stride_z & 1) == 0 && (stride_a & 1) == 0 && (nbt & 1) == 0;
      if (launch_gemm3_uniform(ctx, d_descs + 3 * grp + (size_t)p * batch, batch, mrow, ncols, pc, kGemmAmBk, false, 1.0, 1.0,
                               al) != SC_OK)
        SC_TRY(launch_gemm_f64(ctx, d_descs + 3 * grp + (size_t)p * batch, batch, mrow, ncols, kGemmTile, 1, false, false,
                               kGemmAmBk));
    }
    t_u.stop();
  }
  SC_HIP(ctx, hipGetLastError());
  t_w.finish(); t_u.finish();
  return SC_OK;
}
