// float64 GEMM on the CDNA4 matrix cores — see gemm_f64.h for the contract.
//
// Block tile BM x 128 (BM = 128 or 64), K step 16, 256 threads = 4 waves arranged 2 x 2, each wave
// owning a (BM/2) x 64 sub-tile built from v_mfma_f64_16x16x4_f64 tiles (64 lanes x 4 f64 results).
// Two blocks per CU = 2 waves per SIMD, >= 8 independent accumulators per wave: the regime in which
// the f64 MFMA pipe reached its practical ceiling in profiles/r01_probe_f64.txt (47 TFLOP/s).
//
// Operand staging: global -> registers (next K step, issued before the MFMAs of the current step)
// -> LDS [k][m|n] double-buffered, one barrier per K step.  LDS rows are padded to tile+16 doubles
// so the four k-rows a ds_read_b64 wave-instruction touches fall on disjoint banks.
//
// The MFMA is fed "swapped" (its A operand comes from the B tile, its B operand from the A tile):
// the accumulator then holds C^T fragments whose 16 consecutive lanes map to 16 consecutive ROWS
// of the column-major C, so C loads / stores are 128-byte contiguous segments.
#include <cmath>
#include <vector>

#include "gemm_f64.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int PAD = 16;

template <int BM, int BN, int BK, int MINW, bool GATHER, bool TRI = false>
__global__ __launch_bounds__(256, MINW) void k_gemm_f64(const GemmDesc* __restrict__ descs, int split_k) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int LDA_S = BM + PAD, LDB_S = BN + PAD;
  constexpr int A_PER_THREAD = BM * BK / 256, B_PER_THREAD = BN * BK / 256;

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sA = smem;                        // [2][BK][LDA_S]
  double* sB = smem + 2 * BK * LDA_S;       // [2][BK][LDB_S]

  const int z = blockIdx.z;
  const int slice = split_k > 1 ? z % split_k : 0;
  const GemmDesc& D = descs[split_k > 1 ? z / split_k : z];
  const int M = D.m, N = D.n, K = D.k;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  if (m0 >= M || n0 >= N) return;   // K == 0 still runs: it stores beta*C (zeros for beta = 0)
  if (D.lower_only && (m0 + BM - 1 + D.row_off) < (n0 + D.col_off)) return;

  int k_begin = 0, k_end = K;
  if (split_k > 1) {
    const int chunk = ((K + split_k - 1) / split_k + BK - 1) / BK * BK;
    k_begin = slice * chunk;
    k_end = min(K, k_begin + chunk);
  }
  // triangular A operand (TRI instantiation, the symmetric products of the band reduction): the block only walks
  // the K range in which its rows have non-zero entries
  const int tri = TRI ? D.a_tri : 0;
  if (TRI && tri == 1) k_end = min(k_end, m0 + BM);
  if (TRI && tri == 2) k_begin = max(k_begin, m0);

  const double* __restrict__ A = D.a;
  const double* __restrict__ B = D.b;
  const long long sa_i = D.sa_i, sa_k = D.sa_k, sb_k = D.sb_k, sb_j = D.sb_j;
  // gather lists are only looked at by the GATHER instantiation (the D&C merges); the plain one pays nothing
  const int* __restrict__ kidx = GATHER ? D.a_kidx : nullptr;
  const int* __restrict__ bkidx = GATHER ? D.b_kidx : nullptr;
  const bool a_mcontig = (sa_i == 1);
  const bool b_ncontig = (sb_j == 1);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  double ra[A_PER_THREAD], rb[B_PER_THREAD];

  auto load_tiles = [&](int kt) {
    // A tile: BM x BK
    if (a_mcontig) {
      const int i = tid % BM, kq = tid / BM;
      constexpr int KSTEP = 256 / BM;
#pragma unroll
      for (int p = 0; p < A_PER_THREAD; ++p) {
        const int k = kt + kq + p * KSTEP;
        const int gi = m0 + i;
        double v = 0.0;
        if (gi < M && k < k_end && (!TRI || tri == 0 || (tri == 1 ? gi >= k : k > gi))) {
          const long long kk = (GATHER && kidx) ? (long long)kidx[k] : (long long)k;
          v = A[(long long)gi + kk * sa_k];
        }
        ra[p] = v;
      }
    } else {
      const int k = tid % BK, iq = tid / BK;
      constexpr int ISTEP = 256 / BK;
#pragma unroll
      for (int p = 0; p < A_PER_THREAD; ++p) {
        const int gi = m0 + iq + p * ISTEP;
        const int gk = kt + k;
        ra[p] = (gi < M && gk < k_end && (!TRI || tri == 0 || (tri == 1 ? gi >= gk : gk > gi)))
                    ? A[(long long)gi * sa_i + gk] : 0.0;
      }
    }
    // B tile: BK x BN
    if (b_ncontig) {
      const int j = tid % BN, kq = tid / BN;
      constexpr int KSTEP = 256 / BN;
#pragma unroll
      for (int p = 0; p < B_PER_THREAD; ++p) {
        const int k = kt + kq + p * KSTEP;
        const int gj = n0 + j;
        double v = 0.0;
        if (gj < N && k < k_end) v = B[((GATHER && bkidx) ? (long long)bkidx[k] : (long long)k) * sb_k + gj];
        rb[p] = v;
      }
    } else {
      const int k = tid % BK, jq = tid / BK;
      constexpr int JSTEP = 256 / BK;
#pragma unroll
      for (int p = 0; p < B_PER_THREAD; ++p) {
        const int gj = n0 + jq + p * JSTEP;
        const int gk = kt + k;
        double v = 0.0;
        if (gj < N && gk < k_end)
          v = B[((GATHER && bkidx) ? (long long)bkidx[gk] : (long long)gk) + (long long)gj * sb_j];
        rb[p] = v;
      }
    }
  };

  auto store_tiles = [&](int buf) {
    double* a_s = sA + buf * BK * LDA_S;
    double* b_s = sB + buf * BK * LDB_S;
    if (a_mcontig) {
      const int i = tid % BM, kq = tid / BM;
      constexpr int KSTEP = 256 / BM;
#pragma unroll
      for (int p = 0; p < A_PER_THREAD; ++p) a_s[(kq + p * KSTEP) * LDA_S + i] = ra[p];
    } else {
      const int k = tid % BK, iq = tid / BK;
      constexpr int ISTEP = 256 / BK;
#pragma unroll
      for (int p = 0; p < A_PER_THREAD; ++p) a_s[k * LDA_S + iq + p * ISTEP] = ra[p];
    }
    if (b_ncontig) {
      const int j = tid % BN, kq = tid / BN;
      constexpr int KSTEP = 256 / BN;
#pragma unroll
      for (int p = 0; p < B_PER_THREAD; ++p) b_s[(kq + p * KSTEP) * LDB_S + j] = rb[p];
    } else {
      const int k = tid % BK, jq = tid / BK;
      constexpr int JSTEP = 256 / BK;
#pragma unroll
      for (int p = 0; p < B_PER_THREAD; ++p) b_s[k * LDB_S + jq + p * JSTEP] = rb[p];
    }
  };

  d4 acc[NT][MT];
  const int fr = lane & 15, fk = lane >> 4;

  // C(row = m0 + wm*WM + mi*16 + fr, col = n0 + wn*WN + ni*16 + fk + 4r) <-> acc[ni][mi][r]
  double* __restrict__ C = D.c + (split_k > 1 ? (long long)slice * D.split_stride : 0LL);
  const double alpha = D.alpha;
  const double beta = split_k > 1 ? 0.0 : D.beta;
  const long long ldc = D.ldc;
  const int* __restrict__ jidx = D.c_jidx;
  const bool lower = D.lower_only != 0;
  const int roff = D.row_off, coff = D.col_off;

  load_tiles(k_begin);
  // beta * C goes straight into the accumulators (scaled by 1/alpha), its loads in flight together with
  // the first operand tiles: the epilogue is then store-only.
  if (beta != 0.0 && alpha != 0.0) {
    const double scale = beta / alpha;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = n0 + wn * WN + ni * 16 + fk + 4 * r;
        const long long dcol = (jidx && col < N) ? (long long)jidx[col] : (long long)col;
        const double* ccol = C + dcol * ldc;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int row = m0 + wm * WM + mi * 16 + fr;
          double v = 0.0;
          if (col < N && row < M && !(lower && (row + roff) < (col + coff))) v = ccol[row] * scale;
          acc[ni][mi][r] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = d4{0.0, 0.0, 0.0, 0.0};
  }
  store_tiles(0);
  __syncthreads();
  int buf = 0;
  for (int kt = k_begin; kt < k_end; kt += BK) {
    const bool has_next = kt + BK < k_end;
    if (has_next) load_tiles(kt + BK);
    const double* a_s = sA + buf * BK * LDA_S;
    const double* b_s = sB + buf * BK * LDB_S;
#pragma unroll
    for (int k4 = 0; k4 < BK / 4; ++k4) {
      double af[MT], bf[NT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) af[mi] = a_s[(k4 * 4 + fk) * LDA_S + wm * WM + mi * 16 + fr];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) bf[ni] = b_s[(k4 * 4 + fk) * LDB_S + wn * WN + ni * 16 + fr];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    if (has_next) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue (store only)
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = n0 + wn * WN + ni * 16 + fk + 4 * r;
      if (col >= N) continue;
      const long long dcol = jidx ? (long long)jidx[col] : (long long)col;
      double* ccol = C + dcol * ldc;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int row = m0 + wm * WM + mi * 16 + fr;
        if (row >= M) continue;
        if (lower && (row + roff) < (col + coff)) continue;
        ccol[row] = alpha * acc[ni][mi][r];
      }
    }
  }
}

}  // namespace

int launch_gemm_f64(sc_ctx* ctx, const GemmDesc* d_desc, int count, int max_m, int max_n, int tile,
                    int split_k, bool gather, bool tri) {
  if (count <= 0 || max_m <= 0 || max_n <= 0) return SC_OK;
  if (split_k < 1) split_k = 1;
  const int bm = tile == 0 ? 128 : 64;
  const int bn = (tile == 2 || tile == 3) ? 64 : 128;
  const int bk = tile == 3 ? 8 : 16;
  dim3 grid((unsigned)((max_m + bm - 1) / bm), (unsigned)((max_n + bn - 1) / bn), (unsigned)(count * split_k));
  const size_t lds = sizeof(double) * 2 * bk * ((size_t)(bm + PAD) + (bn + PAD));
  if (tri)      // triangular A operands: default tile only
    hipLaunchKernelGGL((k_gemm_f64<64, 64, 8, 4, false, true>), grid, dim3(256), lds, ctx->stream, d_desc, split_k);
  else if (gather)   // only the default tile is instantiated with gather support
    hipLaunchKernelGGL((k_gemm_f64<64, 64, 8, 4, true>), grid, dim3(256), lds, ctx->stream, d_desc, split_k);
  else if (tile == 1)
    hipLaunchKernelGGL((k_gemm_f64<64, 128, 16, 2, false>), grid, dim3(256), lds, ctx->stream, d_desc, split_k);
  else if (tile == 2)
    hipLaunchKernelGGL((k_gemm_f64<64, 64, 16, 4, false>), grid, dim3(256), lds, ctx->stream, d_desc, split_k);
  else if (tile == 3)
    hipLaunchKernelGGL((k_gemm_f64<64, 64, 8, 4, false>), grid, dim3(256), lds, ctx->stream, d_desc, split_k);
  else
    hipLaunchKernelGGL((k_gemm_f64<128, 128, 16, 2, false>), grid, dim3(256), lds, ctx->stream, d_desc, split_k);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

// ---- debug / tuning entry point (not part of the public C ABI) ----------------------------------------------
// Times `iters` launches of one GEMM shape on freshly allocated buffers.  mode: 0 = NN (A m x k col-major,
// B k x n col-major), 1 = NT lower-only (SYR2K shape: B stored n x k), 2 = TN (A stored k x m).
extern "C" int sc_dbg_gemm_bench(sc_ctx* ctx, int m, int n, int k, int mode, int tile, int split_k, int iters,
                                 int beta_one, double* ms_out, double* max_err_out) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t ea = (size_t)m * k, eb = (size_t)k * n, ec = (size_t)m * n * (split_k > 1 ? split_k : 1);
  double *a, *b, *c;
  GemmDesc* dd;
  SC_HIP(ctx, hipMalloc(&a, ea * 8));
  SC_HIP(ctx, hipMalloc(&b, eb * 8));
  SC_HIP(ctx, hipMalloc(&c, ec * 8));
  SC_HIP(ctx, hipMalloc(&dd, sizeof(GemmDesc)));
  std::vector<double> ha(ea), hb(eb), hc(ec, 0.0);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)((long long)(s % 2001) - 1000) / 1000.0; };
  for (auto& x : ha) x = rnd();
  for (auto& x : hb) x = rnd();
  SC_HIP(ctx, hipMemcpy(a, ha.data(), ea * 8, hipMemcpyHostToDevice));
  SC_HIP(ctx, hipMemcpy(b, hb.data(), eb * 8, hipMemcpyHostToDevice));
  SC_HIP(ctx, hipMemset(c, 0, ec * 8));
  GemmDesc D{};
  D.a = a; D.b = b; D.c = c; D.m = m; D.n = n; D.k = k; D.ldc = m;
  D.alpha = 1.0; D.beta = beta_one ? 1.0 : 0.0;
  if (mode == 0) { D.sa_i = 1; D.sa_k = m; D.sb_k = 1; D.sb_j = k; }
  if (mode == 1) { D.sa_i = 1; D.sa_k = m; D.sb_k = n; D.sb_j = 1; D.lower_only = 1; }
  if (mode == 2) { D.sa_i = k; D.sa_k = 1; D.sb_k = 1; D.sb_j = k; }
  D.split_stride = (long long)m * n;
  SC_HIP(ctx, hipMemcpy(dd, &D, sizeof(D), hipMemcpyHostToDevice));
  SC_TRY(launch_gemm_f64(ctx, dd, 1, m, n, tile, split_k));
  SC_HIP(ctx, hipStreamSynchronize(st));
  // spot check 64 entries against a host dot product (beta path: C was 0 before the first launch)
  SC_HIP(ctx, hipMemcpy(hc.data(), c, ec * 8, hipMemcpyDeviceToHost));
  double maxerr = 0.0;
  for (int t = 0; t < 64; ++t) {
    int i = (int)((t * 7919ull) % m), j = (int)((t * 104729ull) % n);
    if (mode == 1 && i < j) { int tmp = i; i = j; j = tmp; if (i >= m || j >= n) continue; }
    double ref = 0.0;
    for (int kk = 0; kk < k; ++kk) {
      const double av = mode == 2 ? ha[(size_t)i * k + kk] : ha[(size_t)kk * m + i];
      const double bv = mode == 1 ? hb[(size_t)kk * n + j] : hb[(size_t)j * k + kk];
      ref += av * bv;
    }
    double got = 0.0;
    for (int sl = 0; sl < (split_k > 1 ? split_k : 1); ++sl) got += hc[(size_t)sl * m * n + (size_t)j * m + i];
    maxerr = fmax(maxerr, fabs(got - ref));
  }
  if (max_err_out) *max_err_out = maxerr;
  hipEvent_t e0, e1;
  SC_HIP(ctx, hipEventCreate(&e0));
  SC_HIP(ctx, hipEventCreate(&e1));
  SC_HIP(ctx, hipEventRecord(e0, st));
  for (int it = 0; it < iters; ++it) SC_TRY(launch_gemm_f64(ctx, dd, 1, m, n, tile, split_k));
  SC_HIP(ctx, hipEventRecord(e1, st));
  SC_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  SC_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  if (ms_out) *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(dd);
  return SC_OK;
}
