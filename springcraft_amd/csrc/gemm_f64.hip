// float64 GEMM on the CDNA4 matrix cores — see gemm_f64.h for the contract.
// (Round 1's stride-agnostic 64 x 64 x 8 kernel, kept behind switches through round 2 for A/B runs, is gone: every launch
// of the solver states its operand layout and runs k_gemm2.)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "gemm_f64.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// ================================================================================================================
// k_gemm2: the MFMA-paced kernel behind every launch.
//
// The f64 matrix pipe issues one v_mfma_f64_16x16x4_f64 per 64 cycles per SIMD at the 2.4 GHz the chip holds under this
// load (profiles/r02_probe_clock.txt: 78 TFLOP/s, the datasheet rate), so what a GEMM has to do is keep that pipe
// issuing: long MFMA runs between barriers and nothing in the loop that can stall it.
//   * block tile BM x BN (128 x 128, 128 x 64 or 64 x 64), K step BK = 16, 4 waves as 2 x 2: a wave owns (BM/2) x (BN/2),
//     i.e. up to 4 x 4 MFMA tiles = 16 independent accumulators and 64 MFMAs (4096 pipe cycles) per K step against
//     8 fragment reads per 16 MFMAs;
//   * two workgroups per CU (<= 256 VGPRs, <= 74 KB LDS each): the partner's MFMAs fill the barrier / staging gaps;
//   * operand staging global -> registers -> LDS with ONE register set, written to the other LDS buffer after the MFMA
//     run of the current K step and re-issued for K step t + 2 right away; the loads are unconditional (clamped
//     indices) and masked by a select, so the loop holds no branch and no load the compiler must wait for early;
//   * fragments of k-step k4 + 1 are read while the MFMAs of k4 issue (register double buffer);
//   * LDS images are conflict-free for both the staging writes and the fragment reads, for either source layout:
//       source contiguous along the tile's own axis x (m for A, n for B):  [k][X + 16] doubles
//       source contiguous along k:                                      [k / 2][X + 1][2] doubles
//   * the MFMA is fed swapped (A operand from the B tile) so that 16 consecutive lanes of an accumulator register are
//     16 consecutive rows of the column-major C: C loads / stores are 128-byte segments.
template <int X, bool XCONTIG>
struct StageImage {
  // doubles per buffer
  static constexpr int kSize = XCONTIG ? 16 * (X + 16) : 8 * (X + 1) * 2;
  __device__ static __forceinline__ int at(int x, int k) {
    return XCONTIG ? k * (X + 16) + x : ((k >> 1) * (X + 1) + x) * 2 + (k & 1);
  }
};

// Diagnostic build (-DGEMM_STAMPS): shader cycles of the prologue / K loop / epilogue of every wave, summed
// (sc_dbg_gemm_stamps, tools/gemm_stamps.py); no stamp executes in the normal build.
#ifdef GEMM_STAMPS
__device__ unsigned long long g_gemm_stamps[8];
constexpr int kTraceMax = 1 << 16;
__device__ unsigned long long g_gemm_trace[6 * kTraceMax];
__device__ unsigned int g_gemm_trace_n;
#define GEMM_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");
#else
#define GEMM_STAMP(var)
#endif
typedef const double __attribute__((address_space(1)))* gptr_c;   // loads through these are global_load, not flat_load
typedef double __attribute__((address_space(1)))* gptr;             // (a flat access also counts on lgkmcnt: the barrier's wait would wait for it)

// GATHER (layout "A m-contiguous, B k-contiguous" only; the merges of the tridiagonal divide & conquer): A's k axis and B's
// k axis go through the records' index lists, C's columns through c_jidx.  The indices of a K step are loaded one K
// step before the operands that need them, so the dependent load pair is never waited for.
template <int BM, int BN, bool AM, bool BNC, bool TRI, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void k_gemm2(const GemmDesc* __restrict__ descs, int split_k, int lower_grid,
                                                  int gx_, int gy_, int gz_) {
  static_assert(!GATHER || (AM && !BNC && !TRI), "gather lists: A m-contiguous, B k-contiguous, no triangular mask");
  constexpr int BK = 16;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int MT = WM / 16, NT = WN / 16;
  using ImgA = StageImage<BM, AM>;
  using ImgB = StageImage<BN, BNC>;
  constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;

  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* sA = smem;                       // [2][ImgA::kSize]
  double* sB = smem + 2 * ImgA::kSize;     // [2][ImgB::kSize]

  // lower_grid: the launch only holds the tiles on and below the diagonal (all records lower_only with row_off =
  // col_off = 0; grid (tiles, records)): row tile b has (BM / BN)(b + 1) column tiles.  A rectangular grid would spend
  // half of its workgroup dispatches on tiles that return at once, which costs occupancy (1.8 of 3 waves per SIMD).
  int t_x = blockIdx.x, t_y = blockIdx.y, z = blockIdx.z;
  if (lower_grid) {
    constexpr int r = BM / BN;
    const long long t = blockIdx.x;
    int b = (int)((sqrt(1.0 + 8.0 * (double)t / r) - 1.0) * 0.5);
    while ((long long)r * (b + 1) * (b + 2) / 2 <= t) ++b;
    while ((long long)r * b * (b + 1) / 2 > t) --b;
    t_x = b; t_y = (int)(t - (long long)r * b * (b + 1) / 2); z = blockIdx.y;
  }
  // lower_grid == 2 (triangular-operand launches, grid (tiles x * tiles y * records), tiles y, tiles x as kernel
  // arguments gy_, gx_): the K range of a tile grows (a_tri 1) or shrinks (a_tri 2) with its row, so the tiles are dealt
  // longest first across all records -- the launch then ends with the shortest tiles instead of with one record's
  // longest.
  if (TRI && lower_grid == 2) {
    z = blockIdx.x % gz_;
    const int rest = blockIdx.x / gz_;
    t_y = rest % gy_;
    t_x = rest / gy_;
  }
  // lower_grid == 3 (few row tiles, e.g. the 256-row products V^T Z of the back-transformation): the gx_ row tiles that
  // read the same column panel of B run as consecutive workgroups of ONE XCD (workgroups are dealt round-robin to the 8
  // XCDs), so that panel crosses the fabric once and is served from that XCD's L2 afterwards.
  if (!TRI && lower_grid == 3) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = (slot / gx_) * 8 + xcd;
    t_x = slot % gx_;
    t_y = group % gy_;
    z = group / gy_;
    if (z >= gz_) return;
  }
  const int slice = split_k > 1 ? z % split_k : 0;
  const GemmDesc& D = descs[split_k > 1 ? z / split_k : z];
  if (TRI && lower_grid == 2 && D.a_tri == 1) t_x = gx_ - 1 - t_x;
  const int M = D.m, N = D.n, K = D.k;
  const int m0 = t_x * BM, n0 = t_y * BN;
  if (m0 >= M || n0 >= N) return;   // K == 0 still runs: it stores beta*C (zeros for beta = 0)
  if (D.lower_only && (m0 + BM - 1 + D.row_off) < (n0 + D.col_off)) return;

  GEMM_STAMP(t_start)
  int k_begin = 0, k_end = K;
  if (split_k > 1) {
    const int chunk = ((K + split_k - 1) / split_k + BK - 1) / BK * BK;
    k_begin = slice * chunk;
    k_end = min(K, k_begin + chunk);
  }
  const int tri = TRI ? D.a_tri : 0;
  if (TRI && tri == 1) k_end = min(k_end, m0 + BM);
  if (TRI && tri == 2) k_begin = max(k_begin, m0 / BK * BK);

  const char* Ab = (const char*)D.a;
  const char* Bb = (const char*)D.b;
  const long long sa = AM ? D.sa_k : D.sa_i;   // the stride that is not 1
  const long long sb = BNC ? D.sb_k : D.sb_j;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 15, fk = lane >> 4;

  // ---- staging maps: element p of this thread is (x, k) of the tile
  //   x-contiguous source: x = tid % X, k = tid / X + p * (256 / X);   k-contiguous: k = tid % 16, x = tid / 16 + 16 p
  const int ax = AM ? tid % BM : tid / BK, ak = AM ? tid / BM : tid % BK;
  const int bx = BNC ? tid % BN : tid / BK, bk = BNC ? tid / BN : tid % BK;
  constexpr int A_KSTEP = AM ? 256 / BM : 0, A_XSTEP = AM ? 0 : 256 / BK;
  constexpr int B_KSTEP = BNC ? 256 / BN : 0, B_XSTEP = BNC ? 0 : 256 / BK;

  // Byte offsets that do not change along K (rows / columns clamped into the matrix: what lands in the out-of-range
  // part of a tile is never stored).  x-contiguous: one offset, element p adds the uniform p * KSTEP * stride;
  // k-contiguous: element p is row (x0 + 16 p + tid / 16), its offset relative to the uniform row x0 + 16 p.
  long long a_thr = 0, b_thr = 0;
  int a_off[AM ? 1 : A_PER], b_off[BNC ? 1 : B_PER];
  if (AM) a_thr = ((long long)min(m0 + ax, M - 1) + (long long)ak * sa) * 8;
  else {
#pragma unroll
    for (int p = 0; p < A_PER; ++p)
      a_off[p] = (int)(((long long)(min(m0 + p * A_XSTEP + ax, M - 1) - (m0 + p * A_XSTEP)) * sa + ak) * 8);
  }
  if (BNC) b_thr = ((long long)min(n0 + bx, N - 1) + (long long)bk * sb) * 8;
  else {
#pragma unroll
    for (int p = 0; p < B_PER; ++p)
      b_off[p] = (int)(((long long)(min(n0 + p * B_XSTEP + bx, N - 1) - (n0 + p * B_XSTEP)) * sb + bk) * 8);
  }

  double ra[A_PER], rb[B_PER];
  // gather lists: this thread's k indices of the K step whose operands are loaded next
  const int* __restrict__ a_kidx = GATHER ? D.a_kidx : nullptr;
  const int* __restrict__ b_kidx = GATHER ? D.b_kidx : nullptr;
  int ia[GATHER ? A_PER : 1], ib = 0;
  long long b_col[GATHER ? B_PER : 1];   // byte offsets of this thread's B columns
  if (GATHER) {
#pragma unroll
    for (int p = 0; p < B_PER; ++p) b_col[p] = (long long)min(n0 + bx + p * B_XSTEP, N - 1) * sb * 8;
  }
  auto i_load = [&](int kt) {
    if (GATHER) {
#pragma unroll
      for (int p = 0; p < A_PER; ++p) {
        const int k = min(kt + ak + p * A_KSTEP, K - 1);
        ia[p] = a_kidx ? a_kidx[k] : k;
      }
      const int kb = min(kt + bk, K - 1);
      ib = b_kidx ? b_kidx[kb] : kb;
    }
  };
  // Raw loads, always in bounds; the masks are applied when the registers go to LDS, one MFMA run later, so that
  // nothing in between waits for the loads.  Full K steps (no clamp on k): thread base + uniform offsets only.
  auto g_load = [&](int kt) {
    if (GATHER) {
      const char* arow = Ab + (long long)min(m0 + ax, M - 1) * 8;
#pragma unroll
      for (int p = 0; p < A_PER; ++p) ra[p] = *(gptr_c)(arow + (long long)ia[p] * sa * 8);
#pragma unroll
      for (int p = 0; p < B_PER; ++p) rb[p] = *(gptr_c)(Bb + b_col[p] + (long long)ib * 8);
      i_load(kt + BK);
    } else if (kt + BK <= K) {
      if (AM) {
        const char* t0 = Ab + a_thr + (long long)kt * sa * 8;
#pragma unroll
        for (int p = 0; p < A_PER; ++p) ra[p] = *(gptr_c)(t0 + (long long)(p * A_KSTEP) * sa * 8);
      } else {
#pragma unroll
        for (int p = 0; p < A_PER; ++p)
          ra[p] = *(gptr_c)(Ab + ((long long)(m0 + p * A_XSTEP) * sa + kt) * 8 + a_off[p]);
      }
      if (BNC) {
        const char* t0 = Bb + b_thr + (long long)kt * sb * 8;
#pragma unroll
        for (int p = 0; p < B_PER; ++p) rb[p] = *(gptr_c)(t0 + (long long)(p * B_KSTEP) * sb * 8);
      } else {
#pragma unroll
        for (int p = 0; p < B_PER; ++p)
          rb[p] = *(gptr_c)(Bb + ((long long)(n0 + p * B_XSTEP) * sb + kt) * 8 + b_off[p]);
      }
    } else {   // last, partial K step: k clamped as well
#pragma unroll
      for (int p = 0; p < A_PER; ++p) {
        const int x = ax + p * A_XSTEP, k = kt + ak + p * A_KSTEP;
        const long long gi = min(m0 + x, M - 1), gk = min(k, K - 1);
        ra[p] = *(gptr_c)(Ab + (AM ? gi + gk * sa : gi * sa + gk) * 8);
      }
#pragma unroll
      for (int p = 0; p < B_PER; ++p) {
        const int x = bx + p * B_XSTEP, k = kt + bk + p * B_KSTEP;
        const long long gj = min(n0 + x, N - 1), gk = min(k, K - 1);
        rb[p] = *(gptr_c)(Bb + (BNC ? gj + gk * sb : gj * sb + gk) * 8);
      }
    }
  };
  auto s_write = [&](int buf, int kt) {
    double* a_s = sA + buf * ImgA::kSize;
    double* b_s = sB + buf * ImgB::kSize;
    // masks only where the K step leaves [k_begin, k_end) or meets the diagonal of a triangular operand
    bool mask = kt + BK > k_end || kt < k_begin;
    if (TRI) mask = mask || (tri == 1 && kt + BK - 1 > m0) || (tri == 2 && kt <= m0 + BM - 1);
    if (mask) {
#pragma unroll
      for (int p = 0; p < A_PER; ++p) {
        const int x = ax + p * A_XSTEP, k = kt + ak + p * A_KSTEP;
        bool ok = k < k_end && k >= k_begin;
        if (TRI) {
          const int gi = m0 + x;
          ok = ok && (tri == 0 || (tri == 1 ? gi >= k : k > gi));
        }
        a_s[ImgA::at(x, ak + p * A_KSTEP)] = ok ? ra[p] : 0.0;
      }
#pragma unroll
      for (int p = 0; p < B_PER; ++p) {
        const int k = kt + bk + p * B_KSTEP;
        b_s[ImgB::at(bx + p * B_XSTEP, bk + p * B_KSTEP)] = (k < k_end && k >= k_begin) ? rb[p] : 0.0;
      }
    } else {
#pragma unroll
      for (int p = 0; p < A_PER; ++p) a_s[ImgA::at(ax + p * A_XSTEP, ak + p * A_KSTEP)] = ra[p];
#pragma unroll
      for (int p = 0; p < B_PER; ++p) b_s[ImgB::at(bx + p * B_XSTEP, bk + p * B_KSTEP)] = rb[p];
    }
  };

  d4 acc[NT][MT];
  // C(row = m0 + wm*WM + mi*16 + fr, col = n0 + wn*WN + ni*16 + fk + 4r) <-> acc[ni][mi][r]
  gptr C = (gptr)(D.c + (split_k > 1 ? (long long)slice * D.split_stride : 0LL));
  const double alpha = D.alpha;
  const double beta = split_k > 1 ? 0.0 : D.beta;
  const long long ldc = D.ldc;
  const bool lower = D.lower_only != 0, lower2 = D.lower_only == 2;
  const int roff = D.row_off, coff = D.col_off;

  // beta * C goes straight into the accumulators (scaled by 1/alpha); its loads fly together with the first tiles
  const bool with_c = beta != 0.0 && alpha != 0.0;
  if (with_c) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = n0 + wn * WN + ni * 16 + fk + 4 * r;
        const int colc = min(col, N - 1);
        gptr_c ccol = C + (long long)((GATHER && D.c_jidx) ? D.c_jidx[colc] : colc) * ldc;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int row = m0 + wm * WM + mi * 16 + fr;
          acc[ni][mi][r] = ccol[min(row, M - 1)];
        }
      }
  }
  if (GATHER && K > 0) i_load(k_begin);
  if (k_begin < k_end) g_load(k_begin);
  else {
#pragma unroll
    for (int p = 0; p < A_PER; ++p) ra[p] = 0.0;
#pragma unroll
    for (int p = 0; p < B_PER; ++p) rb[p] = 0.0;
  }
  s_write(0, k_begin);
  if (k_begin + BK < k_end) g_load(k_begin + BK);
  if (with_c) {
    const double scale = beta / alpha;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[ni][mi] *= scale;
  } else {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[ni][mi] = d4{0.0, 0.0, 0.0, 0.0};
  }
  __syncthreads();

  // K loop.  Iteration t: the registers hold K step t + 1 (loaded one MFMA run ago): write them to the other buffer
  // (all waves left it at the barrier that ended iteration t - 1), re-issue the loads for K step t + 2, then the MFMA
  // run of K step t with the fragments of k4 + 1 read while the MFMAs of k4 issue.
  GEMM_STAMP(t_loop)
  int buf = 0;
  for (int kt = k_begin; kt < k_end; kt += BK) {
    if (kt + BK < k_end) s_write(buf ^ 1, kt + BK);
    if (kt + 2 * BK < k_end) g_load(kt + 2 * BK);
    const double* a_s = sA + buf * ImgA::kSize + ImgA::at(wm * WM + fr, fk);
    const double* b_s = sB + buf * ImgB::kSize + ImgB::at(wn * WN + fr, fk);
    double af[2][MT], bf[2][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) af[0][mi] = a_s[ImgA::at(mi * 16, 0)];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) bf[0][ni] = b_s[ImgB::at(ni * 16, 0)];
#pragma unroll
    for (int k4 = 0; k4 < BK / 4; ++k4) {
      const int cur = k4 & 1, nxt = cur ^ 1;
      if (k4 + 1 < BK / 4) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) af[nxt][mi] = a_s[ImgA::at(mi * 16, (k4 + 1) * 4)];
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) bf[nxt][ni] = b_s[ImgB::at(ni * 16, (k4 + 1) * 4)];
      }
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[cur][ni], af[cur][mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();
    buf ^= 1;
  }

  GEMM_STAMP(t_epi)
  // ---- epilogue (store only)
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = n0 + wn * WN + ni * 16 + fk + 4 * r;
      if (col >= N) continue;
      gptr ccol = C + (long long)((GATHER && D.c_jidx) ? D.c_jidx[col] : col) * ldc;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int row = m0 + wm * WM + mi * 16 + fr;
        if (row >= M) continue;
        // (lower_only = 2: also the first super-diagonal entry of every even row, which k_symm3 reads -- symm3.hip)
        if (lower && (lower2 ? ((row + roff) | 1) : (row + roff)) < (col + coff)) continue;
        ccol[row] = alpha * acc[ni][mi][r];
      }
    }
  }
#ifdef GEMM_STAMPS
  {
    GEMM_STAMP(t_end)
    if (lane == 0) {
      atomicAdd(&g_gemm_stamps[0], t_loop - t_start);
      atomicAdd(&g_gemm_stamps[1], t_epi - t_loop);
      atomicAdd(&g_gemm_stamps[2], t_end - t_epi);
      atomicAdd(&g_gemm_stamps[3], 1ull);
    }
    if (tid == 0) {
      const unsigned slot = atomicAdd(&g_gemm_trace_n, 1u);
      if (slot < (unsigned)kTraceMax) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* r = g_gemm_trace + 6ull * slot;
        r[0] = ((unsigned long long)xcc << 32) | hwid;
        r[1] = t_start; r[2] = t_loop; r[3] = t_epi; r[4] = t_end;
        r[5] = blockIdx.x + (unsigned long long)gridDim.x * (blockIdx.y + (unsigned long long)gridDim.y * blockIdx.z);
      }
    }
  }
#endif
}

// One instantiation per (block tile, layout, triangular A); its dynamic LDS size is raised above the 64 KB default once.
template <int BM, int BN, bool AM, bool BNC, bool TRI, bool GATHER = false>
void launch_gemm2_inst(hipStream_t st, dim3 grid, const GemmDesc* d_desc, int split_k, bool lower = false) {
  constexpr size_t lds = sizeof(double) * 2 * (size_t)(StageImage<BM, AM>::kSize + StageImage<BN, BNC>::kSize);
  // (per device: sc_raise_dyn_lds; a refusal surfaces as the launch error the caller checks)
  (void)sc_raise_dyn_lds(reinterpret_cast<const void*>(&k_gemm2<BM, BN, AM, BNC, TRI, GATHER>), (int)lds);
  if (lower) {   // square launch: only the tiles of the lower triangle
    static_assert(BM % BN == 0, "row tiles are whole multiples of column tiles");
    const unsigned tiles = (unsigned)((long long)(BM / BN) * grid.x * (grid.x + 1) / 2);
    grid = dim3(tiles, grid.z, 1);
  }
  int mode = lower ? 1 : 0;
  const int gx = (int)grid.x, gy = (int)grid.y, gz = (int)grid.z;
  static const bool no_balance = getenv("SPRINGCRAFT_GEMM_NO_BALANCE") != nullptr;
  if (TRI && !lower && !no_balance && (long long)grid.x * grid.y * grid.z < 0x7fffffffLL) {
    mode = 2;   // longest tiles first, over all records
    grid = dim3(grid.x * grid.y * grid.z, 1, 1);
  }
  static const bool no_pair = getenv("SPRINGCRAFT_GEMM_NO_PAIR") != nullptr;
  if (!TRI && !lower && !no_pair && gx >= 2 && gx <= 4 && (long long)(gy * gz + 8) * gx < 0x7fffffffLL) {
    mode = 3;   // the row tiles of one B panel on one XCD
    grid = dim3((unsigned)(((long long)gy * gz + 7) / 8 * 8 * gx), 1, 1);
  }
  hipLaunchKernelGGL((k_gemm2<BM, BN, AM, BNC, TRI, GATHER>), grid, dim3(256), lds, st, d_desc, split_k, mode, gx, gy, gz);
}

template <int BM, int BN, bool TRI>
void launch_gemm2_layout(hipStream_t st, dim3 grid, const GemmDesc* d_desc, int split_k, int layout, bool lower) {
  if (layout == kGemmAmBn) {
    if constexpr (!TRI) launch_gemm2_inst<BM, BN, true, true, false>(st, grid, d_desc, split_k, lower);
  } else if (layout == kGemmAmBk) {
    launch_gemm2_inst<BM, BN, true, false, TRI>(st, grid, d_desc, split_k);
  } else {
    // both operands k-contiguous: 128 x 128 would spill (8 + 8 row offsets on top of 128 accumulator registers)
    if constexpr (!(BM == 128 && BN == 128)) launch_gemm2_inst<BM, BN, false, false, TRI>(st, grid, d_desc, split_k);
  }
}

}  // namespace

namespace {
int g_gemm2_force_tile = 0;   // debugging (sc_dbg_gemm_bench): 1 = 128 x 128, 2 = 128 x 64, 3 = 64 x 64
// block tile of k_gemm2: 128-wide in a dimension when the problem is at least that wide there and the launch still has
// >= 2 workgroups per CU (lower-only launches: about half of the square grid does work)
void gemm2_tile(const sc_ctx* ctx, int count, int max_m, int max_n, int split_k, int layout, int* bm, int* bn) {
  static const int env_force = [] { const char* e = getenv("SPRINGCRAFT_GEMM2_TILE"); return e ? atoi(e) : 0; }();
  const int force = g_gemm2_force_tile ? g_gemm2_force_tile : env_force;
  if (force == 1 && layout != kGemmAkBk) { *bm = 128; *bn = 128; return; }
  if (force == 1) { *bm = 128; *bn = 64; return; }
  if (force == 2) { *bm = 128; *bn = 64; return; }
  if (force == 3) { *bm = 64; *bn = 64; return; }
  const long long cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  auto blocks = [&](int a, int b) {
    return (long long)((max_m + a - 1) / a) * ((max_n + b - 1) / b) * count * split_k;
  };
  *bm = 64; *bn = 64;
  if (max_m > 64 && blocks(128, 64) >= 2 * cus) *bm = 128;
  if (*bm == 128 && max_n > 64 && layout != kGemmAkBk && blocks(128, 128) >= 3 * cus) *bn = 128;
}
}  // namespace

int launch_gemm_f64(sc_ctx* ctx, const GemmDesc* d_desc, int count, int max_m, int max_n, int tile,
                    int split_k, bool gather, bool tri, int layout, bool lower_grid) {
  static const bool no_lower = getenv("SPRINGCRAFT_GEMM_NO_LOWER_GRID") != nullptr;
  const bool lower = lower_grid && !no_lower && layout == kGemmAmBn && !tri && !gather && split_k <= 1 && max_m == max_n;
  if (count <= 0 || max_m <= 0 || max_n <= 0) return SC_OK;
  if (split_k < 1) split_k = 1;
  if (split_k > 65535) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "split_k %d too large", split_k);
  // grid.z (records x K slices) is limited to 65535: longer record lists go out in chunks (the D&C merges of a batch of
  // some thousand matrices get there)
  if ((long long)count * split_k > 65535) {
    const int max_rec = 65535 / split_k;
    for (int r0 = 0; r0 < count; r0 += max_rec)
      SC_TRY(launch_gemm_f64(ctx, d_desc + r0, std::min(max_rec, count - r0), max_m, max_n, tile, split_k, gather, tri,
                             layout, lower_grid));
    return SC_OK;
  }
  hipStream_t st = ctx->stream;
  if (gather && !tri && layout == kGemmAmBk) {
    // merges of the divide & conquer: 128 x 64 tiles while they fill the chip, else 64 x 64
    const long long cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const long long gz = (long long)count * split_k;
    if (gz > 65535) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "GEMM launch with %lld records x slices (max 65535)", gz);
    const bool big = max_m > 64 && (long long)((max_m + 127) / 128) * ((max_n + 63) / 64) * gz >= 2 * cus;
    const int bm = big ? 128 : 64;
    dim3 grid((unsigned)((max_m + bm - 1) / bm), (unsigned)((max_n + 63) / 64), (unsigned)gz);
    if (big) launch_gemm2_inst<128, 64, true, false, false, true>(st, grid, d_desc, split_k);
    else launch_gemm2_inst<64, 64, true, false, false, true>(st, grid, d_desc, split_k);
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
  }
  if (!gather && layout >= 0 && layout <= 2 && !(tri && layout == kGemmAmBn)) {
    int bm, bn;
    gemm2_tile(ctx, count, max_m, max_n, split_k, layout, &bm, &bn);
    const long long gz = (long long)count * split_k;
    if (gz > 65535) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "GEMM launch with %lld records x slices (max 65535)", gz);
    dim3 grid((unsigned)((max_m + bm - 1) / bm), (unsigned)((max_n + bn - 1) / bn), (unsigned)gz);
    if (bm == 128 && bn == 128) {
      if (tri) launch_gemm2_layout<128, 128, true>(st, grid, d_desc, split_k, layout, lower);
      else launch_gemm2_layout<128, 128, false>(st, grid, d_desc, split_k, layout, lower);
    } else if (bm == 128) {
      if (tri) launch_gemm2_layout<128, 64, true>(st, grid, d_desc, split_k, layout, lower);
      else launch_gemm2_layout<128, 64, false>(st, grid, d_desc, split_k, layout, lower);
    } else {
      if (tri) launch_gemm2_layout<64, 64, true>(st, grid, d_desc, split_k, layout, lower);
      else launch_gemm2_layout<64, 64, false>(st, grid, d_desc, split_k, layout, lower);
    }
    SC_HIP(ctx, hipGetLastError());
    return SC_OK;
  }
  (void)tile;
  return sc_set_error(ctx, SC_ERR_INVALID_ARG, "GEMM launch without a supported operand layout (layout %d, gather %d, tri %d)",
                      layout, (int)gather, (int)tri);
}

extern "C" int sc_dbg_gemm_stamps(unsigned long long* out4, int reset) {
#ifdef GEMM_STAMPS
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_gemm_stamps), 32) != hipSuccess) return 5;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), z, 64) != hipSuccess) return 5;
  }
  return 0;
#else
  (void)out4; (void)reset;
  return 1;
#endif
}

extern "C" int sc_dbg_gemm_trace(unsigned long long* out, int max_records, int* count, int reset) {
#ifdef GEMM_STAMPS
  unsigned int n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_gemm_trace_n), 4) != hipSuccess) return 5;
  const int m = (int)std::min<unsigned>(std::min<unsigned>(n, (unsigned)kTraceMax), (unsigned)std::max(0, max_records));
  if (m > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_trace), (size_t)m * 48) != hipSuccess) return 5;
  if (count) *count = m;
  if (reset) {
    n = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace_n), &n, 4) != hipSuccess) return 5;
  }
  return 0;
#else
  (void)out; (void)max_records; (void)count; (void)reset;
  return 1;
#endif
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
  // tile 10 .. 13: k_gemm2 (10: automatic block tile, 11: 128 x 128, 12: 128 x 64, 13: 64 x 64)
  if (tile < 10 || tile > 13) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "tile id %d (10 .. 13)", tile);
  const int layout = mode == 0 ? kGemmAmBk : (mode == 1 ? kGemmAmBn : kGemmAkBk);
  g_gemm2_force_tile = tile - 10;
  SC_TRY(launch_gemm_f64(ctx, dd, 1, m, n, tile, split_k, false, false, layout, mode == 1 && m == n));
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
  for (int it = 0; it < iters; ++it) SC_TRY(launch_gemm_f64(ctx, dd, 1, m, n, tile, split_k, false, false, layout, mode == 1 && m == n));
  SC_HIP(ctx, hipEventRecord(e1, st));
  SC_HIP(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  SC_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
  if (ms_out) *ms_out = ms / iters;
  g_gemm2_force_tile = 0;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(dd);
  return SC_OK;
}

// ---- debug entry point for the GPU unit tests of the launch paths (tests/test_gemm_gpu.py): ONE launch of one record
// on host data.  a, b as sc_dbg_gemm_bench lays them out for `mode`; c: m x n column-major, in / out (with split_k > 1
// the slices are summed into c on the host and beta must be 0).  lower_grid as launch_gemm_f64 takes it.
extern "C" int sc_dbg_gemm_host(sc_ctx* ctx, const double* a, const double* b, double* c, int m, int n, int k, int mode,
                                int tile, int split_k, double alpha, double beta, int lower_grid) {
  if (!ctx || !a || !b || !c || m <= 0 || n <= 0 || k < 0 || split_k < 1) return SC_ERR_INVALID_ARG;
  if (split_k > 1 && beta != 0.0) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t ea = (size_t)m * std::max(k, 1), eb = (size_t)std::max(k, 1) * n, ec = (size_t)m * n * split_k;
  char* base = nullptr;
  SC_HIP(ctx, hipMalloc((void**)&base, (ea + eb + ec) * 8 + 256));
  double* da = (double*)base;
  double* db = da + ea;
  double* dc = db + eb;
  GemmDesc* dd = (GemmDesc*)(dc + ec);
  int rc = SC_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == SC_OK) rc = sc_set_error(ctx, SC_ERR_HIP, "%s", hipGetErrorString(e)); };
  if (k > 0) { fail(hipMemcpy(da, a, (size_t)m * k * 8, hipMemcpyHostToDevice)); fail(hipMemcpy(db, b, (size_t)k * n * 8, hipMemcpyHostToDevice)); }
  for (int sl = 0; sl < split_k; ++sl) fail(hipMemcpy(dc + (size_t)sl * m * n, c, (size_t)m * n * 8, hipMemcpyHostToDevice));
  GemmDesc D{};
  D.a = da; D.b = db; D.c = dc; D.m = m; D.n = n; D.k = k; D.ldc = m;
  D.alpha = alpha; D.beta = beta;
  if (mode == 0) { D.sa_i = 1; D.sa_k = m; D.sb_k = 1; D.sb_j = k; }
  if (mode == 1) { D.sa_i = 1; D.sa_k = m; D.sb_k = n; D.sb_j = 1; D.lower_only = 1; }
  if (mode == 2) { D.sa_i = k; D.sa_k = 1; D.sb_k = 1; D.sb_j = k; }
  D.split_stride = (long long)m * n;
  fail(hipMemcpy(dd, &D, sizeof(D), hipMemcpyHostToDevice));
  if (rc == SC_OK) {
    const int layout = mode == 0 ? kGemmAmBk : (mode == 1 ? kGemmAmBn : kGemmAkBk);
    if (tile < 10 || tile > 13) {
      rc = sc_set_error(ctx, SC_ERR_INVALID_ARG, "tile id %d (10 .. 13)", tile);
    } else {
      g_gemm2_force_tile = tile - 10;
      rc = launch_gemm_f64(ctx, dd, 1, m, n, tile, split_k, false, false, layout, lower_grid != 0);
      g_gemm2_force_tile = 0;
    }
    fail(hipStreamSynchronize(st));
  }
  if (rc == SC_OK) {
    std::vector<double> h(ec);
    fail(hipMemcpy(h.data(), dc, ec * 8, hipMemcpyDeviceToHost));
    if (rc == SC_OK) {
      if (split_k == 1) std::copy(h.begin(), h.end(), c);
      else
        for (size_t i = 0; i < (size_t)m * n; ++i) {
          double sum = 0.0;
          for (int sl = 0; sl < split_k; ++sl) sum += h[(size_t)sl * m * n + i];
          c[i] = sum;
        }
    }
  }
  (void)hipFree(base);
  return rc;
}
