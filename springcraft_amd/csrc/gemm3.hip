// k_gemm3: the role-split, persistent float64 GEMM for the short-K updates of the eigensolver (round 5).
//
//     C = beta C + A B,  beta in {0, 1},  the same (m, n, k) for every record of the launch        (gemm_f64.h: GemmDesc)
//
// Why a second kernel (tools/probe_gemm3.hip is its probe, profiles/r05_probe_gemm3_*.txt the measurements): in k_gemm2 a
// workgroup's life is "load C, K loop, store C", and on this chip the memory instructions of one workgroup do not hide
// behind the MFMAs of its neighbour -- a SIMD that is busy with f64 MFMAs issues other waves' vector instructions at a
// fraction of their rate -- so at K = 256 the C traffic costs as much time again as a third of the MFMAs.  Here the
// waves of a workgroup have fixed roles and the MFMA waves never touch global memory:
//   * one workgroup per CU (16 waves), persistent: it walks the tiles wg, wg + #workgroups, ... of the launch (128 x 64
//     tiles, column by column, matrix by matrix);
//   * waves 0-3: one MFMA wave per SIMD, 2 x 2 over the tile (64 x 32 = 8 accumulator tiles each).  K steps of 16 out of
//     an LDS ring of three slots; the fragment reads run one k4 ahead of the MFMAs ACROSS the step barrier, so the
//     barrier sits between two runs of eight MFMAs whose operands are in registers;
//   * waves 4-11: operand loaders.  Wave (quarter q, parity p) fetches quarter q of the K steps g with g % 2 == p: six
//     1-KB LDS-DMA instructions (global_load_lds_dwordx4) behind ONE write of M0 -- a write to M0 waits for the wave's
//     LDS-DMA in flight, so a wave never has two groups in flight; the two parities keep two K steps in flight;
//   * waves 12-15: C waves.  Each owns 16 columns of a 72 KB LDS image of the C tile.  While a tile's K loop runs they
//     stream the NEXT tile's C into the image (LDS-DMA, a column per instruction); at the tile boundary the MFMA waves
//     SWAP accumulators and image (result out, next C in: LDS traffic only); during the next K loop the C waves store the
//     result from the image, a column (one 1-KB row-contiguous store) at a time, and refill the column's place;
//   * everything a helper wave computes per K step is scalar arithmetic: its VALU instructions would have to find issue
//     slots on a SIMD that an MFMA wave keeps busy (an integer division in a helper wave made the whole workgroup wait
//     ~1500 cycles at the step's barrier);
//   * alpha is 1: a sign is folded into an operand by the caller (flipping the sign bit of the B fragments in the K loop
//     cost 6 % of it: VALU instructions between MFMAs).
// Bit for bit the result is k_gemm2's: the same MFMA instruction, the k-steps of a dot product in the same order.
//
// LDS images (bytes; kRow = 1152 = 1 KB piece + 128 B so that consecutive pieces sit in opposite bank halves):
//   slot = 4 quarters x (4 A rows | 2 B pieces), quarter q = k-steps 4 q .. 4 q + 3 of the K step
//     A row k:  [m]                          1 KB = 128 rows of the tile, one LDS-DMA instruction
//     B piece, B n-contiguous ("NT"):  k-pair (2 rows) as [x / 16][k & 1][16]     one instruction per k-pair
//     B piece, B k-contiguous ("NN"):  8 columns x 16 k as [k / 2][x % 8][2]      one instruction per 8 columns; quarter q
//                                       holds columns 16 q .. 16 q + 15 (all 16 k of the step)
//     A k-contiguous ("TN"): pieces like those of a k-contiguous B: 8 rows x 16 k each, quarter q holds rows 32 q .. + 31
//   C image: column c at c * kRow, 128 rows each.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "gemm_f64.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int kRow = 1152;
constexpr int kQuarter = 6 * kRow;
constexpr int kSlot = 4 * kQuarter;
constexpr int kRing = 3;
constexpr int kImg = 64 * kRow;
constexpr int kFlagOff = kRing * kSlot + kImg;   // one word: "another tile follows", C wave 0 -> MFMA waves
constexpr int kG3Lds = kFlagOff + 16;           // 156 688 B

// A value that is the same in every lane, handed to the compiler as such (what is loaded from a record in global memory
// counts as divergent for hipcc, and an "s" operand of an asm statement must be provably uniform).
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

struct G3Args {
  const GemmDesc* descs;
  int count, m, n, k;
  int lower;      // 1 / 2: records are lower_only (= this value) with row_off = col_off = 0 and m == n: tiles above the diagonal are skipped
  int beta_one;   // 1: C += A B; 0: C = A B
  int order;      // tile order: 0 flat (column by column, workgroup wg takes tiles wg, wg + nwg, ...), 1 strips of four tile rows, nwg / 8 consecutive tiles per XCD
  int nwg;        // workgroups of the launch (a multiple of 8; 256 = one per CU)
};

template <int LAYOUT>   // kGemmAmBn (2), kGemmAmBk (0) or kGemmAkBk (1)
__global__ __launch_bounds__(1024, 1) void k_gemm3(G3Args P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = P.k / 16;
  const int TM = (P.m + 127) >> 7, TN = (P.n + 63) >> 6;
  const int lower = P.lower;
  // Tile order.  The dispatcher deals workgroups round-robin over the 8 XCDs (wg & 7; each XCD has its own 4 MB L2).  The
  // tiles of a matrix are numbered strip by strip -- a strip = four tile rows (512 rows), walked column by column, in a
  // lower-only launch only the tiles that reach the diagonal (tm >= tn >> 1) --, the numbers run on from matrix to matrix,
  // and in round i the 32 workgroups of XCD x take the 32 consecutive tiles 256 i + 32 x .. + 31: a block of 4 x 8 tiles
  // whose 4 row panels of A and 8 column panels of B cross the fabric once and serve 8 resp. 4 tiles from that L2.
  // Every workgroup gets the same number of tiles (+- 1) and none is ever skipped.  (Measured, 32 x (6000 x 6000, K = 256):
  // numbered column by column over ALL rows, workgroup wg taking wg, wg + 256, ..., an XCD's 32 tiles lie in 32 different
  // row panels -- 8 MB per round -- and the kernel ran at 0.71 of the MFMA peak instead of 0.77; fixed super-tiles of 4 x 8
  // with the tiles that do not exist skipped were slower still: the workgroups drift apart.)  order 0 keeps the flat
  // numbering for comparison.
  const int wg = (int)blockIdx.x;
  const int order = P.order;
  const int nwg = P.nwg;
  const int w_first = order == 0 ? wg : (nwg >> 3) * (wg & 7) + (wg >> 3);   // this workgroup's tile number in round 0
  struct It { int z, s, r, tm, tn; bool ok; };   // matrix, strip, number inside the strip -> (tm, tn)
  const int NS = order == 0 ? 1 : (TM + 3) >> 2;
  auto s_rows = [&](int st) { return order == 0 ? TM : min(4, TM - 4 * st); };
  auto s_row0 = [&](int st) { return order == 0 ? 0 : 4 * st; };
  // lower: columns 0 .. full - 1 of a strip have all its rows, column tn beyond them the rows tn >> 1 .. last
  auto s_full = [&](int st) { return lower ? min(TN, 2 * s_row0(st) + 2) : TN; };
  auto s_len = [&](int st) {
    const int h = s_rows(st), r0 = s_row0(st), full = s_full(st);
    int len = h * full;
    if (lower)
      for (int tn = full; tn < TN && (tn >> 1) < r0 + h; ++tn) len += r0 + h - (tn >> 1);
    return len;
  };
  auto it_decode = [&](It& it) {   // (s, r) -> (tm, tn): scalar arithmetic, a division by the strip height 1 .. 4
    const int h = s_rows(it.s), r0 = s_row0(it.s), full = s_full(it.s);
    int r = it.r;
    if (r < h * full) {
      const int c = order == 0 ? r / h : (h == 4 ? r >> 2 : (h == 2 ? r >> 1 : (h == 1 ? r : (r * 43691) >> 17)));
      it.tn = c;
      it.tm = r0 + r - c * h;
      return;
    }
    r -= h * full;
    int tn = full;
    for (;;) {
      const int cnt = r0 + h - (tn >> 1);
      if (r < cnt) break;
      r -= cnt;
      ++tn;
    }
    it.tn = tn;
    it.tm = (tn >> 1) + r;
  };
  auto it_norm = [&](It& it) {   // carry r over the strips and matrices
    it.ok = false;
    for (;;) {
      if (it.z >= P.count) return;
      const int len = s_len(it.s);
      if (it.r < len) break;
      it.r -= len;
      if (++it.s == NS) { it.s = 0; ++it.z; }
    }
    it.ok = true;
    it_decode(it);
  };
  auto it_first = [&]() {
    It it;
    it.z = it.s = it.tm = it.tn = 0;
    it.r = w_first;
    it_norm(it);
    return it;
  };
  auto it_next = [&](It& it) {
    if (!it.ok) return;
    it.r += nwg;
    it_norm(it);
  };
  // (no count of the workgroup's tiles up front -- a walk over all of them is some hundred scalar iterations per wave
  // and launch, 0.6 ms measured --: every role walks the iterator itself and stops when it runs out; all roles see the
  // same sequence, so they execute the same number of step barriers)
  if (!it_first().ok) return;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  if (w >= 4 && w < 12) {
    // ---------------------------------------------------------------------------------- operand loader waves
    const int d = w - 4, q = d & 3, parity = d >> 2;
    // groups are issued in order g = parity, parity + 2, ...: (tile, K step inside it) advance by two K steps at a time;
    // `cur` follows the K step the workgroup is in (one barrier per step until the tiles run out)
    It lt = it_first(), cur = lt;
    int l_ks = parity, l_slot = parity % kRing;
    while (lt.ok && l_ks >= KS) { l_ks -= KS; it_next(lt); }
    unsigned long long ab_t = 0, bb_t = 0, lda8 = 0, ldb8 = 0;
    unsigned voff_a[4] = {0, 0, 0, 0}, voff_b0 = 0, voff_b1 = 0;
    bool l_new = true;
    auto issue_group = [&]() {
      if (l_new) {
        const GemmDesc& D = P.descs[lt.z];
        const int row0 = lt.tm * 128, col0 = lt.tn * 64;
        const int mrem = min(128, P.m - row0), nrem = min(64, P.n - col0);
        if (LAYOUT == kGemmAkBk) {
          lda8 = uni64((unsigned long long)D.sa_i * 8ull);
          ab_t = uni64((unsigned long long)(size_t)D.a + (unsigned long long)row0 * lda8);
          // (rows beyond the matrix re-read the last row; what lands there is never stored)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            voff_a[i] = (unsigned)((unsigned long long)min(32 * q + 8 * i + (lane & 7), mrem - 1) * lda8 + (unsigned)(lane >> 3) * 16u);
        } else {
          lda8 = uni64((unsigned long long)D.sa_k * 8ull);
          ab_t = uni64((unsigned long long)(size_t)D.a + (unsigned long long)row0 * 8ull);
          // rows / columns beyond the matrix: the lane re-reads the last valid pair (what lands there is never stored)
          voff_a[0] = voff_a[1] = voff_a[2] = voff_a[3] = (unsigned)min(lane * 16, (mrem - 2) * 8);
        }
        if (LAYOUT == kGemmAmBn) {
          ldb8 = uni64((unsigned long long)D.sb_k * 8ull);
          bb_t = uni64((unsigned long long)(size_t)D.b + (unsigned long long)col0 * 8ull);
          const int x = (lane >> 4) * 16 + (lane & 7) * 2;
          voff_b0 = voff_b1 = (unsigned)(((lane >> 3) & 1) * (long long)ldb8 + min(x, nrem - 2) * 8);
        } else {
          ldb8 = uni64((unsigned long long)D.sb_j * 8ull);
          bb_t = uni64((unsigned long long)(size_t)D.b + (unsigned long long)col0 * ldb8);
          const int x0 = 16 * q + (lane & 7);
          voff_b0 = (unsigned)((unsigned long long)min(x0, nrem - 1) * ldb8 + (unsigned)(lane >> 3) * 16u);
          voff_b1 = (unsigned)((unsigned long long)min(x0 + 8, nrem - 1) * ldb8 + (unsigned)(lane >> 3) * 16u);
        }
        l_new = false;
      }
      const int k0 = l_ks * 16;
      const int slot_l = l_slot;
      // advance to this wave's next group
      l_ks += 2;
      if (l_ks >= KS) { l_ks -= KS; it_next(lt); l_new = true; }
      l_slot += 2;
      if (l_slot >= kRing) l_slot -= kRing;
      const unsigned m0v = lds_base + (unsigned)(slot_l * kSlot + q * kQuarter + 2880);
      unsigned long long a0, a1, a2, a3, bb0, bb1;
      if (LAYOUT == kGemmAkBk) {
        a0 = a1 = a2 = a3 = ab_t + (unsigned long long)k0 * 8ull;
      } else {
        a0 = ab_t + (unsigned long long)(k0 + 4 * q) * lda8;
        a1 = a0 + lda8; a2 = a1 + lda8; a3 = a2 + lda8;
      }
      if (LAYOUT == kGemmAmBn) {
        bb0 = bb_t + (unsigned long long)(k0 + 4 * q) * ldb8;
        bb1 = bb0 + 2ull * ldb8;
      } else {
        bb0 = bb_t + (unsigned long long)k0 * 8ull;
        bb1 = bb0;
      }
      // scalar bases with the instruction offsets taken out (the offset moves the LDS and the global address alike).
      // s_nop 4: an SGPR written by a VALU instruction (v_readfirstlane / v_readlane of a spill) needs five wait states
      // before a global_* instruction reads it as its base, and hipcc pads nothing inside an asm statement.
      const unsigned long long s0 = a0 + 2880ull, s1 = a1 + 1728ull, s2 = a2 + 576ull, s3 = a3 - 576ull, s4 = bb0 - 1728ull,
                               s5 = bb1 - 2880ull;
      asm volatile(
          "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %7 offset:-2880\n\t"
          "global_load_lds_dwordx4 %2, %8 offset:-1728\n\t"
          "global_load_lds_dwordx4 %3, %9 offset:-576\n\t"
          "global_load_lds_dwordx4 %4, %10 offset:576\n\t"
          "global_load_lds_dwordx4 %5, %11 offset:1728\n\t"
          "global_load_lds_dwordx4 %6, %12 offset:2880"
          :
          : "s"(m0v), "v"(voff_a[0]), "v"(voff_a[1]), "v"(voff_a[2]), "v"(voff_a[3]), "v"(voff_b0), "v"(voff_b1), "s"(s0),
            "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5)
          : "memory");
    };
    if (lt.ok) issue_group();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int c_ks = 0;
    for (int g = 0; cur.ok; ++g) {
      if ((g & 1) == parity) {
        if (lt.ok) issue_group();   // group g + 2
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (++c_ks == KS) { c_ks = 0; it_next(cur); }
    }
    __builtin_amdgcn_s_barrier();   // (the MFMA waves' last swap)
    return;
  }

  if (w >= 12) {
    // ------------------------------------------------------------------------------------------------ C waves
    const int cw = w - 12;
    const bool with_c = P.beta_one != 0;
    // One tile as a C wave sees it (all scalar but the lane offset of the clamped DMA reads)
    struct CT {
      unsigned long long base, col;   // byte address of (row0, first column of this wave), byte stride of a column
      int row0, colg, mrem, ncol;     // first row, first global column of this wave, rows / columns of this wave that exist
      unsigned voff;                  // lane offset of the DMA reads (rows beyond the matrix re-read the last pair)
    };
    auto ct_of = [&](const It& it) {
      CT t;
      const GemmDesc& D = P.descs[it.z];
      t.row0 = it.tm * 128;
      t.colg = it.tn * 64 + 16 * cw;
      t.mrem = min(128, P.m - t.row0);
      t.ncol = max(0, min(16, P.n - t.colg));
      t.col = uni64((unsigned long long)D.ldc * 8ull);
      t.base = uni64((unsigned long long)(size_t)D.c + (unsigned long long)t.colg * t.col + (unsigned long long)t.row0 * 8ull);
      t.voff = (unsigned)min(lane * 16, (t.mrem - 2) * 8);
      return t;
    };
    // M0 of the group of eight columns j = 8 h .. 8 h + 7 (LDS-DMA lands at M0 + immediate + 16 lane)
    auto set_m0 = [&](int h) {
      const unsigned m0v = lds_base + (unsigned)(kRing * kSlot + (16 * cw + 8 * h + 3) * kRow + 576);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(m0v) : "memory");
    };
    auto dma_col = [&](const CT& t, int j) {   // column j of the tile -> image column j
      if (j >= t.ncol) return;
      const unsigned long long a = t.base + (unsigned long long)j * t.col;
      switch (j & 7) {
        case 0: { const unsigned long long b = a + 4032ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-4032" : : "v"(t.voff), "s"(b) : "memory"); break; }
        case 1: { const unsigned long long b = a + 2880ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2880" : : "v"(t.voff), "s"(b) : "memory"); break; }
        case 2: { const unsigned long long b = a + 1728ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-1728" : : "v"(t.voff), "s"(b) : "memory"); break; }
        case 3: { const unsigned long long b = a + 576ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-576" : : "v"(t.voff), "s"(b) : "memory"); break; }
        case 4: { const unsigned long long b = a - 576ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:576" : : "v"(t.voff), "s"(b) : "memory"); break; }
        case 5: { const unsigned long long b = a - 1728ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:1728" : : "v"(t.voff), "s"(b) : "memory"); break; }
        case 6: { const unsigned long long b = a - 2880ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:2880" : : "v"(t.voff), "s"(b) : "memory"); break; }
        default: { const unsigned long long b = a - 4032ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:4032" : : "v"(t.voff), "s"(b) : "memory"); break; }
      }
    };
    const char* img = lds + kRing * kSlot + (16 * cw) * kRow + lane * 16;
    const unsigned voff_st = (unsigned)lane * 16u;
    // image column j -> column j of the tile: the rows that exist, and in a lower-only launch the rows on / below the
    // diagonal (a pair of rows that straddles it stores its second row alone)
    auto out_col = [&](const CT& t, int j) {
      if (j >= t.ncol) return;
      // first row of the tile that is stored (lower = 2: from the even row on -- the first super-diagonal entry of every
      // even row is kept as well, symm3.hip; the tile's first row is even)
      const int cj = lower == 2 ? ((t.colg + j) & ~1) : t.colg + j;
      const int rel = lower ? max(0, cj - t.row0) : 0;
      if (rel >= t.mrem) return;
      const d2v v = *(const d2v*)(img + j * kRow);
      const unsigned long long a = t.base + (unsigned long long)j * t.col;
      if (rel == 0 && t.mrem == 128) {
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff_st), "v"(v), "s"(a) : "memory");
      } else {
        if (2 * lane >= rel && 2 * lane < t.mrem)
          asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff_st), "v"(v), "s"(a) : "memory");
        if ((rel & 1) && lane == (rel >> 1)) {
          const double hi = v[1];
          const unsigned long long a8 = a + 8ull;
          asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_nop 1" : : "v"(voff_st), "v"(hi), "s"(a8) : "memory");
        }
      }
    };
    It i_cur = it_first(), i_next = i_cur;
    it_next(i_next);
    CT t_prev{}, t_cur = ct_of(i_cur), t_next{};
    if (i_next.ok) t_next = ct_of(i_next);
    if (with_c) {
      set_m0(0);
      for (int j = 0; j < 8; ++j) dma_col(t_cur, j);
      set_m0(1);
      for (int j = 8; j < 16; ++j) dma_col(t_cur, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // columns per K step and the first step of the two groups of eight (the second group's M0 write waits for the first
    // group's DMA in flight: idle steps in between)
    const int cps = KS >= 13 ? 2 : (KS >= 9 ? 4 : 8);
    const int gsteps = 8 / cps;
    const int first0 = 1, first1 = 1 + gsteps + (KS >= 9 ? 2 : 1);
    for (int t = 0; i_cur.ok; ++t) {
      const bool dma_ok = with_c && i_next.ok;
      const bool st_ok = t > 0;
      for (int ks = 0; ks < KS; ++ks) {
        int j0 = -1;
        if (ks >= first0 && ks < first0 + gsteps) j0 = (ks - first0) * cps;
        if (ks >= first1 && ks < first1 + gsteps) j0 = 8 + (ks - first1) * cps;
        if (j0 >= 0) {
          if (dma_ok && (ks == first0 || ks == first1)) set_m0(j0 >> 3);
          for (int j = j0; j < j0 + cps; ++j) {
            if (st_ok) out_col(t_prev, j);
            if (dma_ok) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the column has been read: its place is free)
              dma_col(t_next, j);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (dma_ok && ks == KS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ks == KS - 1 && cw == 0) {
          // (the MFMA waves keep no tile iterator -- its loops in their K loop cost them registers --: they read this
          // word behind the barrier of a tile's last step)
          *(__attribute__((address_space(3))) int*)(lds + kFlagOff) = i_next.ok ? 1 : 0;   // (ordered by the asm statements' memory clobbers)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
      t_prev = t_cur;
      t_cur = t_next;
      i_cur = i_next;
      it_next(i_next);
      if (i_next.ok) t_next = ct_of(i_next);
    }
    __builtin_amdgcn_s_barrier();   // (the MFMA waves' last swap)
    for (int j = 0; j < 16; ++j) out_col(t_prev, j);
    return;
  }

  // ---------------------------------------------------------------------------------------------- MFMA waves
  const int wm = w & 1, wn = w >> 1;
  const int fr = lane & 15, fk = lane >> 4;
  __builtin_amdgcn_s_barrier();   // (the loaders' prologue)

  // fragment addresses inside a slot (bytes): ONE per-lane offset per operand, everything else is an immediate of the
  // LDS instruction (a tile row step is 128 B, a k4 step a quarter)
  const unsigned a_base = LAYOUT == kGemmAkBk
                              ? (unsigned)((2 * wm) * kQuarter + (fr >> 3) * kRow + (((fk >> 1) * 8 + (fr & 7)) * 16) + (fk & 1) * 8)
                              : (unsigned)(fk * kRow + (wm * 64 + fr) * 8);
  const unsigned b_base = LAYOUT == kGemmAmBn
                              ? (unsigned)(4 * kRow + (fk >> 1) * kRow + (wn * 2) * 256 + (fk & 1) * 128 + fr * 8)
                              : (unsigned)((2 * wn) * kQuarter + 4 * kRow + (fr >> 3) * kRow + (((fk >> 1) * 8 + (fr & 7)) * 16) + (fk & 1) * 8);
  constexpr int kBni = LAYOUT == kGemmAmBn ? 256 : kQuarter, kBk4 = LAYOUT == kGemmAmBn ? kQuarter : 256;
  // A: row tile mi and k-step k4 as immediates (k-contiguous A: 8-row pieces, two per row tile, four per quarter)
  constexpr int kAk4 = LAYOUT == kGemmAkBk ? 256 : kQuarter;
  auto a_mi = [](int mi) { return LAYOUT == kGemmAkBk ? (mi >> 1) * kQuarter + (mi & 1) * 2 * kRow : mi * 128; };
  char* cimg = lds + kRing * kSlot + (wn * 32 + fk) * kRow + (wm * 64 + fr) * 8;   // + (ni * 16 + 4 r) * kRow + mi * 128
  const bool with_c = P.beta_one != 0;
  // C(row = wm 64 + mi 16 + fr, col = wn 32 + ni 16 + fk + 4 r) <-> acc[ni][mi][r]
  d4 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi][r] = with_c ? *(const double*)(cimg + (ni * 16 + 4 * r) * kRow + mi * 128) : 0.0;

  double af[3][4], bf[3][2];   // [2]: the first fragments of the NEXT step, read behind the barrier
  auto read_frags = [&](int buf, const char* sp, int k4) {
    const char* pa = sp + a_base;
    const char* pb = sp + b_base;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) af[buf][mi] = *(const double*)(pa + k4 * kAk4 + a_mi(mi));
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bf[buf][ni] = *(const double*)(pb + k4 * kBk4 + ni * kBni);
  };
  // The reads behind the barrier as inline asm: hipcc sinks ordinary loads below the MFMAs that are meant to cover their
  // latency (and a scheduling fence does not hold them).  Their completion is waited for by hand (frags_arrived).
  auto read_frags_asm = [&](int buf, unsigned slot_addr) {   // k4 = 0 of the slot at LDS byte address slot_addr
    const unsigned pa = slot_addr + a_base, pb = slot_addr + b_base;
    asm volatile("ds_read_b64 %0, %1" : "=v"(af[buf][0]) : "v"(pa) : "memory");
    if (LAYOUT == kGemmAkBk) {
      asm volatile("ds_read_b64 %0, %1 offset:2304" : "=v"(af[buf][1]) : "v"(pa) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:6912" : "=v"(af[buf][2]) : "v"(pa) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:9216" : "=v"(af[buf][3]) : "v"(pa) : "memory");
    } else {
      asm volatile("ds_read_b64 %0, %1 offset:128" : "=v"(af[buf][1]) : "v"(pa) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:256" : "=v"(af[buf][2]) : "v"(pa) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:384" : "=v"(af[buf][3]) : "v"(pa) : "memory");
    }
    asm volatile("ds_read_b64 %0, %1" : "=v"(bf[buf][0]) : "v"(pb) : "memory");
    if (LAYOUT == kGemmAmBn) asm volatile("ds_read_b64 %0, %1 offset:256" : "=v"(bf[buf][1]) : "v"(pb) : "memory");
    else asm volatile("ds_read_b64 %0, %1 offset:6912" : "=v"(bf[buf][1]) : "v"(pb) : "memory");
  };
  auto frags_arrived = [&](int buf) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]), "+v"(bf[buf][0]), "+v"(bf[buf][1])
                 :
                 : "memory");
  };
  // (operands swapped, as in k_gemm2: 16 consecutive lanes of an accumulator register are 16 consecutive rows of C)
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[buf][ni], af[buf][mi], acc[ni][mi], 0, 0, 0);
  };
  // The swap at a tile boundary: result out, next C in, 8 values at a time (all 32 at once would need 64 registers).
  auto swap_all = [&]() {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        double nxt[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) nxt[rr][mi] = *(const double*)(cimg + (ni * 16 + 4 * (2 * r2 + rr)) * kRow + mi * 128);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            *(double*)(cimg + (ni * 16 + 4 * (2 * r2 + rr)) * kRow + mi * 128) = acc[ni][mi][2 * r2 + rr];
            acc[ni][mi][2 * r2 + rr] = with_c ? nxt[rr][mi] : 0.0;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  // An iteration = K step g from its k4 = 1 on, plus the k4 = 0 of step g + 1 (the tail of the LAST iteration works on a
  // slot and on accumulators nobody needs).  ONE flat loop with ONE copy of the MFMAs: with a tile loop around it hipcc
  // peels the boundary iterations, and with a second copy of the MFMAs in a boundary branch it moves the accumulator
  // tuples through scratch where the paths join.
  read_frags(2, lds, 0);
  read_frags(1, lds, 1);
  mfmas(2);
  int ks = 0, slot = 0;
  bool more = true;
#pragma clang loop unroll(disable)
  while (more) {
    const char* sp = lds + slot * kSlot;
    slot = slot == kRing - 1 ? 0 : slot + 1;
    read_frags(0, sp, 2);
    mfmas(1);
    read_frags(1, sp, 3);
    mfmas(0);
    __builtin_amdgcn_sched_barrier(0);   // (MFMAs have no side effects: without the fence hipcc moves them across the barrier)
    // (the k4 = 3 fragments as inputs: hipcc then knows that they have arrived and does not wait for "its" loads again
    // behind the barrier, where the wait would catch the asm reads as well)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 :
                 : "v"(af[1][0]), "v"(af[1][1]), "v"(af[1][2]), "v"(af[1][3]), "v"(bf[1][0]), "v"(bf[1][1])
                 : "memory");
    __builtin_amdgcn_s_barrier();
    const char* sn = lds + slot * kSlot;
    const bool boundary = ks == KS - 1;
    ks = boundary ? 0 : ks + 1;
    read_frags_asm(2, lds_base + (unsigned)(slot * kSlot));
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);                   // k4 = 3 of step g (at a boundary: the accumulators are final behind these)
    __builtin_amdgcn_sched_barrier(0);
    frags_arrived(2);
    read_frags(1, sn, 1);
    if (boundary) {
      swap_all();
      more = __builtin_amdgcn_readfirstlane(*(const __attribute__((address_space(3))) int*)(lds + kFlagOff)) != 0;
    }
    mfmas(2);                   // k4 = 0 of step g + 1
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // the last result is in the image
}

}  // namespace

namespace {
bool g_gemm3_any_size = false;   // debugging (sc_dbg_gemm3_host): take every launch that qualifies, whatever its size
int g_gemm3_order = -1;          // debugging (sc_dbg_gemm3_bench): tile order, -1 = SPRINGCRAFT_GEMM3_ORDER / the default
}

// Launch `count` records with the same (m, n, k) on k_gemm3, or say that the launch is not one it takes (returns 1: the
// caller uses launch_gemm_f64).  Taken: layout kGemmAmBn / kGemmAmBk, alpha = 1, beta in {0, 1}, k a multiple of 16 and
// >= 128, m (and n for kGemmAmBn) even, pointers and leading dimensions that keep 16-byte alignment (aligned16: the
// caller knows its records), enough tiles to give every CU a few.  lower: as launch_gemm_f64's lower_grid.
bool gemm3_would_take(sc_ctx* ctx, int count, int m, int n, int k, int layout, int lower, double alpha, double beta,
                      bool aligned16) {
  static const int env = [] { const char* e = getenv("SPRINGCRAFT_GEMM3"); return e ? atoi(e) : 1; }();
  if (env == 0 || count <= 0) return false;
  if (layout != kGemmAmBn && layout != kGemmAmBk && layout != kGemmAkBk) return false;
  if (alpha != 1.0 || (beta != 0.0 && beta != 1.0)) return false;
  if (!aligned16 || k < 128 || (k & 15) || (m & 1) || m < 2 || n < 1) return false;
  if (layout == kGemmAmBn && (n & 1)) return false;
  if (lower && m != n) return false;
  // (the lower-only trailing update of the band reduction: 0.66 against k_gemm2's 0.65 of the MFMA peak on 32 matrices
  // alone, 192 against 187 ms inside the C3 step -- its diagonal tiles store under lane predicates and its strips start
  // with few tiles --: left to k_gemm2 unless SPRINGCRAFT_GEMM3_LOWER = 1)
  // For a few large matrices it wins (one n = 24000 matrix, config C5: 186 -> 178 ms of trailing updates per solve): there
  // the two half batches on two streams, whose panel QRs hide beside k_gemm2's workgroups, do not exist.
  // Round 6: with the two half batches on two streams (ctx->gemm3_side_by_side, set by the band reduction) it takes the
  // update after all -- on 224 workgroups instead of 256, which leaves an eighth of the CUs to the other half's panel QR
  // and small products: C3 step 2200 -> 2150-2160 ms (same box, tools/quick_env_ab.sh: 208 / 216 / 224 workgroups 2157-2177 /
  // 2149-2169 / 2145-2166, 232 / 240: 2190-2210, 256: 2191-2196; profiles/r06_syr2k_wgs.txt).
  static const int env_lower = [] { const char* e = getenv("SPRINGCRAFT_GEMM3_LOWER"); return e ? atoi(e) : -1; }();
  if (lower && !g_gemm3_any_size && (env_lower == 0 || (env_lower < 0 && count >= 8 && !ctx->gemm3_side_by_side))) return false;
  // (the kernel's tile order is built on 8 XCDs x 32 workgroups: a device -- or a partition -- with fewer CUs stays on k_gemm2)
  if (ctx->num_cus < 256 || ctx->gemm3_attr == 0) return false;
  const long long TM = (m + 127) / 128, TN = (n + 63) / 64;
  const long long hN = TN / 2;
  const long long T1 = lower ? TM * TN - hN * (hN - 1) - ((TN & 1) ? hN : 0) : TM * TN;
  const long long total = T1 * count;
  if (env != 2 && !g_gemm3_any_size && total < 4LL * 256) return false;   // (SPRINGCRAFT_GEMM3 = 2: every launch that qualifies, for the tests)
  return total <= 0x3fffffffLL;
}

int launch_gemm3_uniform(sc_ctx* ctx, const GemmDesc* d_desc, int count, int m, int n, int k, int layout, int lower,
                         double alpha, double beta, bool aligned16) {
  if (!gemm3_would_take(ctx, count, m, n, k, layout, lower, alpha, beta, aligned16)) return 1;
  if (ctx->gemm3_attr < 0) {   // per device, hence per context
    const bool ok0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm3<kGemmAmBk>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kG3Lds) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm3<kGemmAkBk>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kG3Lds) == hipSuccess;
    const bool ok2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm3<kGemmAmBn>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kG3Lds) == hipSuccess;
    ctx->gemm3_attr = (ok0 && ok2) ? 1 : 0;
  }
  if (ctx->gemm3_attr != 1) return 1;
  static const int env_order = [] { const char* e = getenv("SPRINGCRAFT_GEMM3_ORDER"); return e ? atoi(e) : 1; }();
  // Workgroups of a lower-only launch (the band reduction's trailing update), SPRINGCRAFT_GEMM3_LOWER_WGS (a multiple of
  // 8, default 256): fewer than one per CU leaves CUs to the other half batch's panel QR that runs beside it
  static const int env_lower_wgs = [] { const char* e = getenv("SPRINGCRAFT_GEMM3_LOWER_WGS"); return e ? std::max(8, std::min(256, atoi(e) / 8 * 8)) : 0; }();
  const int nwg = lower ? (env_lower_wgs > 0 ? env_lower_wgs : (ctx->gemm3_side_by_side ? 224 : 256)) : 256;
  G3Args A{d_desc, count, m, n, k, lower, beta != 0.0 ? 1 : 0, g_gemm3_order >= 0 ? g_gemm3_order : env_order, nwg};
  const unsigned grid = (unsigned)nwg;
  if (layout == kGemmAmBn) hipLaunchKernelGGL(k_gemm3<kGemmAmBn>, dim3(grid), dim3(1024), kG3Lds, ctx->stream, A);
  else if (layout == kGemmAkBk) hipLaunchKernelGGL(k_gemm3<kGemmAkBk>, dim3(grid), dim3(1024), kG3Lds, ctx->stream, A);
  else hipLaunchKernelGGL(k_gemm3<kGemmAmBk>, dim3(grid), dim3(1024), kG3Lds, ctx->stream, A);
  if (hipGetLastError() != hipSuccess) {
    ctx->gemm3_attr = 0;   // refused once: this context stays on k_gemm2
    return 1;
  }
  ++ctx->cnt_gemm3_launches;
  return SC_OK;
}

// ---- debug entry for the GPU unit tests (tests/test_gemm_gpu.py; not part of the public C ABI): `count` products of one
// shape on host data through k_gemm3, whatever their size.  a: count x (m x k) column-major; b: count x (k x n)
// column-major (layout kGemmAmBk) or count x (n x k) column-major (kGemmAmBn); c: count x (m x n) column-major, in / out.
// Returns SC_OK, or SC_ERR_INVALID_ARG when the kernel does not take the shape.
extern "C" int sc_dbg_gemm3_host(sc_ctx* ctx, const double* a, const double* b, double* c, int count, int m, int n, int k,
                                 int layout, int lower, double beta) {
  if (!ctx || !a || !b || !c || count < 1 || m < 1 || n < 1 || k < 1) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  const size_t ea = (size_t)m * k, eb = (size_t)k * n, ec = (size_t)m * n;
  char* base = nullptr;
  SC_HIP(ctx, hipMalloc((void**)&base, (ea + eb + ec) * count * 8 + sizeof(GemmDesc) * (size_t)count + 256));
  double* da = (double*)base;
  double* db = da + ea * count;
  double* dc = db + eb * count;
  GemmDesc* dd = (GemmDesc*)(dc + ec * count);
  int rc = SC_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == SC_OK) rc = sc_set_error(ctx, SC_ERR_HIP, "%s", hipGetErrorString(e)); };
  fail(hipMemcpy(da, a, ea * count * 8, hipMemcpyHostToDevice));
  fail(hipMemcpy(db, b, eb * count * 8, hipMemcpyHostToDevice));
  fail(hipMemcpy(dc, c, ec * count * 8, hipMemcpyHostToDevice));
  std::vector<GemmDesc> h((size_t)count);
  for (int z = 0; z < count; ++z) {
    GemmDesc D{};
    D.a = da + ea * z; D.b = db + eb * z; D.c = dc + ec * z;
    D.m = m; D.n = n; D.k = k; D.ldc = m; D.alpha = 1.0; D.beta = beta;
    D.sa_i = 1; D.sa_k = m;
    if (layout == kGemmAkBk) { D.sa_i = k; D.sa_k = 1; }   // a: count x (k x m) column-major
    if (layout == kGemmAmBn) { D.sb_k = n; D.sb_j = 1; } else { D.sb_k = 1; D.sb_j = k; }
    D.lower_only = lower;
    h[(size_t)z] = D;
  }
  fail(hipMemcpy(dd, h.data(), sizeof(GemmDesc) * (size_t)count, hipMemcpyHostToDevice));
  if (rc == SC_OK) {
    g_gemm3_any_size = true;
    const int took = launch_gemm3_uniform(ctx, dd, count, m, n, k, layout, lower, 1.0, beta, true);
    g_gemm3_any_size = false;
    if (took != SC_OK) rc = sc_set_error(ctx, SC_ERR_INVALID_ARG, "k_gemm3 does not take m %d n %d k %d layout %d", m, n, k, layout);
    fail(hipStreamSynchronize(ctx->stream));
  }
  if (rc == SC_OK) fail(hipMemcpy(c, dc, ec * count * 8, hipMemcpyDeviceToHost));
  (void)hipFree(base);
  return rc;
}

// ---- debug / tuning entry (not part of the public C ABI): `count` matrices of one shape on freshly allocated buffers, timed
// through k_gemm3 (kernel 3; order: its tile order) or k_gemm2 (kernel 2) with the SAME records.  C += A B, or with
// lower != 0 the lower triangle of it (m == n).  tools/gemm3_shapes.py
extern "C" int sc_dbg_gemm3_bench(sc_ctx* ctx, int count, int m, int n, int k, int layout, int lower, int kernel, int order,
                                  int iters, double* ms_out) {
  if (!ctx || count < 1 || iters < 1) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  // (experiments: leading dimensions of A / C and of an n-contiguous B other than m / n)
  static const int ld_a = [] { const char* e = getenv("SC_DBG_LDA"); return e ? atoi(e) : 0; }();
  static const int ld_c = [] { const char* e = getenv("SC_DBG_LDC"); return e ? atoi(e) : 0; }();
  static const int ld_b = [] { const char* e = getenv("SC_DBG_LDB"); return e ? atoi(e) : 0; }();
  const int lda = std::max(m, ld_a), ldc = std::max(m, ld_c), ldbn = std::max(n, ld_b);
  const size_t ea = (size_t)lda * k, eb = layout == kGemmAmBn ? (size_t)ldbn * k : (size_t)k * n, ec = (size_t)ldc * n;
  char* base = nullptr;
  SC_HIP(ctx, hipMalloc((void**)&base, (ea + eb + ec) * count * 8 + sizeof(GemmDesc) * (size_t)count + 256));
  double* da = (double*)base;
  double* db = da + ea * count;
  double* dc = db + eb * count;
  GemmDesc* dd = (GemmDesc*)(dc + ec * count);
  std::vector<double> hv(std::max(ea, eb));
  unsigned long long sd = 88172645463325252ull;
  for (auto& x : hv) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; x = (double)((long long)(sd % 2001) - 1000) / 1000.0; }
  int rc = SC_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == SC_OK) rc = sc_set_error(ctx, SC_ERR_HIP, "%s", hipGetErrorString(e)); };
  for (int z = 0; z < count; ++z) {
    fail(hipMemcpy(da + ea * z, hv.data(), ea * 8, hipMemcpyHostToDevice));
    fail(hipMemcpy(db + eb * z, hv.data(), eb * 8, hipMemcpyHostToDevice));
  }
  fail(hipMemset(dc, 0, ec * count * 8));
  std::vector<GemmDesc> h((size_t)count);
  for (int z = 0; z < count; ++z) {
    GemmDesc D{};
    D.a = da + ea * z; D.b = db + eb * z; D.c = dc + ec * z;
    D.m = m; D.n = n; D.k = k; D.ldc = ldc; D.alpha = 1.0; D.beta = 1.0;
    D.sa_i = 1; D.sa_k = lda;
    if (layout == kGemmAkBk) { D.sa_i = k; D.sa_k = 1; }
    if (layout == kGemmAmBn) { D.sb_k = ldbn; D.sb_j = 1; } else { D.sb_k = 1; D.sb_j = k; }
    D.lower_only = lower;
    h[(size_t)z] = D;
  }
  fail(hipMemcpy(dd, h.data(), sizeof(GemmDesc) * (size_t)count, hipMemcpyHostToDevice));
  auto launch = [&]() -> int {
    if (kernel == 3) {
      g_gemm3_any_size = true;
      g_gemm3_order = order;
      const int took = launch_gemm3_uniform(ctx, dd, count, m, n, k, layout, lower, 1.0, 1.0, true);
      g_gemm3_any_size = false;
      g_gemm3_order = -1;
      return took == SC_OK ? SC_OK : SC_ERR_INVALID_ARG;
    }
    return launch_gemm_f64(ctx, dd, count, m, n, kGemmTile, 1, false, false, layout, lower != 0 && layout == kGemmAmBn);
  };
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == SC_OK) {
    fail(hipEventCreate(&e0));
    fail(hipEventCreate(&e1));
    for (int it = 0; it < 2 && rc == SC_OK; ++it) rc = launch();
    fail(hipEventRecord(e0, ctx->stream));
    for (int it = 0; it < iters && rc == SC_OK; ++it) rc = launch();
    fail(hipEventRecord(e1, ctx->stream));
    fail(hipEventSynchronize(e1));
    float ms = 0.f;
    if (rc == SC_OK) fail(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(base);
  return rc;
}
