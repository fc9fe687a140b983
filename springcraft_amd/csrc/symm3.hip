// k_symm3: X = A V for a symmetric A of which only the lower triangle is stored -- the SYMM of the band reduction
// (twostage.hip, X = A22 V: 2/3 n^3 flops of the eigensolver) as ONE role-split persistent launch (round 6).
//
// Until round 5 the product ran as two triangular-operand launches of k_gemm2 (X1 = L V with the K range 0 .. row,
// X2 = strict(L)^T V with the K range row .. m) whose tiles have K ranges between 0 and m: the lowest MFMA group of the
// headline step (0.57 of the f64 matrix peak) and of C5 (0.41).  Here every 128 x 64 tile of X has the SAME K loop, all m
// columns of A's row block in one pass:
//
//     X[I] = sum_{k < I0} L(I, k) V(k)                 row part:     A(i, k) at a + i + k lda        (m-contiguous, "Am")
//          + sym(L(I, I)) V(I)                         diagonal block, see below
//          + sum_{k >= I0 + 128} L(k, I)^T V(k)        column part:  A(k, i) at a + k + i lda        (k-contiguous, "Ak")
//
// with k_gemm3's roles (gemm3.hip: 4 MFMA waves that touch LDS only, 8 operand loaders issuing nothing but LDS-DMA, 4 C
// waves that carry the finished tile out of an LDS image).  What is new:
//   * ONE LDS image for both operand layouts: k_gemm3's image of a k-contiguous A -- 16 pieces of 1 KB = one LDS-DMA
//     instruction each, piece p = the 8 rows 8 p .. 8 p + 7 x the 16 k of the step, the four k4 quarters 256 bytes apart
//     inside it.  From k-contiguous memory a lane fetches a k pair of one row (8 full 128-byte lines per instruction) and
//     lands at [kp][row]; from m-contiguous memory it fetches a row pair at one k (16 half lines per instruction, the other
//     halves by the wave's next instruction) and lands at [k][rp].  Either way the 32 fragment reads an MFMA wave makes
//     in one piece are 256 contiguous bytes (no bank conflict), and the two layouts differ in ONE per-lane base only:
//     the MFMA stream is k_gemm3's, with the base switched at the step where the tile's K loop passes the diagonal.
//     (A first version had 1-KB blocks of 32 rows x 4 k: its k-contiguous pass fetched 32-byte pieces of 32 lines per
//     instruction and the product ran at 0.62 of the MFMA peak.)
//   * The diagonal block runs TWICE, once per layout, each pass with the 16-byte pieces the other one covers (or nobody:
//     above the diagonal in the m-contiguous pass, on / below it in the k-contiguous one) fetched from a page of zeros
//     instead.  Pieces are pairs, so the split is on pair boundaries: the m-contiguous pass takes k <= (i | 1), the
//     k-contiguous one k > (i | 1).  The piece of rows (2c, 2c + 1) at column 2c + 1 holds A(2c, 2c + 1) -- an entry ABOVE
//     the diagonal -- next to the diagonal entry: the band reduction therefore keeps the first super-diagonal entry of
//     every even row equal to its mirror image (k_mirror_lower leaves the whole upper triangle symmetric; the trailing
//     updates store (row | 1) >= col instead of row >= col: gemm_f64.hip, gemm3.hip; the scaling pass likewise).  Only
//     these 16 + 16 K steps per tile (and a tile's steps beyond the matrix) compute per-lane addresses; all others issue
//     from a scalar base.
//   * Every tile has m / 16 + 8 K steps: no triangular imbalance, no "longest first" order, no second launch, no sum of
//     X1 and X2 (the callers keep their [X1 | X2 | V] operand: X lands in X1, X2 is zeroed once per solve).
//   * Tried and dropped (profiles/r06_symm3.txt): a ring of five slots with three K steps in flight, twelve loader waves
//     and the MFMA waves storing their result themselves (no C image).  The loaders then never wait for a group (0.8 %
//     of their loop instead of 9.8 %) and the product is no faster (same box: 2137-2140 against 2124-2136 ms per C3 step):
//     neither the memory latency nor the fetched bytes bound it -- with every column of A redirected to one hot column
//     the launch set takes 180 instead of 190 ms -- the MFMA waves issue at 0.67-0.70 of their pipes' time either way.
// K slices for launches with few tiles (split > 1): slice s runs the steps [s KS / split, (s + 1) KS / split) of every tile
// into its own copy of X (GemmDesc::split_stride); k_sum_xslices adds them up as before.
//
// Requires: m even and 16-byte aligned operands (the callers: n even, even offsets); a K range that is not a multiple of
// 16 ends in one more per-lane-address step.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "gemm_f64.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int kRow = 1152;           // 1 KB piece + 128 B: consecutive pieces sit in opposite bank halves
constexpr int kQuarter = 6 * kRow;   // 4 A pieces (32 rows) | 2 B pieces
constexpr int kSlot = 4 * kQuarter;
constexpr int kRing = 3;
constexpr int kImg = 64 * kRow;
constexpr int kFlagOff = kRing * kSlot + kImg;   // one word: 0 = no further tile, else 1 + the next tile's switch step
constexpr int kS3Lds = kFlagOff + 16;

__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// Diagnostic build (-DSYMM3_STAMPS, tools/r06_symm3_stamps.sh): cycles the first loader wave of every workgroup spends
// waiting for its group to land (s_waitcnt vmcnt(0)), at the step barrier, and in the loop; steps it waited in.
#ifdef SYMM3_STAMPS
__device__ unsigned long long g_symm3_stamps[4];
#endif

struct S3Args {
  const GemmDesc* descs;   // a = A(0, 0) of the symmetric matrix (lower stored, column stride sa_k), b = V (sb_k = 1, column
                           // stride sb_j), c = X (ldc), split_stride
  int count, m;            // records, order of A = rows of V and X (even)
  int split;               // K slices
  int nwg;                 // workgroups of the launch (a multiple of 8)
  const double* zeros;     // middle of a page of zeros (+- 4 KB readable)
  int gb[17];              // slice s runs the K steps gb[s] .. gb[s + 1] - 1 of a tile (gb[split] = ceil(m / 16) + 8)
  int hot;                 // diagnostic (SPRINGCRAFT_SYMM3_DBG_HOT, results wrong): bit 0: every column of A reads column 0 (L2
                           // hits); bit 1: no step takes the per-lane-address path
};

__global__ __launch_bounds__(1024, 1) void k_symm3(S3Args P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = P.m;
  const int TM = (m + 127) >> 7;
  const int split = P.split;
  const int nwg = P.nwg;
  const int wg = (int)blockIdx.x;
  // work items: (matrix z, tile row tm, slice s), numbered z-major, then tm, then s; workgroup wg of XCD x = wg & 7 takes
  // the items nwg i + (nwg / 8) x + (wg >> 3): consecutive items -- the slices of a tile, the tiles of a matrix -- share
  // V and neighbouring rows of A in that XCD's L2
  const int per_z = TM * split;
  const int total = per_z * P.count;
  struct It { int r, z, tm, s; bool ok; };
  // (divisions once per wave, here; an item advances by nwg = adv_tm tile rows + adv_s slices without one: helper waves
  // have no vector ALU to spare inside the K loop)
  const int adv_tm = __builtin_amdgcn_readfirstlane(nwg / split), adv_s = nwg - adv_tm * split;
  auto it_norm = [&](It& it) {
    it.ok = it.r < total;
    if (!it.ok) return;
    if (it.s >= split) { it.s -= split; ++it.tm; }
    while (it.tm >= TM) { it.tm -= TM; ++it.z; }
  };
  auto it_first = [&]() {
    It it;
    it.r = (nwg >> 3) * (wg & 7) + (wg >> 3);
    it.z = 0;
    it.tm = __builtin_amdgcn_readfirstlane(it.r / split);
    it.s = it.r - it.tm * split;
    it_norm(it);
    return it;
  };
  auto it_next = [&](It& it) {
    if (!it.ok) return;
    it.r += nwg;
    it.tm += adv_tm;
    it.s += adv_s;
    it_norm(it);
  };
  // K steps [g0, g1) of an item, and the step at which its K loop changes from the m-contiguous to the k-contiguous layout
  auto g_lo = [&](const It& it) { return P.gb[it.s]; };
  auto g_hi = [&](const It& it) { return P.gb[it.s + 1]; };
  auto g_sw = [&](const It& it) { return 8 * it.tm + 8; };
  if (!it_first().ok) return;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  if (w >= 4 && w < 12) {
    // ---------------------------------------------------------------------------------- operand loader waves
    const int d = w - 4, q = d & 3, parity = d >> 2;
    // this wave fetches quarter q of the steps of its parity: `lt` / l_g = the item and step of its next group, `cur` /
    // c_g = the step the workgroup is in (one barrier per step until the items run out)
    It lt = it_first(), cur = lt;
    int l_g = g_lo(lt) + parity, l_slot = parity % kRing;
    // (the group counter runs on from item to item: what is left over at an item's end is carried into the next one)
    auto l_carry = [&]() {
      while (lt.ok && l_g >= g_hi(lt)) {
        const int over = l_g - g_hi(lt);
        it_next(lt);
        if (lt.ok) l_g = g_lo(lt) + over;
      }
    };
    l_carry();
    unsigned long long a8 = 0, b8 = 0, lda8 = 0, ldb8 = 0;
    int I0 = 0, mrem = 0, sw = 0;
    unsigned voff_m[4] = {0, 0, 0, 0}, voff_k[4] = {0, 0, 0, 0}, voff_b0 = 0, voff_b1 = 0;
    const unsigned long long zp = uni64((unsigned long long)(size_t)P.zeros);
    bool l_new = true;
    // lane roles inside a piece (8 rows x 16 k): m-contiguous (k = lane >> 2, row pair lane & 3), k-contiguous (k pair
    // lane >> 3, row lane & 7); this wave's pieces x = 0 .. 3 are the rows 32 q + 8 x .. + 7 of the tile
    const int km_l = lane >> 2, rp_l = lane & 3, kp_l = lane >> 3, x8_l = lane & 7;
    auto issue_group = [&]() {
      if (l_new) {
        const GemmDesc& D = P.descs[lt.z];
        I0 = lt.tm * 128;
        mrem = min(128, m - I0);
        sw = g_sw(lt);
        lda8 = (P.hot & 1) ? 0ull : uni64((unsigned long long)D.sa_k * 8ull);
        ldb8 = uni64((unsigned long long)D.sb_j * 8ull);
        a8 = uni64((unsigned long long)(size_t)D.a);
        b8 = uni64((unsigned long long)(size_t)D.b);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          // (rows beyond the matrix re-read the last pair / row; what lands there is never stored)
          voff_m[mi] = (unsigned)(min(32 * q + 8 * mi + 2 * rp_l, mrem - 2) * 8) + (unsigned)((unsigned long long)km_l * lda8);
          voff_k[mi] = (unsigned)((unsigned long long)min(32 * q + 8 * mi + x8_l, mrem - 1) * lda8) + (unsigned)(kp_l * 16);
        }
        const int x0 = 16 * q + (lane & 7);
        voff_b0 = (unsigned)((unsigned long long)x0 * ldb8 + (unsigned)(lane >> 3) * 16u);
        voff_b1 = (unsigned)((unsigned long long)(x0 + 8) * ldb8 + (unsigned)(lane >> 3) * 16u);
        l_new = false;
      }
      const int g = l_g;
      const int slot_l = l_slot;
      const bool am = g < sw;
      const int k0 = am ? 16 * g : 16 * (g - 8);
      const bool special = !(P.hot & 2) && ((g >= sw - 8 && g < sw + 8) || k0 + 16 > m);
      const int I0_now = I0;
      // advance to this wave's next group
      l_g += 2;
      if (l_g >= g_hi(lt)) { l_carry(); l_new = true; }
      l_slot += 2;
      if (l_slot >= kRing) l_slot -= kRing;
      const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_base + (unsigned)(slot_l * kSlot + q * kQuarter + 2880)));
      const unsigned long long bb = b8 + (unsigned long long)k0 * 8ull;
      if (!special) {
        // scalar bases with the instruction offsets taken out (the offset moves the LDS and the global address alike)
        unsigned long long s0, s1, s2, s3;
        unsigned v0, v1, v2, v3;
        if (am) {
          const unsigned long long a0 = a8 + (unsigned long long)I0_now * 8ull + (unsigned long long)k0 * lda8;
          s0 = a0 + 2880ull; s1 = a0 + 1728ull; s2 = a0 + 576ull; s3 = a0 - 576ull;
          v0 = voff_m[0]; v1 = voff_m[1]; v2 = voff_m[2]; v3 = voff_m[3];
        } else {
          const unsigned long long a0 = a8 + (unsigned long long)I0_now * lda8 + (unsigned long long)k0 * 8ull;
          s0 = a0 + 2880ull; s1 = a0 + 1728ull; s2 = a0 + 576ull; s3 = a0 - 576ull;
          v0 = voff_k[0]; v1 = voff_k[1]; v2 = voff_k[2]; v3 = voff_k[3];
        }
        const unsigned long long s4 = bb - 1728ull, s5 = bb - 2880ull;
        asm volatile(
            "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
            "global_load_lds_dwordx4 %1, %7 offset:-2880\n\t"
            "global_load_lds_dwordx4 %2, %8 offset:-1728\n\t"
            "global_load_lds_dwordx4 %3, %9 offset:-576\n\t"
            "global_load_lds_dwordx4 %4, %10 offset:576\n\t"
            "global_load_lds_dwordx4 %5, %11 offset:1728\n\t"
            "global_load_lds_dwordx4 %6, %12 offset:2880"
            :
            : "s"(m0v), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(voff_b0), "v"(voff_b1), "s"(s0), "s"(s1), "s"(s2), "s"(s3),
              "s"(s4), "s"(s5)
            : "memory");
      } else {
        // a step that meets the diagonal block or the end of the matrix: one 64-bit address per lane, pieces that must
        // not count come from the page of zeros
        unsigned long long ad[6];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          bool valid;
          unsigned long long ga;
          if (am) {
            const int i = 32 * q + 8 * mi + 2 * rp_l, gi = I0_now + i, k = k0 + km_l;
            valid = k <= (gi | 1) && k < m && gi + 1 < m;
            ga = a8 + (unsigned long long)gi * 8ull + (unsigned long long)k * lda8;
          } else {
            const int i = 32 * q + 8 * mi + x8_l, gi = I0_now + i, kk = k0 + 2 * kp_l;
            valid = kk > (gi | 1) && kk < m && gi < m;
            ga = a8 + (unsigned long long)kk * 8ull + (unsigned long long)gi * lda8;
          }
          ad[mi] = valid ? ga : zp;
        }
        {
          const int kk = k0 + 2 * (lane >> 3);
          const bool valid = kk < m;
          ad[4] = valid ? bb + (unsigned long long)voff_b0 : zp;
          ad[5] = valid ? bb + (unsigned long long)voff_b1 : zp;
        }
        ad[0] += 2880ull; ad[1] += 1728ull; ad[2] += 576ull; ad[3] -= 576ull; ad[4] -= 1728ull; ad[5] -= 2880ull;
        asm volatile(
            "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
            "global_load_lds_dwordx4 %1, off offset:-2880\n\t"
            "global_load_lds_dwordx4 %2, off offset:-1728\n\t"
            "global_load_lds_dwordx4 %3, off offset:-576\n\t"
            "global_load_lds_dwordx4 %4, off offset:576\n\t"
            "global_load_lds_dwordx4 %5, off offset:1728\n\t"
            "global_load_lds_dwordx4 %6, off offset:2880"
            :
            : "s"(m0v), "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5])
            : "memory");
      }
    };
    if (lt.ok) issue_group();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int c_g = g_lo(cur), c_hi = g_hi(cur);
#ifdef SYMM3_STAMPS
    unsigned long long st_wait = 0, st_bar = 0, st_n = 0, st_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory");
#endif
    for (int g = 0; cur.ok; ++g) {
      if ((g & 1) == parity) {
        if (lt.ok) issue_group();   // group g + 2
      } else {
#ifdef SYMM3_STAMPS
        unsigned long long ta, tb;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta)::"memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb)::"memory");
        st_wait += tb - ta;
        ++st_n;
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      }
#ifdef SYMM3_STAMPS
      unsigned long long tc, td;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc)::"memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td)::"memory");
      st_bar += td - tc;
#else
      __builtin_amdgcn_s_barrier();
#endif
      if (++c_g == c_hi) { it_next(cur); if (cur.ok) { c_g = g_lo(cur); c_hi = g_hi(cur); } }
    }
#ifdef SYMM3_STAMPS
    if (d == 0 && lane == 0) {
      unsigned long long te;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te)::"memory");
      atomicAdd(&g_symm3_stamps[0], st_wait);
      atomicAdd(&g_symm3_stamps[1], st_bar);
      atomicAdd(&g_symm3_stamps[2], te - st_t0);
      atomicAdd(&g_symm3_stamps[3], st_n);
    }
#endif
    __builtin_amdgcn_s_barrier();   // (the MFMA waves' last swap)
    return;
  }

  if (w >= 12) {
    // ------------------------------------------------------------------------------------------------ C waves
    const int cw = w - 12;
    struct CT {
      unsigned long long base, col;   // byte address of (row0, first column of this wave), byte stride of a column
      int mrem;                       // rows of the tile that exist
    };
    auto ct_of = [&](const It& it) {
      CT t;
      const GemmDesc& D = P.descs[it.z];
      const int row0 = it.tm * 128;
      t.mrem = min(128, m - row0);
      t.col = uni64((unsigned long long)D.ldc * 8ull);
      t.base = uni64((unsigned long long)(size_t)D.c + (unsigned long long)it.s * (unsigned long long)D.split_stride * 8ull +
                     (unsigned long long)(16 * cw) * t.col + (unsigned long long)row0 * 8ull);
      return t;
    };
    const char* img = lds + kRing * kSlot + (16 * cw) * kRow + lane * 16;
    const unsigned voff_st = (unsigned)lane * 16u;
    auto out_col = [&](const CT& t, int j) {   // image column j -> column j of the tile (rows in pairs: m is even)
      const d2v v = *(const d2v*)(img + j * kRow);
      const unsigned long long a = t.base + (unsigned long long)j * t.col;
      if (t.mrem == 128) {
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff_st), "v"(v), "s"(a) : "memory");
      } else if (2 * lane < t.mrem) {
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff_st), "v"(v), "s"(a) : "memory");
      }
    };
    It i_cur = it_first(), i_next = i_cur;
    it_next(i_next);
    CT t_prev{}, t_cur = ct_of(i_cur);
    __builtin_amdgcn_s_barrier();   // (the loaders' prologue)
    for (int t = 0; i_cur.ok; ++t) {
      const bool st_ok = t > 0;
      const int KS = g_hi(i_cur) - g_lo(i_cur);
      // the previous item's result leaves the image during this item's steps 1, 2, ... (the image is complete behind the
      // barrier of step 0 and is overwritten behind the barrier of the last step), two columns per step, what is left
      // in the last one: every item has at least two steps (symm3_would_take)
      int done = 0;
      for (int ks = 0; ks < KS; ++ks) {
        if (st_ok && ks >= 1 && done < 16) {
          const int upto = ks == KS - 1 ? 16 : min(16, done + 2);
          for (; done < upto; ++done) out_col(t_prev, done);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (ks == KS - 1 && cw == 0) {
          // (the MFMA waves keep no item iterator: they read this word behind the barrier of an item's last step.  0: no
          // further item; else the next item's step count << 16 | 0x4000 + the step at which its layout changes, counted
          // from its first step -- a slice may start behind the change or end in front of it)
          int word = 0;
          if (i_next.ok) word = ((g_hi(i_next) - g_lo(i_next)) << 16) | ((0x4000 + g_sw(i_next) - g_lo(i_next)) & 0xffff);
          *(__attribute__((address_space(3))) int*)(lds + kFlagOff) = word;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
      t_prev = t_cur;
      i_cur = i_next;
      it_next(i_next);
      if (i_cur.ok) t_cur = ct_of(i_cur);
    }
    __builtin_amdgcn_s_barrier();   // (the MFMA waves' last swap)
    for (int j = 0; j < 16; ++j) out_col(t_prev, j);
    return;
  }

  // ---------------------------------------------------------------------------------------------- MFMA waves
  const int wm = w & 1, wn = w >> 1;
  const int fr = lane & 15, fk = lane >> 4;
  __builtin_amdgcn_s_barrier();   // (the loaders' prologue)
  // fragment addresses inside a slot (bytes): piece p (rows 8 p .. 8 p + 7) at (p >> 2) kQuarter + (p & 3) kRow, quarter
  // k4 of the step 256 k4 bytes into it; inside a quarter [fk][row pair] (m-contiguous) or [k pair][row] (k-contiguous)
  const unsigned a_base_m = (unsigned)((2 * wm) * kQuarter + (fr >> 3) * kRow + (fk * 4 + ((fr & 7) >> 1)) * 16 + (fr & 1) * 8);
  const unsigned a_base_k = (unsigned)((2 * wm) * kQuarter + (fr >> 3) * kRow + (((fk >> 1) * 8 + (fr & 7)) * 16) + (fk & 1) * 8);
  const unsigned b_base = (unsigned)((2 * wn) * kQuarter + 4 * kRow + (fr >> 3) * kRow + (((fk >> 1) * 8 + (fr & 7)) * 16) + (fk & 1) * 8);
  char* cimg = lds + kRing * kSlot + (wn * 32 + fk) * kRow + (wm * 64 + fr) * 8;   // + (ni * 16 + 4 r) * kRow + mi * 128
  d4 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = d4{0.0, 0.0, 0.0, 0.0};

  double af[3][4], bf[3][2];   // [2]: the first fragments of the NEXT step, read behind the barrier
  auto read_frags = [&](int buf, const char* sp, unsigned abase, int k4) {
    const char* pa = sp + abase;
    const char* pb = sp + b_base;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) af[buf][mi] = *(const double*)(pa + k4 * 256 + (mi >> 1) * kQuarter + (mi & 1) * 2 * kRow);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bf[buf][ni] = *(const double*)(pb + k4 * 256 + ni * kQuarter);
  };
  auto read_frags_asm = [&](int buf, unsigned slot_addr, unsigned abase) {   // k4 = 0 of the slot at LDS byte address slot_addr
    const unsigned pa = slot_addr + abase, pb = slot_addr + b_base;
    asm volatile("ds_read_b64 %0, %1" : "=v"(af[buf][0]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:2304" : "=v"(af[buf][1]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:6912" : "=v"(af[buf][2]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:9216" : "=v"(af[buf][3]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1" : "=v"(bf[buf][0]) : "v"(pb) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:6912" : "=v"(bf[buf][1]) : "v"(pb) : "memory");
  };
  auto frags_arrived = [&](int buf) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(af[buf][0]), "+v"(af[buf][1]), "+v"(af[buf][2]), "+v"(af[buf][3]), "+v"(bf[buf][0]), "+v"(bf[buf][1])
                 :
                 : "memory");
  };
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[buf][ni], af[buf][mi], acc[ni][mi], 0, 0, 0);
  };
  // The swap at an item boundary: result out, accumulators cleared, 8 values at a time
  auto swap_all = [&]() {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) {
            *(double*)(cimg + (ni * 16 + 4 * (2 * r2 + rr)) * kRow + mi * 128) = acc[ni][mi][2 * r2 + rr];
            acc[ni][mi][2 * r2 + rr] = 0.0;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
  };
  // steps of the current item: ks counts from 0, the layout changes at step sw (relative to the item's first step), the
  // item ends after KS steps.  (One flat loop with ONE copy of the MFMAs, as in k_gemm3.)
  It first = it_first();
  int sw = g_sw(first) - g_lo(first);
  unsigned ab_cur = 0 < sw ? a_base_m : a_base_k;
  read_frags(2, lds, ab_cur, 0);
  read_frags(1, lds, ab_cur, 1);
  mfmas(2);
  int ks = 0, slot = 0;
  // (the MFMA waves know their item's step count from the iterator only for the first item; later ones through the flag
  // word: 1 + switch step, and the count is the same for every slice index except for rounding -- carried in the word's
  // upper half)
  int KS = g_hi(first) - g_lo(first);
  bool more = true;
#pragma clang loop unroll(disable)
  while (more) {
    const char* sp = lds + slot * kSlot;
    slot = slot == kRing - 1 ? 0 : slot + 1;
    read_frags(0, sp, ab_cur, 2);
    mfmas(1);
    read_frags(1, sp, ab_cur, 3);
    mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)"
                 :
                 : "v"(af[1][0]), "v"(af[1][1]), "v"(af[1][2]), "v"(af[1][3]), "v"(bf[1][0]), "v"(bf[1][1])
                 : "memory");
    __builtin_amdgcn_s_barrier();
    const char* sn = lds + slot * kSlot;
    const bool boundary = ks == KS - 1;
    ks = boundary ? 0 : ks + 1;
    // (the first step of an item is never past its switch unless the item starts there: decided below for a new item)
    unsigned ab_next = (boundary || ks < sw) ? a_base_m : a_base_k;
    int word = 0;
    if (boundary) {
      word = __builtin_amdgcn_readfirstlane(*(const __attribute__((address_space(3))) int*)(lds + kFlagOff));
      // word = 0: no further item; else (KS_next << 16 | (1 + sw_next)) with sw_next possibly <= 0 (a slice that starts
      // past the diagonal): biased by 0x4000
      const int swn = (word & 0xffff) - 0x4000;
      if (word != 0 && 0 >= swn) ab_next = a_base_k;
    }
    read_frags_asm(2, lds_base + (unsigned)(slot * kSlot), ab_next);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);                   // k4 = 3 of step g (at a boundary: the accumulators are final behind these)
    __builtin_amdgcn_sched_barrier(0);
    frags_arrived(2);
    read_frags(1, sn, ab_next, 1);
    if (boundary) {
      swap_all();
      more = word != 0;
      sw = (word & 0xffff) - 0x4000;
      KS = (int)((unsigned)word >> 16);
    }
    ab_cur = ab_next;
    mfmas(2);                   // k4 = 0 of step g + 1
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // the last result is in the image
}

}  // namespace

namespace {
bool g_symm3_any_size = false;   // debugging (sc_dbg_symm3_host): take every launch that qualifies, whatever its size
}

// Whether launch_symm3 takes X = A V for `count` symmetric matrices of order m with `split` K slices (see the header).
static bool symm3_takes(sc_ctx* ctx, int count, int m, int split, bool aligned16, bool any_size) {
  static const int env = [] { const char* e = getenv("SPRINGCRAFT_SYMM3"); return e ? atoi(e) : 1; }();
  if (env == 0 || count <= 0 || !aligned16) return false;
  if (m < 256 || (m & 1)) return false;
  if (split < 1 || split > 16 || ((m + 15) / 16 + 8) / split < 4) return false;   // (every item has a few K steps)
  if (ctx->num_cus < 256) return false;   // (the item order is built on 8 XCDs x 32 workgroups)
  const long long items = (long long)((m + 127) / 128) * split * count;
  if (env != 2 && !any_size && items < 256) return false;
  return items <= 0x3fffffffLL;
}

// X = A V on k_symm3 (records: a = A(0, 0), sa_i = 1, sa_k = lda; b = V, sb_k = 1, sb_j = ldb; c = X, ldc; split_stride);
// returns SC_OK when the launch went out, 1 when it is not one the kernel takes.
bool symm3_would_take(sc_ctx* ctx, int count, int m, int split, bool aligned16) {
  return symm3_takes(ctx, count, m, split, aligned16, g_symm3_any_size);
}

int launch_symm3(sc_ctx* ctx, const GemmDesc* d_desc, int count, int m, int split, bool aligned16, bool any_size) {
  // (any_size: the caller has decided with symm3_would_take for another record count -- the parts of a split batch)
  if (!symm3_takes(ctx, count, m, split, aligned16, any_size || g_symm3_any_size)) return 1;
  if (!sc_raise_dyn_lds(reinterpret_cast<const void*>(&k_symm3), kS3Lds)) return 1;
  if (!ctx->d_zeros) return 1;   // (16 KB of zeros, allocated with the context; the kernel is handed its middle)
  // (workgroups: one per CU, or 224 while the caller runs parts of the batch on several streams -- the other part's panel
  // QR and small products then find CUs beside this launch, as for k_gemm3's trailing update; SPRINGCRAFT_SYMM3_WGS)
  static const int env_wgs = [] { const char* e = getenv("SPRINGCRAFT_SYMM3_WGS"); return e ? std::max(8, std::min(256, atoi(e) / 8 * 8)) : 0; }();
  const int nwg = env_wgs > 0 ? env_wgs : (ctx->gemm3_side_by_side ? 224 : 256);
  S3Args A{d_desc, count, m, split, nwg, reinterpret_cast<const double*>(reinterpret_cast<const char*>(ctx->d_zeros) + 8192), {0}, 0};
  const int ks_all = (m + 15) / 16 + 8;   // (a last step that reaches beyond the matrix takes its pieces from the zeros)
  for (int q = 0; q <= split; ++q) A.gb[q] = (int)((long long)ks_all * q / split);
  static const int env_hot = [] { const char* e = getenv("SPRINGCRAFT_SYMM3_DBG_HOT"); return e ? atoi(e) : 0; }();
  A.hot = env_hot;
  hipLaunchKernelGGL(k_symm3, dim3((unsigned)nwg), dim3(1024), kS3Lds, ctx->stream, A);
  if (hipGetLastError() != hipSuccess) return 1;
  ++ctx->cnt_symm3_launches;
  return SC_OK;
}

// ---- debug entry for the GPU unit tests (tests/test_gemm_gpu.py; not part of the public C ABI): x = sym(a) v for `count`
// matrices on host data.  a: count x (m x m) column-major, only (row | 1) >= col is read; v, x: count x (m x 64).
extern "C" int sc_dbg_symm3_host(sc_ctx* ctx, const double* a, const double* v, double* x, int count, int m, int split) {
  if (!ctx || !a || !v || !x || count < 1 || m < 1 || split < 1) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  const size_t ea = (size_t)m * m, ev = (size_t)m * 64;
  char* base = nullptr;
  SC_HIP(ctx, hipMalloc((void**)&base, (ea + ev + ev * split) * count * 8 + sizeof(GemmDesc) * (size_t)count + 256));
  double* da = (double*)base;
  double* dv = da + ea * count;
  double* dx = dv + ev * count;
  GemmDesc* dd = (GemmDesc*)(dx + ev * split * count);
  int rc = SC_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == SC_OK) rc = sc_set_error(ctx, SC_ERR_HIP, "%s", hipGetErrorString(e)); };
  fail(hipMemcpy(da, a, ea * count * 8, hipMemcpyHostToDevice));
  fail(hipMemcpy(dv, v, ev * count * 8, hipMemcpyHostToDevice));
  fail(hipMemset(dx, 0xff, ev * split * count * 8));
  std::vector<GemmDesc> h((size_t)count);
  for (int z = 0; z < count; ++z) {
    GemmDesc D{};
    D.a = da + ea * z; D.b = dv + ev * z; D.c = dx + ev * split * z;
    D.m = m; D.n = 64; D.k = m; D.ldc = m; D.alpha = 1.0; D.beta = 0.0;
    D.sa_i = 1; D.sa_k = m; D.sb_k = 1; D.sb_j = m;
    D.split_stride = (long long)ev;
    h[(size_t)z] = D;
  }
  fail(hipMemcpy(dd, h.data(), sizeof(GemmDesc) * (size_t)count, hipMemcpyHostToDevice));
  if (rc == SC_OK) {
    g_symm3_any_size = true;
    const int took = launch_symm3(ctx, dd, count, m, split, true, true);
    g_symm3_any_size = false;
    if (took != SC_OK) rc = sc_set_error(ctx, SC_ERR_INVALID_ARG, "k_symm3 does not take m %d split %d", m, split);
    fail(hipStreamSynchronize(ctx->stream));
  }
  if (rc == SC_OK) {
    // the slices summed on the host
    std::vector<double> hx(ev * split * count);
    fail(hipMemcpy(hx.data(), dx, hx.size() * 8, hipMemcpyDeviceToHost));
    for (int z = 0; z < count; ++z)
      for (size_t e = 0; e < ev; ++e) {
        double s = 0.0;
        for (int q = 0; q < split; ++q) s += hx[(size_t)z * ev * split + (size_t)q * ev + e];
        x[(size_t)z * ev + e] = s;
      }
  }
  (void)hipFree(base);
  return rc;
}

extern "C" int sc_dbg_symm3_stamps(unsigned long long* out4) {
#ifdef SYMM3_STAMPS
  if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_symm3_stamps), 32) != hipSuccess) return 5;
  const unsigned long long z[4] = {0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_symm3_stamps), z, 32) == hipSuccess ? 0 : 5;
#else
  (void)out4;
  return 1;
#endif
}

// ---- debug / tuning entry (not part of the public C ABI): `count` matrices of order m on freshly allocated buffers, X = A V
// timed through k_symm3.  tools/symm3_bench.py
extern "C" int sc_dbg_symm3_bench(sc_ctx* ctx, int count, int m, int split, int iters, double* ms_out) {
  if (!ctx || count < 1 || iters < 1 || m < 256) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  const size_t ea = (size_t)m * m, ev = (size_t)m * 64;
  char* base = nullptr;
  SC_HIP(ctx, hipMalloc((void**)&base, (ea + ev + ev * split) * count * 8 + sizeof(GemmDesc) * (size_t)count + 256));
  double* da = (double*)base;
  double* dv = da + ea * count;
  double* dx = dv + ev * count;
  GemmDesc* dd = (GemmDesc*)(dx + ev * split * count);
  int rc = SC_OK;
  auto fail = [&](hipError_t e) { if (e != hipSuccess && rc == SC_OK) rc = sc_set_error(ctx, SC_ERR_HIP, "%s", hipGetErrorString(e)); };
  {
    // (random operands: the clock a launch holds depends on what the matrix pipes multiply -- zeros run ~10 % faster)
    std::vector<double> hv(std::max(ea, ev));
    unsigned long long sd = 88172645463325252ull;
    for (auto& x : hv) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; x = (double)((long long)(sd % 2001) - 1000) / 1000.0; }
    for (int z = 0; z < count; ++z) {
      fail(hipMemcpy(da + ea * z, hv.data(), ea * 8, hipMemcpyHostToDevice));
      fail(hipMemcpy(dv + ev * z, hv.data(), ev * 8, hipMemcpyHostToDevice));
    }
  }
  std::vector<GemmDesc> h((size_t)count);
  for (int z = 0; z < count; ++z) {
    GemmDesc D{};
    D.a = da + ea * z; D.b = dv + ev * z; D.c = dx + ev * split * z;
    D.m = m; D.n = 64; D.k = m; D.ldc = m; D.alpha = 1.0; D.beta = 0.0;
    D.sa_i = 1; D.sa_k = m; D.sb_k = 1; D.sb_j = m;
    D.split_stride = (long long)ev;
    h[(size_t)z] = D;
  }
  fail(hipMemcpy(dd, h.data(), sizeof(GemmDesc) * (size_t)count, hipMemcpyHostToDevice));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == SC_OK) {
    fail(hipEventCreate(&e0));
    fail(hipEventCreate(&e1));
    g_symm3_any_size = true;
    for (int it = 0; it < 2 && rc == SC_OK; ++it) rc = launch_symm3(ctx, dd, count, m, split, true, true) == SC_OK ? SC_OK : SC_ERR_INVALID_ARG;
    fail(hipEventRecord(e0, ctx->stream));
    for (int it = 0; it < iters && rc == SC_OK; ++it) rc = launch_symm3(ctx, dd, count, m, split, true, true) == SC_OK ? SC_OK : SC_ERR_INVALID_ARG;
    fail(hipEventRecord(e1, ctx->stream));
    g_symm3_any_size = false;
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
