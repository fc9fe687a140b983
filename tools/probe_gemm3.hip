// Probe for a role-split, persistent f64 GEMM: can the C traffic of a short-K product be hidden
// behind the MFMAs when (i) the operands are staged by LDS-DMA issued from loader waves that own no MFMA work and
// (ii) the NEXT tile's C is prefetched into LDS by LDS-DMA while the current tile's K loop runs?  (gfx950)
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe_gemm3.hip -o /tmp/probe_gemm3 && /tmp/probe_gemm3
//
// One workgroup per CU (16 waves): waves 0-3 = one MFMA wave per SIMD, 2 x 2 over a 128 x 64 tile (a wave owns 64 x 32 = 8
// accumulator tiles); waves 4-11 = operand loader waves; waves 12-15 = C waves.  K steps of 16; the operands of K step g
// live in ring slot g % 3.
//   * loader wave (q, parity): quarter q of the K steps with g % 2 == parity: 4 k-rows of A (1 KB each) + 2 pieces of B,
//     six 1-KB LDS-DMA instructions behind ONE write of M0 (a write to M0 waits for the wave's LDS-DMA in flight, so a
//     wave never has two groups in flight; two parities = two K steps in flight per quarter);
//   * the C tile of the NEXT output tile is fetched by the C waves' LDS-DMA (16 columns each, two groups of 8) into a 72 KB
//     LDS image while the K loop of the current tile runs; at the tile boundary an MFMA wave SWAPS its accumulators with
//     that image (result out, next C in: 32 ds_read_b64 + 32 ds_write_b64) and the C waves store the result from there
//     with 16-byte-per-lane stores during the next tile's first K step: the MFMA waves issue no global memory
//     instruction at all (version 1 stored from the accumulators: 32 global_store_dwordx2 per wave = 4-5 us per tile,
//     store-issue bound);
//   * one s_barrier per K step.
// Layouts: 0 = "NN" (A m-contiguous, B k-contiguous), 2 = "NT" (A m-contiguous, B n-contiguous) -- gemm_f64.h's names.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double __attribute__((address_space(1)))* gptr;
typedef const double __attribute__((address_space(3)))* lptr_c;

constexpr int kRow = 1152;              // bytes between consecutive 1-KB pieces (128 B of padding: bank phase)
constexpr int kQuarter = 6 * kRow;      // 4 A rows + 2 B pieces
constexpr int kSlot = 4 * kQuarter;     // one K step of 16
constexpr int kRing = 3;
constexpr int kCbuf = 64 * kRow;        // 64 columns of 128 rows
constexpr int kLdsBytes = kRing * kSlot + kCbuf;   // 156 672

struct Prob {
  const double* a; const double* b; double* c;
  long long lda, ldb, ldc;       // element strides (of k for A; of k (layout 2) or of j (layout 0) for B; of j for C)
  long long za, zb, zc;          // element strides between the matrices of the batch
  int M, N, K, Z;
  double alpha, beta;
};

__device__ __forceinline__ unsigned long long sgpr64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

#ifdef STAMPS
__device__ unsigned long long g_stamps[256 * 4];   // per workgroup (MFMA wave 0): total | in barriers | in swaps | steps
__device__ unsigned long long g_step[256 * 64];   // per workgroup: [ks < 32] cycles of K step ks of a tile | [32 + ks] of them in the barrier
#define STAMP(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
#else
#define STAMP(v)
#endif

template <int LAYOUT>
__global__ __launch_bounds__(1024, 1) void k_gemm3(Prob P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = P.K / 16;
  const int SM = P.M / 512, SN = P.N / 512;
  const int total_super = SM * SN * P.Z;
  const int xcd = blockIdx.x & 7, sl = blockIdx.x >> 3;
  const int my_tiles = total_super > xcd ? (total_super - xcd + 7) / 8 : 0;
  if (my_tiles == 0) return;
  const int G = my_tiles * KS;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  // Tiles of this workgroup: super-tile S = 8 t + xcd, S -> (z, sn, sm) with sm fastest.  Walked with adds and compares
  // only: an integer division is ~40 VALU instructions, and a helper wave's VALU instructions wait for issue slots of a
  // SIMD that an MFMA wave keeps busy (a C wave that divided in a K step was 1 500 cycles late for its barrier).
  struct TileIt {
    int z, sn, sm;
  };
  auto tile_first = [&]() {
    TileIt it{0, 0, xcd};
    while (it.sm >= SM) { it.sm -= SM; ++it.sn; }
    while (it.sn >= SN) { it.sn -= SN; ++it.z; }
    return it;
  };
  auto tile_next = [&](TileIt& it) {
    it.sm += 8;
    while (it.sm >= SM) { it.sm -= SM; ++it.sn; }
    while (it.sn >= SN) { it.sn -= SN; ++it.z; }
  };
  auto tile_tm = [&](const TileIt& it) { return it.sm * 4 + (sl & 3); };
  auto tile_tn = [&](const TileIt& it) { return it.sn * 8 + (sl >> 2); };

  if (w >= 4 && w < 12) {
    // ---------------------------------------------------------------------------------- operand loader waves
    const int d = w - 4, q = d & 3, parity = d >> 2;
    const unsigned voff_a = (unsigned)lane * 16u;
    unsigned voff_b;
    if (LAYOUT == 2) voff_b = (unsigned)(((lane >> 3) & 1) * P.ldb * 8 + ((lane >> 4) * 16 + (lane & 7) * 2) * 8);
    else voff_b = (unsigned)((lane & 7) * P.ldb * 8 + (lane >> 3) * 16);
    // groups are issued in order g = parity, parity + 2, ...: (tile, K step inside it) advance by two K steps at a time
    TileIt lt = tile_first();
    int l_ks = parity, l_slot = parity % kRing;
    while (l_ks >= KS) { l_ks -= KS; tile_next(lt); }
    unsigned long long ab_t = 0, bb_t = 0;
    bool l_new = true;
    auto issue_group = [&]() {
      if (l_new) {
        const int z = lt.z, tm = tile_tm(lt), tn = tile_tn(lt);
        ab_t = (unsigned long long)(size_t)(P.a + (size_t)z * P.za) + (unsigned long long)tm * 1024ull;
        if (LAYOUT == 2) bb_t = (unsigned long long)(size_t)(P.b + (size_t)z * P.zb) + (unsigned long long)tn * 512ull;
        else bb_t = (unsigned long long)(size_t)(P.b + (size_t)z * P.zb) + (unsigned long long)(tn * 64 + 16 * q) * P.ldb * 8ull;
        l_new = false;
      }
      const int k0 = l_ks * 16;
      const int slot_l = l_slot;
      // advance to this wave's next group
      l_ks += 2;
      if (l_ks >= KS) { l_ks -= KS; tile_next(lt); l_new = true; }
      l_slot += 2;
      if (l_slot >= kRing) l_slot -= kRing;
      const unsigned m0v = lds_base + (unsigned)(slot_l * kSlot + q * kQuarter + 2880);
      const unsigned long long row = (unsigned long long)P.lda * 8ull;
      const unsigned long long ab = ab_t + (unsigned long long)(k0 + 4 * q) * row;
      unsigned long long bb0, bb1;
      if (LAYOUT == 2) {
        bb0 = bb_t + (unsigned long long)(k0 + 4 * q) * P.ldb * 8ull;
        bb1 = bb0 + 2ull * P.ldb * 8ull;
      } else {
        bb0 = bb_t + (unsigned long long)k0 * 8ull;
        bb1 = bb0 + 8ull * P.ldb * 8ull;
      }
      // scalar bases with the instruction offsets taken out (the offset moves the LDS and the global address alike)
      const unsigned long long s0 = sgpr64(ab + 2880ull), s1 = sgpr64(ab + row + 1728ull), s2 = sgpr64(ab + 2 * row + 576ull),
                               s3 = sgpr64(ab + 3 * row - 576ull), s4 = sgpr64(bb0 - 1728ull), s5 = sgpr64(bb1 - 2880ull);
      asm volatile(
          "s_mov_b32 m0, %0\n\ts_nop 4\n\t"
          "global_load_lds_dwordx4 %1, %3 offset:-2880\n\t"
          "global_load_lds_dwordx4 %1, %4 offset:-1728\n\t"
          "global_load_lds_dwordx4 %1, %5 offset:-576\n\t"
          "global_load_lds_dwordx4 %1, %6 offset:576\n\t"
          "global_load_lds_dwordx4 %2, %7 offset:1728\n\t"
          "global_load_lds_dwordx4 %2, %8 offset:2880"
          :
          : "s"(m0v), "v"(voff_a), "v"(voff_b), "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5)
          : "memory");
    };
    if (parity < G) issue_group();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int g = 0; g < G; ++g) {
      if ((g & 1) == parity) {
        if (g + 2 < G) issue_group();
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_barrier();   // (the MFMA waves' last swap)
    return;
  }

  if (w >= 12) {
    // ------------------------------------------------------------------------------------------------ C waves
    // Wave cw owns columns 16 cw .. 16 cw + 15 of the C image and walks them a few per K step: a column's result (left
    // there by the MFMA waves' swap) is read into four registers and stored with ONE 1-KB row-contiguous store, and the
    // next tile's C for that column is fetched into the place that has just been read by ONE LDS-DMA instruction.  All
    // addresses are scalar arithmetic, the column number only selects the instruction's immediate offset: a helper
    // wave's VALU instructions have to find issue slots on a SIMD that an MFMA wave keeps busy (version 4 kept 16
    // columns in registers and predicated 16 unrolled stores: its register spills and predicates were ~20 VALU
    // instructions per step and made the whole workgroup wait ~1000 cycles at the barrier of every such step).
    const int cw = w - 12;
    const unsigned voff_c = (unsigned)lane * 16u;
    typedef double d2v __attribute__((ext_vector_type(2)));
    const unsigned long long col = (unsigned long long)P.ldc * 8ull;
    auto c_base = [&](const TileIt& it) {   // first of this wave's columns of the tile (scalar)
      return (unsigned long long)(size_t)(P.c + (size_t)it.z * P.zc) +
             ((unsigned long long)(tile_tn(it) * 64 + 16 * cw) * P.ldc + (unsigned long long)tile_tm(it) * 128) * 8ull;
    };
    // M0 of the group of eight columns j = 8 h .. 8 h + 7 (LDS-DMA lands at M0 + immediate + 16 lane)
    auto set_m0 = [&](int h) {
      const unsigned m0v = lds_base + (unsigned)(kRing * kSlot + (16 * cw + 8 * h + 3) * kRow + 576);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(m0v) : "memory");
    };
    auto dma_col = [&](unsigned long long tile_base, int j) {   // column j of the tile at tile_base -> image column j
      const int jj = j & 7;
      const unsigned long long a = tile_base + (unsigned long long)j * col;
      switch (jj) {
        case 0: { const unsigned long long b = a + 4032ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-4032" : : "v"(voff_c), "s"(b) : "memory"); break; }
        case 1: { const unsigned long long b = a + 2880ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2880" : : "v"(voff_c), "s"(b) : "memory"); break; }
        case 2: { const unsigned long long b = a + 1728ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-1728" : : "v"(voff_c), "s"(b) : "memory"); break; }
        case 3: { const unsigned long long b = a + 576ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:-576" : : "v"(voff_c), "s"(b) : "memory"); break; }
        case 4: { const unsigned long long b = a - 576ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:576" : : "v"(voff_c), "s"(b) : "memory"); break; }
        case 5: { const unsigned long long b = a - 1728ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:1728" : : "v"(voff_c), "s"(b) : "memory"); break; }
        case 6: { const unsigned long long b = a - 2880ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:2880" : : "v"(voff_c), "s"(b) : "memory"); break; }
        default: { const unsigned long long b = a - 4032ull; asm volatile("s_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 offset:4032" : : "v"(voff_c), "s"(b) : "memory"); break; }
      }
    };
    const char* img = lds + kRing * kSlot + (16 * cw) * kRow + lane * 16;
    auto out_col = [&](unsigned long long tile_base, int j) {   // image column j -> column j of the tile at tile_base
      const d2v v = *(const d2v*)(img + j * kRow);
      const unsigned long long a = tile_base + (unsigned long long)j * col;
      asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(voff_c), "v"(v), "s"(a) : "memory");
    };
    const bool with_c = P.beta != 0.0;
    TileIt t_cur = tile_first(), t_next = t_cur;
    tile_next(t_next);
    unsigned long long base_prev = 0, base_cur = c_base(t_cur), base_next = c_base(t_next);
    if (with_c) {
      set_m0(0);
      for (int j = 0; j < 8; ++j) dma_col(base_cur, j);
      set_m0(1);
      for (int j = 8; j < 16; ++j) dma_col(base_cur, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // columns per K step, first step of the two groups (the second group's M0 write waits for the first group's DMA in
    // flight: two idle steps in between), CW_VARIANT (ablations, results wrong): 3 no stores | 4 no C DMA | 5 neither
#ifndef CW_VARIANT
#define CW_VARIANT 0
#endif
    const int cps = KS >= 13 ? 2 : (KS >= 9 ? 4 : 8);
    const int gsteps = 8 / cps;
    const int first0 = 1, first1 = 1 + gsteps + (KS >= 9 ? 2 : 1);
    for (int t = 0; t < my_tiles; ++t) {
      const bool dma_ok = CW_VARIANT != 4 && CW_VARIANT != 5 && with_c && t + 1 < my_tiles;
      const bool st_ok = CW_VARIANT != 3 && CW_VARIANT != 5 && t > 0;
      for (int ks = 0; ks < KS; ++ks) {
        int j0 = -1;
        if (ks >= first0 && ks < first0 + gsteps) j0 = (ks - first0) * cps;
        if (ks >= first1 && ks < first1 + gsteps) j0 = 8 + (ks - first1) * cps;
        if (j0 >= 0) {
          if (dma_ok && (ks == first0 || ks == first1)) set_m0(j0 >> 3);
          for (int j = j0; j < j0 + cps; ++j) {
            if (st_ok) out_col(base_prev, j);
            if (dma_ok) {
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the column has been read: its place is free)
              dma_col(base_next, j);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (dma_ok && ks == KS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      base_prev = base_cur;
      base_cur = base_next;
      t_cur = t_next;
      tile_next(t_next);
      base_next = c_base(t_next);
    }
    __builtin_amdgcn_s_barrier();   // (the MFMA waves' last swap)
    for (int j = 0; j < 16; ++j) out_col(base_prev, j);
    return;
  }

  // ---------------------------------------------------------------------------------------------- MFMA waves
  const int wm = w & 1, wn = w >> 1;
  const int fr = lane & 15, fk = lane >> 4;
  __builtin_amdgcn_s_barrier();   // (the loaders' prologue)

  // fragment addresses inside a slot (bytes): ONE per-lane offset per operand, everything else is an immediate of the
  // LDS instruction (a tile row step is 128 B, a k4 step a quarter)
  const unsigned a_base = (unsigned)(fk * kRow + (wm * 64 + fr) * 8);
  const unsigned b_base = LAYOUT == 2 ? (unsigned)(4 * kRow + (fk >> 1) * kRow + (wn * 2) * 256 + (fk & 1) * 128 + fr * 8)
                                      : (unsigned)((2 * wn) * kQuarter + 4 * kRow + (fr >> 3) * kRow + (((fk >> 1) * 8 + (fr & 7)) * 16) + (fk & 1) * 8);
  constexpr int kBni = LAYOUT == 2 ? 256 : kQuarter, kBk4 = LAYOUT == 2 ? kQuarter : 256;
  char* cimg = lds + kRing * kSlot + (wn * 32 + fk) * kRow + (wm * 64 + fr) * 8;   // + (ni * 16 + 4 r) * kRow + mi * 128
  // alpha = 1, beta in {0, 1}: the accumulators hold C itself and the tile boundary needs no VALU work (f64 VALU
  // instructions queue behind the MFMAs in flight; flipping the sign bit of the B fragments for alpha = -1 -- two 32-bit
  // VALU instructions per k4 -- cost 6 % of the K loop: callers hand over an operand with the sign folded in instead)
  const bool with_c = P.beta != 0.0;
  d4 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi][r] = with_c ? *(const double*)(cimg + (ni * 16 + 4 * r) * kRow + mi * 128) : 0.0;
#ifdef STAMPS
  unsigned long long st_bar = 0, st_swap = 0;
  STAMP(t_begin)
  unsigned long long t_prev = t_begin;
#endif
  // The fragment reads run one k-step of 4 ahead of the MFMAs, ACROSS the step barrier: the barrier of K step g sits
  // between the MFMAs of its k4 = 2 and k4 = 3; behind it the reads of step g + 1's first fragments go out and the eight
  // MFMAs of k4 = 3 (operands in registers) cover their latency.  Slot g is not touched after the barrier.
  double af[3][4], bf[3][2];   // [2]: the first fragments of the NEXT step, read behind the barrier
  auto read_frags = [&](int buf, const char* sp, int k4) {
    const char* pa = sp + a_base;
    const char* pb = sp + b_base;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) af[buf][mi] = *(const double*)(pa + k4 * kQuarter + mi * 128);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) bf[buf][ni] = *(const double*)(pb + k4 * kBk4 + ni * kBni);
  };
  // The reads behind the barrier as inline asm: hipcc sinks ordinary loads below the MFMAs that are meant to cover their
  // latency (and a fence does not hold them).  Their completion is waited for by hand (frags_arrived).
  auto read_frags_asm = [&](int buf, unsigned slot_addr) {   // k4 = 0 of the slot at LDS byte address slot_addr
    const unsigned pa = slot_addr + a_base, pb = slot_addr + b_base;
    asm volatile("ds_read_b64 %0, %1" : "=v"(af[buf][0]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:128" : "=v"(af[buf][1]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:256" : "=v"(af[buf][2]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:384" : "=v"(af[buf][3]) : "v"(pa) : "memory");
    asm volatile("ds_read_b64 %0, %1" : "=v"(bf[buf][0]) : "v"(pb) : "memory");
    if (LAYOUT == 2) asm volatile("ds_read_b64 %0, %1 offset:256" : "=v"(bf[buf][1]) : "v"(pb) : "memory");
    else asm volatile("ds_read_b64 %0, %1 offset:6912" : "=v"(bf[buf][1]) : "v"(pb) : "memory");
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
  // An iteration = K step g from its k4 = 1 on, plus the k4 = 0 of step g + 1 (the tail of the LAST iteration works on a
  // slot and on accumulators nobody needs).  At a tile boundary the tail is where the swap goes: accumulator by
  // accumulator pair -- result out, next C in -- between the first MFMAs of the next tile, which only touch pairs that
  // have been swapped already.
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
  read_frags(2, lds, 0);
  read_frags(1, lds, 1);
  mfmas(2);
  int ks = 0, slot = 0;
#pragma clang loop unroll(disable)
  for (int g = 0; g < G; ++g) {
    {
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
      STAMP(t_b0)
      __builtin_amdgcn_s_barrier();
#ifdef STAMPS
      STAMP(t_b1)
      st_bar += t_b1 - t_b0;
      if (w == 0 && lane == 0 && ks < 32) {
        atomicAdd(&g_step[blockIdx.x * 64 + ks], t_b1 - t_prev);
        atomicAdd(&g_step[blockIdx.x * 64 + 32 + ks], t_b1 - t_b0);
      }
      t_prev = t_b1;
#endif
      const char* sn = lds + slot * kSlot;
      const bool boundary = ks == KS - 1;
      ks = boundary ? 0 : ks + 1;
      read_frags_asm(2, lds_base + (unsigned)(slot * kSlot));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(1);                   // k4 = 3 of step g (at a boundary: the accumulators are final behind these)
      __builtin_amdgcn_sched_barrier(0);
      frags_arrived(2);
      read_frags(1, sn, 1);
      if (boundary) {             // ONE copy of the MFMAs: a second one in a branch sends the accumulators through scratch
        STAMP(t_s0)
        swap_all();
#ifdef STAMPS
        STAMP(t_s1)
        st_swap += t_s1 - t_s0;
#endif
      }
      mfmas(2);                   // k4 = 0 of step g + 1
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // the last result is in the image
#ifdef STAMPS
  STAMP(t_end)
  if (w == 0 && lane == 0) {
    g_stamps[blockIdx.x * 4 + 0] = t_end - t_begin;
    g_stamps[blockIdx.x * 4 + 1] = st_bar;
    g_stamps[blockIdx.x * 4 + 2] = st_swap;
    g_stamps[blockIdx.x * 4 + 3] = (unsigned long long)G;
  }
#endif
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s, %d CUs; k_gemm3: 128 x 64 tiles, 4 MFMA + 8 loader + 4 C waves, LDS %d B\n", p.name, cus, kLdsBytes);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm3<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm3<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  struct Case { const char* name; int layout, M, N, K, Z; };
  const Case cases[] = {
      {"NN update  6144 x 6144, K = 128, batch 8", 0, 6144, 6144, 128, 8},
      {"NN update  6144 x 6144, K = 256, batch 8", 0, 6144, 6144, 256, 8},
      {"NN update  6144 x 6144, K = 512, batch 8", 0, 6144, 6144, 512, 8},
      {"NT (syr2k) 6144 x 6144, K = 128, batch 8", 2, 6144, 6144, 128, 8},
      {"NT (syr2k) 6144 x 6144, K = 256, batch 8", 2, 6144, 6144, 256, 8},
      {"NT (syr2k) 6144 x 6144, K = 512, batch 8", 2, 6144, 6144, 512, 8},
      {"NN square  6144^3, batch 1", 0, 6144, 6144, 6144, 1},
  };
  for (const Case& cs : cases) {
    const size_t ea = (size_t)cs.M * cs.K, eb = (size_t)cs.K * cs.N, ec = (size_t)cs.M * cs.N;
    double *a, *b, *c;
    CK(hipMalloc(&a, ea * cs.Z * 8));
    CK(hipMalloc(&b, eb * cs.Z * 8));
    CK(hipMalloc(&c, ec * cs.Z * 8));
    std::vector<double> ha(ea * cs.Z), hb(eb * cs.Z), hc(ec * cs.Z);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)((long long)(s % 2001) - 1000) / 1000.0; };
    for (auto& x : ha) x = rnd();
    for (auto& x : hb) x = rnd();
    for (auto& x : hc) x = rnd();
    CK(hipMemcpy(a, ha.data(), ha.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(c, hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
    Prob P{a, b, c, cs.M, cs.layout == 2 ? cs.N : cs.K, cs.M, (long long)ea, (long long)eb, (long long)ec, cs.M, cs.N, cs.K, cs.Z, 1.0, 1.0};
    auto launch = [&]() {
      if (cs.layout == 2) hipLaunchKernelGGL(k_gemm3<2>, dim3(cus), dim3(1024), kLdsBytes, 0, P);
      else hipLaunchKernelGGL(k_gemm3<0>, dim3(cus), dim3(1024), kLdsBytes, 0, P);
    };
    launch();
    CK(hipDeviceSynchronize());
    std::vector<double> out(ec * cs.Z);
    CK(hipMemcpy(out.data(), c, out.size() * 8, hipMemcpyDeviceToHost));
    double maxerr = 0.0;
    for (int tcase = 0; tcase < 256; ++tcase) {
      const int z = tcase % cs.Z;
      const int i = (int)((tcase * 7919ull + 13) % cs.M), j = (int)((tcase * 104729ull + 5) % cs.N);
      double ref = 0.0;
      for (int k = 0; k < cs.K; ++k) {
        const double av = ha[(size_t)z * ea + (size_t)k * cs.M + i];
        const double bv = cs.layout == 2 ? hb[(size_t)z * eb + (size_t)k * cs.N + j] : hb[(size_t)z * eb + (size_t)j * cs.K + k];
        ref += av * bv;
      }
      ref = hc[(size_t)z * ec + (size_t)j * cs.M + i] + ref;
      maxerr = fmax(maxerr, fabs(out[(size_t)z * ec + (size_t)j * cs.M + i] - ref));
    }
#ifdef STAMPS
    { std::vector<unsigned long long> z(256 * 64, 0ull); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_step), z.data(), z.size() * 8)); }
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 6;
    for (int it = 0; it < 2; ++it) launch();
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double tf = 2.0 * cs.M * cs.N * (double)cs.K * cs.Z / (ms * 1e-3) * 1e-12;
    printf("%-44s %8.3f ms  %6.2f TFLOP/s = %.3f of 78.6   max err %.2e\n", cs.name, ms, tf, tf / 78.6, maxerr);
#ifdef STAMPS
    {
      std::vector<unsigned long long> hs(256 * 4);
      CK(hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(g_stamps), hs.size() * 8));
      std::vector<double> tot, bar, swp;
      for (int wg = 0; wg < 256; ++wg) {
        const double steps = (double)hs[wg * 4 + 3];
        if (steps <= 0) continue;
        tot.push_back(hs[wg * 4] / steps); bar.push_back(hs[wg * 4 + 1] / steps);
        swp.push_back(hs[wg * 4 + 2] / (steps / (cs.K / 16)));
      }
      auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      printf("      MFMA wave 0, median over workgroups: %.0f cycles per K step (2048 = the MFMAs alone), %.0f of them in the barrier; %.0f per swap\n",
             med(tot), med(bar), med(swp));
      std::vector<unsigned long long> hk(256 * 64);
      CK(hipMemcpyFromSymbol(hk.data(), HIP_SYMBOL(g_step), hk.size() * 8));
      const int ksn = std::min(32, cs.K / 16);
      printf("      per K step of a tile (barrier to barrier | of it waiting), workgroup 8:");
      const double tiles = (double)hs[8 * 4 + 3] / (cs.K / 16) * (2 + iters);
      for (int k = 0; k < ksn; ++k) printf(" %d:%.0f|%.0f", k, hk[8 * 64 + k] / tiles, hk[8 * 64 + 32 + k] / tiles);
      printf("\n");
    }
#endif
    fflush(stdout);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c));
  }
  return 0;
}
