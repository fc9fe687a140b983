// Grouped / batched float64 GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64).
//
//     C[i,j] = beta * C[i,j] + alpha * sum_k A(i,k) * B(k,j)          (column-major C, ldc)
//
// A(i,k) = a[i*sa_i + k*sa_k + (a_kidx ? a_kidx[k]*sa_k ... see below)], B(k,j) = b[k*sb_k + j*sb_j]:
// arbitrary strides, so NN / NT / TN products and sub-matrix views need no copies.  One of
// (sa_i, sa_k) and one of (sb_k, sb_j) must be 1 (the loader runs its lanes along that axis).
//
// Problems are described by GemmDesc records in DEVICE memory, so sizes may be produced by a
// previous kernel (the divide & conquer merges: the number of non-deflated eigenvalues is only
// known on the device).  blockIdx.z selects the record; blocks outside m x n exit at once.
#pragma once

#include <cstdint>

#include "common.h"

struct GemmDesc {
  const double* a;
  const double* b;
  double* c;
  int m, n, k;
  long long sa_i, sa_k;   // element strides of A(i,k)
  long long sb_k, sb_j;   // element strides of B(k,j)
  long long ldc;          // column stride of C (row stride is 1)
  double alpha, beta;
  const int* a_kidx;      // optional gather on A's k axis: A(i,k) = a[i*sa_i + a_kidx[k]*sa_k]
  const int* b_kidx;      // optional gather on B's k axis: B(k,j) = b[b_kidx[k]*sb_k + j*sb_j]
  const int* c_jidx;      // optional scatter on C's column axis: C(:, j) lives at column c_jidx[j]
  int lower_only;         // 1: store only elements with (row + row_off) >= (col + col_off) (SYR2K); 2: ((row + row_off) | 1) >= ...
  int row_off, col_off;
  long long split_stride; // split-K launches: slice s writes alpha*partial to C + s*split_stride (beta ignored)
  int a_tri;              // launches with tri = true: 1: A(i,k) counts only for i >= k; 2: only for k > i
};

// (`tile` argument of launch_gemm_f64: a leftover of round 1's kernel, ignored; k_gemm2 picks its block tile from the
// launch size)
constexpr int kGemmTile = 3;

// Launch `count` problems (records d_desc[0..count)); max_m / max_n bound the grid.
// tile: ignored (see kGemmTile).
// split_k > 1: every record is cut into split_k K-slices (all records of a launch share it).
// gather: the records carry a_kidx / b_kidx lists (D&C merges; layout kGemmAmBk only).
// tri: the records carry a_tri masks (lower-stored symmetric A; band reduction; layouts kGemmAmBk / kGemmAkBk), no gathers.
// layout (required): which axis of each operand has stride 1 in ALL records of the launch (the records live in device
// memory, the host cannot look): kGemmAmBk "NN" (sa_i = 1, sb_k = 1), kGemmAkBk "TN" (sa_k = 1, sb_k = 1), kGemmAmBn
// "NT" (sa_i = 1, sb_j = 1).  The block tile is chosen from the launch size.
// lower_grid: every record of the launch is square, lower_only with row_off = col_off = 0 (the SYR2K of the band
// reduction): only the tiles on and below the diagonal are launched (layout kGemmAmBn, no split-K).
constexpr int kGemmAmBk = 0, kGemmAkBk = 1, kGemmAmBn = 2;
int launch_gemm_f64(sc_ctx* ctx, const GemmDesc* d_desc, int count, int max_m, int max_n, int tile,
                    int split_k = 1, bool gather = false, bool tri = false, int layout = -1, bool lower_grid = false);

// The role-split persistent kernel k_gemm3 (gemm3.hip) for launches whose records share (m, n, k): returns SC_OK when it
// took the launch, 1 when the launch is not one it takes (the caller then uses launch_gemm_f64).  alpha must be 1 and beta
// 0 or 1 -- callers fold a sign into an operand --; aligned16: every operand pointer and leading dimension of the
// records keeps 16-byte alignment (the host built them).  SPRINGCRAFT_GEMM3 = 0 switches it off, = 2 takes every
// launch that qualifies whatever its size (tests).
// lower: 0 all of C; 1 the lower triangle (as lower_only = 1 with row_off = col_off = 0, m == n); 2 as lower_only = 2.
int launch_gemm3_uniform(sc_ctx* ctx, const GemmDesc* d_desc, int count, int m, int n, int k, int layout, int lower,
                         double alpha, double beta, bool aligned16);
// (the launcher's decision without the launch: callers that lay out their records differently for the two kernels)
bool gemm3_would_take(sc_ctx* ctx, int count, int m, int n, int k, int layout, int lower, double alpha, double beta,
                      bool aligned16);

// The symmetric product X = A V (A lower stored, V and X with 64 columns) of the band reduction on k_symm3 (symm3.hip):
// SC_OK when it took the launch, 1 when the launch is not one it takes (m a multiple of 16, 16-byte aligned operands,
// enough tiles; SPRINGCRAFT_SYMM3 = 0 switches it off, = 2 takes every launch that qualifies).  split: K slices, slice s
// into c + s * split_stride.  The kernel reads A where (row | 1) >= col: see its header for what the writers of A keep.
// any_size: skip the "enough work items" part of the decision (the caller made it for another record count).
int launch_symm3(sc_ctx* ctx, const GemmDesc* d_desc, int count, int m, int split, bool aligned16, bool any_size = false);
bool symm3_would_take(sc_ctx* ctx, int count, int m, int split, bool aligned16);
