// Internal interfaces between the stages of the batched symmetric eigensolver
//   tridiag.hip (K5)  ->  stedc.hip (K6)  ->  backtransform.hip (K7),  driven by eigh.hip.
#pragma once

#include "common.h"
#include "gemm_f64.h"

// Per-matrix workspace slab of the tridiagonalisation (offsets in doubles from the slab base).
struct TriLayout {
  int n, nb;
  long long slab;    // slab stride between consecutive matrices of the batch
  long long vw;      // [V|W] panel, n x 2nb, column-major ld n
  long long wv;      // [W|V] panel, n x 2nb
  long long xraw;    // updated column a (n)
  long long ypart;   // SYMV partial products, nt x n
  long long dpart;   // partial dots V^T v / W^T v, nt x 2nb
  long long npart;   // partial squared norms (<= n/256 + 1)
  long long wvpart;  // partial w~^T v
  long long d, e, tau;  // tridiagonal + reflector scalars (n each)
  long long hscale;     // [0] = 1/(alpha - beta) of the current column (written by k_symv_tiles);
                        // [1] = bit pattern of max |a_ij|, [2] = power-of-two factor the matrix was scaled by
  long long rctl;       // k_sytrd_resident: control words (16 ints), directly followed by
  long long rrec;       // its exchange records, 3 buffers x (n rounded up to 64) x 16 bytes
};
// bytes from rctl that a solve sets to 0xFF before k_sytrd_resident (control words "undecided", records "empty")
inline size_t tri_resident_bytes(int n) { return sizeof(double) * (8 + (size_t)6 * ((n + 63) / 64 * 64)); }

// d_a: (batch) n x n column-major, lower triangle valid after the mirror pass; on exit column c holds
// v_c (explicit leading 1) in rows c+1.., and ws holds d, e, tau.
// NumPy (row-major) lower triangle -> column-major lower triangle, in place
int mirror_lower_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch);
// mirror + scaling of badly scaled matrices (largest |entry| outside [1e-100, 1e100]) by a power of two; the factor
// stays in the tri slab and unscale_values_batched divides the computed eigenvalues by it.  Call before
// tridiag_batched / sytrd_2stage_batched (neither mirrors by itself).
int prepare_matrix_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, double* d_tri_ws,
                           const TriLayout& L);
int unscale_values_batched(sc_ctx* ctx, double* d_w, long long stride_w, int m, int batch, const double* d_tri_ws,
                           const TriLayout& L);

int tridiag_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, double* d_ws,
                    const TriLayout& L, const GemmDesc* d_syr2k_descs, float* ms_symv, float* ms_syr2k);
// whether tridiag_batched hands the (trailing) matrix of a one-matrix solve to k_sytrd_resident on this context (ctx may
// be null: the environment's answer)
bool resident_enabled(const sc_ctx* ctx);

// ---- divide & conquer ------------------------------------------------------------------------------
struct DcNode {
  int lo, mid, hi;
};

struct DcLayout {
  int n;
  int leaf_max;      // leaves have at most this many rows
  long long slab;    // per-matrix stride (doubles) of the D&C slab
  long long dd, ee;  // scaled copies of d, e (n each)
  long long w0, w1;  // eigenvalue ping-pong (n each)
  long long z;       // rank-one vector (n)
  long long dl, zz;  // non-deflated poles / weights, compacted per node at [lo, lo+K)
  long long ddef;    // deflated eigenvalues per node at [lo, lo+ndef)
  long long lam;     // new roots per node (n)
  long long tauv;    // secular root offsets (n)
  long long zhat;    // Gu-Eisenstat weights (n)
  long long rot_c, rot_s;  // rotation list (n each)
  long long scale;   // [0] = norm used for scaling
  // integer arrays live in the same slab, offsets in doubles (each int array padded to n doubles)
  long long src;     // int[n]: source column of non-deflated k (at lo+k); of deflated e (at lo+K+e)
  long long org;     // int[n]: origin pole of each root
  long long dest;    // int[n]: destination column of non-deflated k (at lo+k) / deflated (at lo+K+e)
  long long rot_a, rot_b;  // int[n] each: rotated column pairs
  long long ktop_src, ktop_k;  // int[n] each: columns that are non-zero in the node's TOP half: source column / pole index
  long long kbot_src, kbot_k;  // same for the BOTTOM half
  long long cnt;     // int[2 * max_nodes]: per node K and number of rotations
};

// Eigen-decomposition of `batch` symmetric tridiagonal matrices (d, e in the TriLayout slab).
// On exit: d_w (batch, n) ascending eigenvalues; eigenvectors (columns) in d_q_out (batch, n, n).
// d_q_tmp / d_u: two more (batch, n, n) work matrices.
int stedc_batched(sc_ctx* ctx, int n, int batch, const double* d_tri_ws, const TriLayout& TL,
                  double* d_dc_ws, const DcLayout& DL, double* d_w, long long stride_w,
                  double* d_q_out, double* d_q_tmp, double* d_u, long long stride_q,
                  GemmDesc* d_merge_descs /* 2 * dc_max_nodes * batch */);
size_t dc_slab_doubles(int n, DcLayout* out);
int dc_max_nodes(int n, int leaf_max);

// Eigenvalues only: Sturm-sequence bisection (one thread per eigenvalue).
int sturm_bisect_batched(sc_ctx* ctx, int n, int batch, const double* d_tri_ws, const TriLayout& TL,
                         double* d_w, long long stride_w);

// ---- back-transformation -----------------------------------------------------------------------------
// Z <- Q_H Z with the reflectors stored in d_a (see tridiag_batched) and tau in the tri slab.
// off: row distance of a reflector's unit entry from the diagonal (1: one-stage; 64: stage 1 of the two-stage path).
struct BtLayout {
  int n, nbt, splits;
  long long slab;
  long long vc;     // (unused)
  long long gram;   // per block: split-K slabs of V^T V, splits_g x nbt x nbt
  long long t;      // per block: nbt x nbt triangular factor
  long long g12;    // per block: summed Gram block V1^T V2 (128 x 128)
  long long tx;     // per block: (V1^T V2) T2
  long long w1;     // split-K slabs of V^T Z: splits x nbt x n
  long long w2;     // T * sum(w1): nbt x n
  int splits_g;
};
size_t bt_slab_doubles(int n, BtLayout* out);
int bt_desc_count(int n, int batch);
// d_a is modified (cleaned in place); d_vt: (batch, n, n) scratch that receives V T.
int backtransform_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch,
                          const double* d_tri_ws, const TriLayout& TL, double* d_bt_ws,
                          const BtLayout& BL, double* d_z, long long stride_z, int ncols, double* d_vt,
                          GemmDesc* d_descs /* bt_desc_count(n, batch) records */, int off = 1, int phase = 0);

// ---- two-stage tridiagonalisation (twostage.hip) ----------------------------------------------------------------
struct SbLayout {
  int n;
  long long slab;
  long long vw, wv;   // [V|W], [W|V] panels of TWO consecutive panels, n x 256 (the trailing update takes them together)
  long long xv;       // [X1|X2|V], n x 192
  long long qrpart, qrpiv;   // panel-QR partial Gram rows / pivot row (two copies each)
  long long qrpart8;         // blocked panel QR: per-chunk partial products of an inner block, nchunk x 8 x 64
  long long xsplit;   // K slices of X1 and X2 when the SYMM runs split-K: 2 x symm_split x (n x 64)
  int symm_split;     // 1: no split
  long long small;    // split-K slices of V^T [X1|X2|V]
  long long cmat;     // [T; T; -S/2], 192 x 64
  long long small2;   // split-K slices of [W1|V1]^T V2 (second panel of a pair), 128 x 64 each
  long long p2;       // their sum
  long long ab;       // band storage 128 x n
  int ngroups;        // sweep groups (64 sweeps each)
  long long ndia;     // diamonds
  long long vd;       // diamonds: V row-major (128 rows x 64 sweeps)
  long long frag;     // diamonds: MFMA fragments of V^T and -(V T), 160 x 64 doubles each (k_dia_tfactor2 -> k_bt2_apply)
  long long tau2;     // ndia x 64
};
// batch: matrices of the solve (few matrices: the SYMM of the band reduction is cut into K slices, which need room)
size_t sb_slab_doubles(int n, int batch, SbLayout* out);
int sb_desc_count(int n, int batch);
int sb_band_width();
// d_a: column-major lower triangle valid.  On exit d, e (and the stage-1 tau) are in the tri slab, the stage-1
// reflectors in A (unit entry of column c at row c + sb_band_width()), the stage-2 reflectors in the sb slab.
int sytrd_2stage_batched(sc_ctx* ctx, double* d_a, long long stride_a, int n, int batch, double* d_tri_ws,
                         const TriLayout& TL, double* d_sb_ws, const SbLayout& SL, int* d_dia_off /* n/64 + 2 ints */,
                         GemmDesc* d_descs /* sb_desc_count */, float* ms_stage1, float* ms_stage2,
                         double* d_band_copy = nullptr);
// Z <- Q2 Z (stage-2 reflectors): bt2_prepare (T factors; independent of Z, may run on another stream), then bt2_batched
int bt2_prepare(sc_ctx* ctx, int n, int batch, double* d_sb_ws, const SbLayout& SL, hipStream_t st);
int bt2_batched(sc_ctx* ctx, int n, int batch, double* d_sb_ws, const SbLayout& SL, const int* d_dia_off, double* d_z,
                long long stride_z, int ncols, float* ms_fused = nullptr);

// ---- partial spectrum (stein.hip) -----------------------------------------------------------------------
size_t stein_workspace_doubles(int n, int m);
int stein_batched(sc_ctx* ctx, int n, int batch, const double* d_tri_ws, const TriLayout& TL, int il, int iu,
                  double* d_w, long long stride_w, double* d_x, long long stride_x, double* d_ws,
                  GemmDesc* d_descs /* 2 * batch */);
