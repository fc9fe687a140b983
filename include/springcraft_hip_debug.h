/*
 * springcraft_hip_debug.h -- tuning / diagnostic entry points of libspringcraft_hip.so.
 *
 * NOT part of the drop-in boundary (that is springcraft_hip.h): nothing in the product path calls these; they exist for
 * the scripts under tools/ and may change between rounds.  They are exported from the same library so that the kernels
 * they exercise are exactly the ones the product launches.
 */
#ifndef SPRINGCRAFT_HIP_DEBUG_H
#define SPRINGCRAFT_HIP_DEBUG_H

#include "springcraft_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Times `iters` launches of one grouped-GEMM shape on buffers it allocates itself and checks the result against a host
 * product (max abs error).  mode 0: C = A B (A m x k, B k x n, column-major); 1: lower triangle of C += A B^T (the
 * SYR2K shape, B stored n x k); 2: C = A^T B (A stored k x m).  tile 10..13: the k_gemm2 tilings (automatic,
 * 128x128, 128x64, 64x64).  Returns an SC_* code.  tools/gemm2_bench.py */
int sc_dbg_gemm_bench(sc_ctx* ctx, int m, int n, int k, int mode, int tile, int split_k, int iters, int beta_one,
                      double* ms_out, double* max_err_out);

/* ONE launch of one GEMM record on host data (a, b laid out as in sc_dbg_gemm_bench for `mode`; c: m x n column-major,
 * in / out; with split_k > 1 the K slices are summed into c on the host and beta must be 0; lower_grid as the launcher
 * takes it).  The GPU unit tests of the launch paths compare the result with NumPy.  tests/test_gemm_gpu.py */
int sc_dbg_gemm_host(sc_ctx* ctx, const double* a, const double* b, double* c, int m, int n, int k, int mode, int tile,
                     int split_k, double alpha, double beta, int lower_grid);

/* `count` products of ONE shape on host data through k_gemm3, the role-split persistent kernel of the short-K updates
 * (csrc/gemm3.hip), whatever their size: a count x (m x k) column-major; b count x (k x n) column-major (layout 0,
 * kGemmAmBk) or count x (n x k) column-major (layout 2, kGemmAmBn); c count x (m x n) column-major, in / out; C = beta C
 * + A B with beta 0 or 1; lower: only entries on / below the diagonal are written (m == n).  Returns
 * SC_ERR_INVALID_ARG for a shape the kernel does not take.  tests/test_gemm_gpu.py */
int sc_dbg_gemm3_host(sc_ctx* ctx, const double* a, const double* b, double* c, int count, int m, int n, int k, int layout,
                      int lower, double beta);

/* x = sym(a) v through k_symm3 (csrc/symm3.hip), the band reduction's symmetric product as one role-split launch:
 * `count` matrices a (m x m, column-major; read where (row | 1) >= col), v and x count x (m x 64) column-major, `split`
 * K slices (summed on the host).  m a multiple of 16.  tests/test_gemm_gpu.py */
int sc_dbg_symm3_host(sc_ctx* ctx, const double* a, const double* v, double* x, int count, int m, int split);
/* Library built with -DSYMM3_STAMPS (else returns 1): out4 = shader cycles the first loader wave of every k_symm3 workgroup
 * spent {waiting for its LDS-DMA group to land, at the step barriers, in its loop}, and the steps it waited in; reset by
 * the call.  tools/r06_symm3_stamps.sh */
int sc_dbg_symm3_stamps(unsigned long long* out4);
/* `count` random matrices of order m (even, >= 256), X = A V through k_symm3 with `split` K slices, `iters` timed launches:
 * average milliseconds per launch.  tools/symm3_bench.py */
int sc_dbg_symm3_bench(sc_ctx* ctx, int count, int m, int split, int iters, double* ms_out);

/* `count` matrices of one shape on freshly allocated buffers, C += A B (lower != 0: the lower triangle, m == n), timed
 * through k_gemm3 (kernel 3; order 0 / 1: its flat / per-XCD super-tile order) or through k_gemm2 (kernel 2) with the same
 * records: ms_out = milliseconds per launch.  tools/gemm3_shapes.py */
int sc_dbg_gemm3_bench(sc_ctx* ctx, int count, int m, int n, int k, int layout, int lower, int kernel, int order, int iters,
                       double* ms_out);

/* s_memtime segment sums of k_gemm2 (library built with -DGEMM_STAMPS; returns 1 otherwise): out4[0..2] =
 * shader cycles summed over waves in the prologue, the K loop and the C epilogue, out4[3] = waves counted; `reset`
 * zeroes the sums after the read.  tools/gemm_stamps.py */
int sc_dbg_gemm_stamps(unsigned long long* out4, int reset);

/* Per-workgroup timeline of k_gemm2 (same diagnostic build): up to `max_records` records of 6 values
 * {hw id (XCC_ID << 32 | HW_ID), t_start, t_loop, t_epilogue, t_end, blockIdx linear} of the workgroups launched since
 * the last reset; returns the number of records written through *count.  tools/gemm_trace.py */
int sc_dbg_gemm_trace(unsigned long long* out, int max_records, int* count, int reset);

/* Band (128 x n, AB(i,j) at [(i-j) + 128 j]) after stage 1 and the tridiagonal (d, e) after stage 2 of ONE host matrix
 * (n x n, NumPy layout, lower triangle read; n >= 256).  tools/check_two_stage.py */
int sc_dbg_two_stage(sc_ctx* ctx, const double* a, int n, double* band_out, double* d_out, double* e_out);

/* Per-wave s_memtime segment sums of k_bt2_apply (library built with -DBT2_STAMPS; returns 1 otherwise):
 * out[64 workgroups][8 waves][16 sums + diamond count].  tools/bt2_stamps.py */
int sc_dbg_bt2_stamps(unsigned long long* out);

/* Shader clock while the most recent k_bt2_apply / k_bt2_role launch ran (library built with -DBT2_CLOCK; returns 1
 * otherwise): out2 = {shader cycles (s_memtime), 100 MHz ticks (s_memrealtime)} over the life of wave 0 of workgroup 0.
 * tools/bt2_clock.py */
int sc_dbg_bt2_clock(unsigned long long* out2);

/* s_memtime in front of every MFMA of one diamond of k_bt2_apply (library built with -DBT2_TRACE; returns 1 otherwise):
 * out[8 waves][2 halves][81].  tools/bt2_trace.py */
int sc_dbg_bt2_trace(unsigned long long* out);

/* s_memtime sums of k_bulge_step over all tasks since the last call (library built with -DBULGE_STAMPS; returns 1
 * otherwise): out6 = {cycles from task start until the off-diagonal block E is in LDS, until E is stored [both: tasks
 * with k > 0], until the diagonal block D is in LDS, until the end; tasks; tasks with k > 0}.  tools/bulge_stamps.py */
int sc_dbg_bulge_stamps(unsigned long long* out6);
/* The same for the pair form of the persistent chase (k_bulge_pair; library built with -DPAIR_STAMPS, else returns 1):
 * 48 entries.  [0..15] thread 0 (team A): [0..6] = cycles between the barriers of a common step (wait + [0], block reads +
 * [1], E right update + reflector, column sums + D image, E left update + D products, w, D update), [8] = steps, [9] =
 * common steps, [12] = block reads alone; [16..31] thread 256 (team B), same layout, [23] its look at the predecessor pair, [30] steps in which it polled, [26] slot reads, [27] store drain,
 * [29] D update + stores; [32..47] lane 0 of the first loader wave (k_bulge_pair<1>): [32] wait for E, [33] barrier [0],
 * [34] [1] [2], [35] wait for D, [36] [3] [4], [37] E requests, [38] [5] [6] [7], [39] D requests, [40] steps counted. */
int sc_dbg_pair_stamps(unsigned long long* out48);
/* The same for the persistent chase with one sweep per workgroup (k_bulge_chase; library built with -DCHASE_STAMPS, else
 * returns 1): out16[1..11] = cycles of all waves between eleven points of a task, summed over all tasks since the last call
 * (the segments are listed at the macro in twostage.hip), [15] = waves x tasks.  tools/chase_stamps.py */
int sc_dbg_chase_stamps(unsigned long long* out16);

/* Persistent bulge chase of this context: mode -1 = SPRINGCRAFT_BULGE_PERSISTENT or the size rule (default), 0 never,
 * 1 by size, 2 always (3: always and in the pair form k_bulge_pair, 4: always with one sweep per workgroup,
 * k_bulge_chase, a matrix on one XCD; 5: as 4 with the workgroups of a matrix on all XCDs -- fewer matrices than XCDs only;
 * 2 picks the form by size like the default).  give_up_after > 0: test hook, every workgroup of the chase raises the
 * time-out flag after that many tasks, which forces the take-over on the device (k_chase_finish; counted in
 * "chase_resumed", not in "chase_timeouts").  tests/test_two_stage_gpu.py */
int sc_dbg_set_chase(sc_ctx* ctx, int mode, int give_up_after);

/* Cooperative panel QR of this context (k_panel_coop: several workgroups of one launch own 256 rows of a panel each;
 * stage 1 of the two-stage tridiagonalisation, few matrices): min_rows = panels of at least that many rows take it
 * (>= 128; the default rule: 300 rows with fewer than four matrices, else above the single-workgroup kernels' 6144),
 * 0 = never, -1 = the default rule
 * (SPRINGCRAFT_QR_COOP / SPRINGCRAFT_QR_COOP_MIN).  Counters "panel_coop_launches" / "panel_coop_timeouts".
 * tests/test_two_stage_gpu.py */
int sc_dbg_set_panel_coop(sc_ctx* ctx, int min_rows);
/* Test hook of the cooperative panel kernel's take-over: from panel `panel` on (-1: never) the abort flags of all
 * matrices are raised before the launch, as after a wait that ran into its bound; k_panel_serial then factors that panel
 * and every later one, counted in "panel_coop_timeouts". */
int sc_dbg_set_panel_coop_fail(sc_ctx* ctx, int panel);

/* One-launch tridiagonalisation of ONE matrix whose rows stay in LDS (k_sytrd_resident, tridiag.hip; one-stage path,
 * the trailing matrix of order <= 2048): mode 0 = never, 1 / -1 = the default rule (SPRINGCRAFT_RESIDENT).  hook: tests,
 * 1 = the roll call of its workgroups fails (nothing stored yet: k_sytrd_takeover reduces the matrix), 2 + c = the
 * exchange of step c fails (the take-over restores the matrix from its other triangle and starts again); 0 = none.
 * workgroups: 0 = by size, else a power of two (SPRINGCRAFT_RESIDENT_WGS).  Every call with mode != 0 re-arms a context
 * that had given the kernel up (after a lost wait, or three failed roll calls).  Counters "resident_launches" /
 * "resident_takeovers" / "resident_rollcall_failures" / "resident_lost_waits".  tests/test_resident_gpu.py */
int sc_dbg_set_resident(sc_ctx* ctx, int mode, int hook, int workgroups);
/* Library built with -DRES_STAMPS (else returns 1): cycles workgroup 0 of k_sytrd_resident spent in the segments of a
 * step -- [0] publish + poll of the records, [1] w~.v and the next column, [2] its norm, the next reflector and its
 * stores, [3] the pass over the own rows, [5] the sums of y --, [7] = steps; summed since the last call.
 * tools/resident_check.py --stamps */
int sc_dbg_resident_stamps(unsigned long long* out8);

#ifdef __cplusplus
}
#endif
#endif
