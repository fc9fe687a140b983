// Driver of the batched dense symmetric eigensolver (replaces np.linalg.eigh at the reference's
// nma.py:61): mirror -> tridiagonalise (tridiag.hip) -> tridiagonal eigenproblem (stedc.hip, or Sturm
// bisection when only eigenvalues are wanted) -> back-transformation (backtransform.hip).
// Also owns the workspace layout: everything the stages need is carved out of ONE cached device
// allocation per context (sc_ctx::ws), sized by eigh_workspace_bytes().
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <vector>

#include "eigh_internal.h"

namespace {

// Tridiagonalisation panel width nb (SYR2K inner dimension K = 2 nb).  Measured on MI355X, N = 2000 C-alpha,
// 16 structures in flight (profiles/r01_panel_width.txt): nb = 64: SYR2K 32.2 TFLOP/s (41 % of the 78.6 TFLOP/s
// f64-MFMA peak), 57.8k modes/s; nb = 96: 36.5 TFLOP/s (46 %), 56.8k; nb = 128: 38.3 TFLOP/s (49 %), 55.4k
// (wider panels make the per-column kernels read more V / W).  Default 64; SPRINGCRAFT_NB = 32|64|96|128 overrides.
int panel_width() {
  static int nb = [] {
    const char* e = getenv("SPRINGCRAFT_NB");
    const int v = e ? atoi(e) : 64;
    return (v == 128 || v == 96 || v == 64 || v == 32) ? v : 64;
  }();
  return nb;
}
#define kNb (panel_width())

size_t tri_slab_doubles(int n, TriLayout* out) {
  TriLayout L{};
  L.n = n;
  L.nb = kNb;
  const long long nt = (n + 63) / 64 + 1;
  long long off = 0;
  auto take = [&](long long cnt) { long long o = off; off += (cnt + 7) / 8 * 8; return o; };
  L.vw = take((long long)n * 2 * kNb);
  L.wv = take((long long)n * 2 * kNb);
  L.xraw = take(n);
  L.ypart = take(nt * n);
  L.dpart = take(nt * 2 * kNb);
  L.npart = take(nt + 8);
  L.wvpart = take(nt + 8);
  L.d = take(n);
  L.e = take(n);
  L.tau = take(n);
  L.hscale = take(8);
  L.rctl = take(8);
  L.rrec = take(6 * ((n + 63) / 64 * 64));
  L.slab = off;
  if (out) *out = L;
  return (size_t)off;
}

// Two-stage tridiagonalisation (twostage.hip) instead of the one-stage panel algorithm.  The one-stage SYMV streams
// 4/3 n^3 bytes per matrix and is bandwidth-bound as soon as a few matrices are in flight; the two-stage path does its
// O(n^3) work in MFMA GEMMs but pays ~3 n short launches and twice the back-transformation flops.  Measured crossover
// on MI355X: batch * n^2 above ~ max(1.7e7, 5e3 n) (round 4: max(2e7, 1e4 n), round 2: 5e7 + 6.7e3 n, round 1: 1.2e8;
// profiles/r0*_two_stage_crossover.txt).
// sc_ctx_set_two_stage(ctx, 1 / 0) or SPRINGCRAFT_TWO_STAGE=1 / 0 force it on / off.
bool two_stage_for(const sc_ctx* ctx, int n, int batch) {
  static const int env_mode = [] {
    const char* e = getenv("SPRINGCRAFT_TWO_STAGE");
    return e ? atoi(e) : -1;
  }();
  const int mode = (ctx && ctx->two_stage >= 0) ? ctx->two_stage : env_mode;
  if (n < 4 * sb_band_width()) return false;
  if (mode == 0) return false;
  if (mode == 1) return true;
  // round 4 (tools/crossover.py, profiles/r04_two_stage_crossover.txt; panel QR in one workgroup, the persistent chase
  // with two workgroups per CU and the secular solver of round 3 moved it down again): n = 6000 from 2 matrices (a tie
  // there), 3000 from 4, 1500 from ~8, 1026 and 513 with large batches (513 x 256: 49 vs 63 ms); one matrix at a time
  // stays on the one-stage path up to n ~ 12000 (n = 6000: 192 vs 237 ms, 3000: 67 vs 82)
  // round 5 (profiles/r05_two_stage_crossover.txt; the panel QR of a few matrices by several workgroups of one launch,
  // k_panel_coop): one matrix ties at n = 4500 (124 vs 122 ms) and n = 6000 (191 vs 187), two-stage from there on (7500:
  // 310 vs 273, 9000: 441 vs 360, 12000: 856 vs 581); 2 x 4500: 157 vs 133, 8 x 1500: 43.5 vs 39.5
  // round 6 (profiles/r06_two_stage_crossover.txt): ONE matrix whose trailing 3072 columns k_sytrd_resident reduces in
  // one launch stays on the one-stage path up to n ~ 7000 (n = 3000: 32 vs 71 ms, 5100: 110 vs 142, 6000: 152 vs 178,
  // 6600: 196 vs 201, 7200: 233 vs 230, 7800: 285 vs 263)
  if (batch == 1 && resident_enabled(ctx)) return n > 7000;
  return n >= 512 && (double)batch * n * n >= std::max(1.7e7, 5.0e3 * n);
}

struct Plan {
  TriLayout TL;
  DcLayout DL;
  BtLayout BL;
  SbLayout SL;
  bool two = false;
  size_t off_tri = 0, off_dc = 0, off_bt = 0, off_qtmp = 0, off_u = 0, off_desc = 0, off_sb = 0, off_dia = 0, total = 0;
  int n_syr2k = 0, n_merge = 0, n_bt = 0;
  long long n_bt2 = 0;
};

Plan make_plan(const sc_ctx* ctx, int n, int batch, bool vectors) {
  Plan P;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  P.off_tri = take(tri_slab_doubles(n, &P.TL) * 8 * batch);
  const int npanels = (n + kNb - 1) / kNb;
  P.n_syr2k = npanels * batch;
  P.two = two_stage_for(ctx, n, batch);
  if (P.two) {
    P.off_sb = take(sb_slab_doubles(n, batch, &P.SL) * 8 * batch);
    P.off_dia = take(sizeof(int) * ((size_t)n / 64 + 8));
    P.n_syr2k = std::max(P.n_syr2k, sb_desc_count(n, batch));
  }
  if (vectors) {
    P.off_dc = take(dc_slab_doubles(n, &P.DL) * 8 * batch);
    P.off_bt = take(bt_slab_doubles(n, &P.BL) * 8 * batch);
    P.off_qtmp = take((size_t)n * n * 8 * batch);
    P.off_u = take((size_t)n * n * 8 * batch);
    P.n_merge = 2 * dc_max_nodes(n, P.DL.leaf_max) * batch;   // two half-GEMMs per merge
    P.n_bt = bt_desc_count(n, batch);
  }
  P.off_desc = take(sizeof(GemmDesc) * ((size_t)P.n_syr2k + P.n_merge + P.n_bt + (size_t)P.n_bt2 + 8));
  P.total = off;
  return P;
}

// ---- eigenvalues only: Sturm bisection, one thread per eigenvalue ------------------------------------------
__global__ __launch_bounds__(256) void k_sturm(const double* __restrict__ tri_all, TriLayout TL,
                                               double* __restrict__ w_all, long long stride_w) {
  const double* tri = tri_all + (size_t)blockIdx.y * TL.slab;
  const double* d = tri + TL.d;
  const double* e = tri + TL.e;
  const int n = TL.n;
  __shared__ double red[8];
  // Gershgorin bounds + pivmin (block-wide)
  double lo = 1e300, hi = -1e300, emax = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double el = i > 0 ? fabs(e[i - 1]) : 0.0, er = i < n - 1 ? fabs(e[i]) : 0.0;
    lo = fmin(lo, d[i] - el - er);
    hi = fmax(hi, d[i] + el + er);
    emax = fmax(emax, er);
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, off));
    hi = fmax(hi, __shfl_xor(hi, off));
    emax = fmax(emax, __shfl_xor(emax, off));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w] = lo; red[4 + w] = hi; }
  __syncthreads();
  lo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
  hi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = emax;
  __syncthreads();
  emax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  const double span = fmax(fabs(lo), fabs(hi));
  const double pivmin = fmax(2.2250738585072014e-308 * fmax(1.0, emax * emax), 1e-290);
  lo -= 2.0 * 2.2e-16 * span * n + 2.0 * pivmin;
  hi += 2.0 * 2.2e-16 * span * n + 2.0 * pivmin;

  const int k = blockIdx.x * blockDim.x + threadIdx.x;  // eigenvalue index (ascending)
  if (k >= n) return;
  double a = lo, b = hi;
  for (int it = 0; it < 120; ++it) {
    const double mid = 0.5 * (a + b);
    if (mid <= a || mid >= b) break;
    int cnt = 0;
    double q = d[0] - mid;
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < 0.0;
    for (int i = 1; i < n; ++i) {
      const double ee = e[i - 1];
      q = d[i] - mid - ee * ee / q;
      if (fabs(q) < pivmin) q = -pivmin;
      cnt += q < 0.0;
    }
    if (cnt > k) b = mid; else a = mid;
  }
  w_all[(size_t)blockIdx.y * stride_w + k] = 0.5 * (a + b);
}

}  // namespace

int sturm_bisect_batched(sc_ctx* ctx, int n, int batch, const double* d_tri_ws, const TriLayout& TL,
                         double* d_w, long long stride_w) {
  hipLaunchKernelGGL(k_sturm, dim3((unsigned)((n + 255) / 256), (unsigned)batch), dim3(256), 0, ctx->stream,
                     d_tri_ws, TL, d_w, stride_w);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

size_t eigh_workspace_bytes(int64_t n, int64_t batch, bool want_vectors) {
  return make_plan(nullptr, (int)n, (int)batch, want_vectors).total;
}

// Once work has been forked onto the context's second stream, EVERY way out of the solve must make the main stream wait
// for it: the kernels queued there read and write d_a, the band workspace and the back-transformation workspace, which
// the caller may reuse or free as soon as the call has returned, and the next solve on the context must be ordered
// after them (the band reduction's side streams got the same guard in round 3).
struct AuxJoinGuard {
  sc_ctx* ctx;
  hipStream_t main;
  bool forked = false, joined = false;
  ~AuxJoinGuard() {
    ctx->stream = main;   // (a failing call between the two assignments below must not leave the context on the aux stream)
    if (forked && !joined) {
      (void)hipEventRecord(ctx->aux_join, ctx->aux_stream);
      (void)hipStreamWaitEvent(main, ctx->aux_join, 0);
    }
  }
};

int eigh_batched_async(sc_ctx* ctx, double* d_a, int64_t n64, int64_t batch64, double* d_w, double* d_v) {
  if (n64 > 46000) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "matrix order %lld too large", (long long)n64);
  const int n = (int)n64, batch = (int)batch64;
  const bool vectors = d_v != nullptr;
  hipStream_t st = ctx->stream;
  const Plan P = make_plan(ctx, n, batch, vectors);
  SC_TRY(sc_reserve_ws(ctx, P.total));
  char* base = (char*)ctx->ws;
  double* tri_ws = (double*)(base + P.off_tri);
  GemmDesc* descs = (GemmDesc*)(base + P.off_desc);
  const long long stride_a = (long long)n * n;

  // SYR2K descriptors: one per (panel, matrix)
  const int npanels = (n + kNb - 1) / kNb;
  std::vector<GemmDesc> h((size_t)npanels * batch);
  for (int p = 0; p < npanels; ++p) {
    const int pend = std::min((p + 1) * kNb, n);
    for (int b = 0; b < batch; ++b) {
      double* ws = tri_ws + (size_t)b * P.TL.slab;
      GemmDesc D{};
      D.a = ws + P.TL.vw + pend; D.sa_i = 1; D.sa_k = n;
      D.b = ws + P.TL.wv + pend; D.sb_k = n; D.sb_j = 1;
      D.c = d_a + (size_t)b * stride_a + (size_t)pend * n + pend; D.ldc = n;
      D.m = n - pend; D.n = n - pend; D.k = 2 * kNb;
      D.alpha = -1.0; D.beta = 1.0;
      D.lower_only = 1;
      h[(size_t)p * batch + b] = D;
    }
  }
  SC_TRY(sc_stage_upload(ctx, descs, h.data(), h.size() * sizeof(GemmDesc)));

  ScopedEvents<4> ev;
  float ms_bt2 = 0.f;
  const bool prof = ctx->profiling;
  if (prof) {
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    SC_HIP(ctx, hipEventRecord(ev[0], st));
  }
  float ms_symv = 0.f, ms_syr2k = 0.f;
  double* sb_ws = (double*)(base + P.off_sb);
  std::unique_ptr<PhaseTimer> t_tf;   // T factors of the stage-2 diamonds (second stream)
  if (prof) ctx->phases.clear();
  SC_TRY(prepare_matrix_batched(ctx, d_a, stride_a, n, batch, tri_ws, P.TL));
  if (P.two) {
    // [3] / [4] then carry the stage-1 / stage-2 times
    SC_TRY(sytrd_2stage_batched(ctx, d_a, stride_a, n, batch, tri_ws, P.TL, sb_ws, P.SL, (int*)(base + P.off_dia),
                                descs, &ms_symv, &ms_syr2k));
  } else {
    SC_TRY(tridiag_batched(ctx, d_a, stride_a, n, batch, tri_ws, P.TL, descs, &ms_symv, &ms_syr2k));
  }
  if (prof) SC_HIP(ctx, hipEventRecord(ev[1], st));

  if (!vectors) {
    SC_TRY(sturm_bisect_batched(ctx, n, batch, tri_ws, P.TL, d_w, n));
    SC_TRY(unscale_values_batched(ctx, d_w, n, n, batch, tri_ws, P.TL));
    if (prof) SC_HIP(ctx, hipEventRecord(ev[2], st));
  } else {
    static const bool no_aux = getenv("SPRINGCRAFT_NO_AUX") != nullptr;
    double* dc_ws = (double*)(base + P.off_dc);
    double* bt_ws = (double*)(base + P.off_bt);
    double* q_tmp = (double*)(base + P.off_qtmp);
    double* u = (double*)(base + P.off_u);
    GemmDesc* bt_descs = descs + P.n_syr2k + P.n_merge;
    const int bt_off = P.two ? sb_band_width() : 1;
    // (one structure on the one-stage path with its launches per column: the fork / join costs more than the overlap
    // brings, 30.0 -> 31.3 ms at N = 512)
    // (round 6: one structure whose tridiagonalisation is the single launch of k_sytrd_resident forks as well -- the T
    // factors' 0.6 ms beside a D&C of 2 ms: N = 100 2.46 -> 1.88 ms, N = 512 9.23 -> 8.69; SPRINGCRAFT_AUX_SINGLE = 0 / 1)
    static const int aux_single = [] { const char* e = getenv("SPRINGCRAFT_AUX_SINGLE"); return e ? atoi(e) : -1; }();
    const bool single_fork = aux_single >= 0 ? aux_single != 0 : (batch == 1 && resident_enabled(ctx));
    const bool use_aux = !no_aux && (P.two || batch >= 4 || single_fork);
    AuxJoinGuard aux{ctx, st};
    if (P.two && no_aux) {
      t_tf.reset(new PhaseTimer(ctx, "dia_tfactor", st));
      t_tf->start();
      SC_TRY(bt2_prepare(ctx, n, batch, sb_ws, P.SL, st));
      t_tf->stop();
    } else if (use_aux) {
      // what the back-transformations need besides Z does not depend on the tridiagonal eigenproblem: the diamonds' T
      // factors (two-stage) and the cleaned reflectors, Gram products and T factors of the Q1 blocks run on a second
      // stream alongside the D&C (their descriptors are uploaded on the main stream before the fork)
      SC_TRY(backtransform_batched(ctx, d_a, stride_a, n, batch, tri_ws, P.TL, bt_ws, P.BL, d_v, stride_a, n, q_tmp,
                                   bt_descs, bt_off, /*phase=*/1));
      SC_TRY(sc_aux_stream(ctx));
      SC_HIP(ctx, hipEventRecord(ctx->aux_fork, st));
      SC_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
      aux.forked = true;
      if (P.two) {
        t_tf.reset(new PhaseTimer(ctx, "dia_tfactor", ctx->aux_stream));
        t_tf->start();
        SC_TRY(bt2_prepare(ctx, n, batch, sb_ws, P.SL, ctx->aux_stream));
        t_tf->stop();
      }
      {
        ctx->stream = ctx->aux_stream;   // (launch_gemm_f64 launches on the context's stream)
        const int rc_prep = backtransform_batched(ctx, d_a, stride_a, n, batch, tri_ws, P.TL, bt_ws, P.BL, d_v, stride_a,
                                                  n, q_tmp, bt_descs, bt_off, /*phase=*/2);
        ctx->stream = st;
        if (rc_prep != SC_OK) return rc_prep;     // (joined by the guard)
        SC_HIP(ctx, hipEventRecord(ctx->aux_join, ctx->aux_stream));
      }
    }
    SC_TRY(stedc_batched(ctx, n, batch, tri_ws, P.TL, dc_ws, P.DL, d_w, n, d_v, q_tmp, u, stride_a, descs + P.n_syr2k));
    SC_TRY(unscale_values_batched(ctx, d_w, n, n, batch, tri_ws, P.TL));
    if (prof) SC_HIP(ctx, hipEventRecord(ev[2], st));
    if (use_aux) {
      SC_HIP(ctx, hipStreamWaitEvent(st, ctx->aux_join, 0));
      aux.joined = true;
    }
    if (P.two) SC_TRY(bt2_batched(ctx, n, batch, sb_ws, P.SL, (const int*)(base + P.off_dia), d_v, stride_a, n, &ms_bt2));
    SC_TRY(backtransform_batched(ctx, d_a, stride_a, n, batch, tri_ws, P.TL, bt_ws, P.BL, d_v, stride_a, n, q_tmp,
                                 bt_descs, bt_off, /*phase=*/use_aux ? 3 : 0));
  }
  if (prof) {
    SC_HIP(ctx, hipEventRecord(ev[3], st));
    SC_HIP(ctx, hipEventSynchronize(ev[3]));
    float t01 = 0, t12 = 0, t23 = 0;
    SC_HIP(ctx, hipEventElapsedTime(&t01, ev[0], ev[1]));
    SC_HIP(ctx, hipEventElapsedTime(&t12, ev[1], ev[2]));
    SC_HIP(ctx, hipEventElapsedTime(&t23, ev[2], ev[3]));
    ctx->last_timings[0] = t01;
    ctx->last_timings[1] = t12;
    ctx->last_timings[2] = t23;
    ctx->last_timings[3] = ms_symv;
    ctx->last_timings[4] = ms_syr2k;
    // two-stage path: [3] = stage 1 (band reduction), [4] = stage 2 (bulge chasing), [5] = the fused kernel that
    // back-transforms the stage-2 reflectors (k_bt2_fused); [5] = 0 marks the one-stage path
    ctx->last_timings[5] = 0.0;
    if (P.two) ctx->last_timings[5] = vectors ? ms_bt2 : 1e-9;
    if (t_tf) t_tf->finish();
  }
  return sc_stage_end(ctx);   // (no synchronisation: the descriptor tables went through the context's pinned arena)
}

// Synchronising form for the host-pointer entry points: errors of THIS solve (non-finite input, QL failure) are
// returned by it, as np.linalg.eigh raises them at nma.py:61.
int eigh_batched(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, double* d_w, double* d_v) {
  SC_TRY(eigh_batched_async(ctx, d_a, n, batch, d_w, d_v));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return sc_deferred_status(ctx);
}


// ---- partial spectrum --------------------------------------------------------------------------------------
int eigh_range_batched_async(sc_ctx* ctx, double* d_a, int64_t n64, int64_t batch64, int64_t il64, int64_t iu64,
                             double* d_w, double* d_v) {
  if (n64 > 46000) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "matrix order %lld too large", (long long)n64);
  const int n = (int)n64, batch = (int)batch64, il = (int)il64, iu = (int)iu64;
  if (il < 0 || iu < il || iu >= n)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad eigenvalue index range [%d, %d] for order %d", il, iu, n);
  const int m = iu - il + 1;
  hipStream_t st = ctx->stream;
  const long long stride_a = (long long)n * n;

  // workspace: tri slab | [sb slab | diamond offsets] | stein | bt slab | VT (n x n) | descriptors
  TriLayout TL;
  BtLayout BL;
  SbLayout SL;
  const bool two = two_stage_for(ctx, n, batch);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t off_tri = take(tri_slab_doubles(n, &TL) * 8 * batch);
  const int npanels = (n + kNb - 1) / kNb;
  size_t off_stein = 0, off_bt = 0, off_vt = 0, off_sb = 0, off_dia = 0;
  int n_bt = 0;
  int n_tri_desc = npanels * batch;
  if (two) {
    off_sb = take(sb_slab_doubles(n, batch, &SL) * 8 * batch);
    off_dia = take(sizeof(int) * ((size_t)n / 64 + 8));
    n_tri_desc = std::max(n_tri_desc, sb_desc_count(n, batch));
  }
  if (d_v) {
    off_stein = take(stein_workspace_doubles(n, m) * 8 * batch);
    off_bt = take(bt_slab_doubles(n, &BL) * 8 * batch);
    off_vt = take((size_t)n * n * 8 * batch);
    n_bt = bt_desc_count(n, batch);
  }
  const size_t n_desc = (size_t)n_tri_desc + n_bt + 2 * (size_t)batch + 8;
  const size_t off_desc = take(sizeof(GemmDesc) * n_desc);
  SC_TRY(sc_reserve_ws(ctx, off));
  char* base = (char*)ctx->ws;
  double* tri_ws = (double*)(base + off_tri);
  double* sb_ws = (double*)(base + off_sb);
  GemmDesc* descs = (GemmDesc*)(base + off_desc);

  std::vector<GemmDesc> h((size_t)npanels * batch);
  // profiled solve: [0] tridiagonalisation, [1] bisection + inverse iteration, [2] back-transformation, and as in
  // eigh_batched [3] / [4] = band reduction / bulge chasing (two-stage) or SYMV / SYR2K sums, [5] = k_bt2_apply
  ScopedEvents<4> ev;
  const bool prof = ctx->profiling;
  float ms_a = 0.f, ms_b = 0.f, ms_bt2 = 0.f;
  if (prof) {
    ctx->phases.clear();
    for (auto& e : ev) SC_HIP(ctx, hipEventCreate(&e));
    SC_HIP(ctx, hipEventRecord(ev[0], st));
  }
  SC_TRY(prepare_matrix_batched(ctx, d_a, stride_a, n, batch, tri_ws, TL));
  if (two) {
    SC_TRY(sytrd_2stage_batched(ctx, d_a, stride_a, n, batch, tri_ws, TL, sb_ws, SL, (int*)(base + off_dia), descs,
                                &ms_a, &ms_b));
  } else {
    for (int p = 0; p < npanels; ++p) {
      const int pend = std::min((p + 1) * kNb, n);
      for (int b = 0; b < batch; ++b) {
        double* ws = tri_ws + (size_t)b * TL.slab;
        GemmDesc D{};
        D.a = ws + TL.vw + pend; D.sa_i = 1; D.sa_k = n;
        D.b = ws + TL.wv + pend; D.sb_k = n; D.sb_j = 1;
        D.c = d_a + (size_t)b * stride_a + (size_t)pend * n + pend; D.ldc = n;
        D.m = n - pend; D.n = n - pend; D.k = 2 * kNb;
        D.alpha = -1.0; D.beta = 1.0;
        D.lower_only = 1;
        h[(size_t)p * batch + b] = D;
      }
    }
    SC_TRY(sc_stage_upload(ctx, descs, h.data(), h.size() * sizeof(GemmDesc)));
    SC_TRY(tridiag_batched(ctx, d_a, stride_a, n, batch, tri_ws, TL, descs, &ms_a, &ms_b));
  }
  if (prof) SC_HIP(ctx, hipEventRecord(ev[1], st));
  GemmDesc* d2 = descs + (size_t)n_tri_desc;
  if (!d_v) {
    SC_TRY(stein_batched(ctx, n, batch, tri_ws, TL, il, iu, d_w, m, nullptr, 0, nullptr, d2));
    SC_TRY(unscale_values_batched(ctx, d_w, m, m, batch, tri_ws, TL));
    if (prof) SC_HIP(ctx, hipEventRecord(ev[2], st));
  } else {
    const long long stride_x = (long long)n * m;
    SC_TRY(stein_batched(ctx, n, batch, tri_ws, TL, il, iu, d_w, m, d_v, stride_x, (double*)(base + off_stein), d2));
    SC_TRY(unscale_values_batched(ctx, d_w, m, m, batch, tri_ws, TL));
    if (prof) SC_HIP(ctx, hipEventRecord(ev[2], st));
    if (two) {
      PhaseTimer t_tf(ctx, "dia_tfactor", st);
      t_tf.start();
      SC_TRY(bt2_prepare(ctx, n, batch, sb_ws, SL, st));
      t_tf.stop();
      SC_TRY(bt2_batched(ctx, n, batch, sb_ws, SL, (const int*)(base + off_dia), d_v, stride_x, m, &ms_bt2));
      t_tf.finish();
    }
    // the back-transformation indexes its scratch with the matrix stride: VT lives in an n x n buffer per matrix
    SC_TRY(backtransform_batched(ctx, d_a, stride_a, n, batch, tri_ws, TL, (double*)(base + off_bt), BL, d_v,
                                 stride_x, m, (double*)(base + off_vt), d2 + 2 * batch, two ? sb_band_width() : 1));
  }
  if (prof) {
    SC_HIP(ctx, hipEventRecord(ev[3], st));
    SC_HIP(ctx, hipEventSynchronize(ev[3]));
    float t01 = 0, t12 = 0, t23 = 0;
    SC_HIP(ctx, hipEventElapsedTime(&t01, ev[0], ev[1]));
    SC_HIP(ctx, hipEventElapsedTime(&t12, ev[1], ev[2]));
    SC_HIP(ctx, hipEventElapsedTime(&t23, ev[2], ev[3]));
    ctx->last_timings[0] = t01;
    ctx->last_timings[1] = t12;
    ctx->last_timings[2] = t23;
    ctx->last_timings[3] = ms_a;
    ctx->last_timings[4] = ms_b;
    ctx->last_timings[5] = two ? (d_v ? ms_bt2 : 1e-9) : 0.0;
  }
  return sc_stage_end(ctx);
}

int eigh_range_batched(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, int64_t il, int64_t iu, double* d_w,
                       double* d_v) {
  SC_TRY(eigh_range_batched_async(ctx, d_a, n, batch, il, iu, d_w, d_v));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return sc_deferred_status(ctx);
}


// ---- Hermitian pseudo-inverse from the eigenpairs (F1: covariance) ------------------------------------------
// np.linalg.pinv(M, hermitian=True, rcond) as used at anm.py:115,135 / gnm.py:108,128:
//   M = U diag(w) U^T;  pinv = (U * s) U^T  with  s_i = 1/w_i if |w_i| > rcond * max|w| else 0.
namespace {
__global__ void k_pinv_scale(const double* __restrict__ q, const double* __restrict__ w, int n, double rcond,
                             double* __restrict__ qs) {
  __shared__ double s_cut;
  if (threadIdx.x == 0) {
    double mx = 0.0;
    for (int i = 0; i < n; ++i) mx = fmax(mx, fabs(w[i]));   // w is ascending: max |w| is at one of the ends
    s_cut = rcond * mx;
  }
  __syncthreads();
  const int col = blockIdx.x;
  const double wi = w[col];
  const double s = fabs(wi) > s_cut ? 1.0 / wi : 0.0;
  for (int r = threadIdx.x; r < n; r += blockDim.x) qs[(size_t)col * n + r] = q[(size_t)col * n + r] * s;
}
}  // namespace

int pinvh_device(sc_ctx* ctx, double* d_a, int64_t n64, double rcond, double* d_out) {
  const int n = (int)n64;
  hipStream_t st = ctx->stream;
  // buffers: Q (n^2) | QS (n^2) | w (n) | desc from a cached grow-only allocation of the context (its other cached
  // buffers are all in use by the caller and by eigh_batched while this runs)
  const size_t nn = align_up(sizeof(double) * (size_t)n * n, 256), nw = align_up(sizeof(double) * (size_t)n, 256);
  SC_TRY(sc_reserve_pinv(ctx, 2 * nn + nw + sizeof(GemmDesc)));
  char* d_base = reinterpret_cast<char*>(ctx->pinv_ws);
  double* d_q = reinterpret_cast<double*>(d_base);
  double* d_qs = reinterpret_cast<double*>(d_base + nn);
  double* d_w = reinterpret_cast<double*>(d_base + 2 * nn);
  GemmDesc* d_desc = reinterpret_cast<GemmDesc*>(d_base + 2 * nn + nw);
  int rc = eigh_batched(ctx, d_a, n, 1, d_w, d_q);
  if (rc == SC_OK) {
    hipLaunchKernelGGL(k_pinv_scale, dim3((unsigned)n), dim3(256), 0, st, d_q, d_w, n, rcond, d_qs);
    GemmDesc D{};
    D.a = d_qs; D.sa_i = 1; D.sa_k = n;       // (U * s)
    D.b = d_q; D.sb_k = n; D.sb_j = 1;        // U^T: B(k,j) = U[j, k]
    D.c = d_out; D.ldc = n; D.m = n; D.n = n; D.k = n;
    D.alpha = 1.0; D.beta = 0.0;
    rc = sc_stage_upload(ctx, d_desc, &D, sizeof(D));
    if (rc == SC_OK) rc = launch_gemm_f64(ctx, d_desc, 1, n, n, kGemmTile, 1, false, false, kGemmAmBn);
    if (rc == SC_OK) rc = sc_stage_end(ctx);
  }
  return rc;
}
