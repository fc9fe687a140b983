// Internal declarations shared by the translation units of libspringcraft_hip.so.
// Nothing here is part of the C ABI (see include/springcraft_hip.h for that).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/springcraft_hip.h"

// ---- context -----------------------------------------------------------------------------
struct sc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // second stream + events for work that is independent of what runs on `stream` (created on first use)
  hipStream_t aux_stream = nullptr;
  // further side streams (bulge chasing runs parts of the batch side by side) with one join event each
  std::vector<hipStream_t> side_streams;
  std::vector<hipEvent_t> side_joins;
  hipEvent_t aux_fork = nullptr, aux_join = nullptr;
  std::string err;
  int num_cus = 0;

  // cached device workspace for the eigensolver (grown on demand, never shrunk)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // small cached scratch for the assembly entry points (patch tables, counts, ...)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // node tables / oversize-merge copies of the divide-and-conquer solver (lives across a whole eigensolve, so it
  // cannot share `scratch`, which the assembly entry points hold while they call the solver)
  void* dc_aux = nullptr;
  size_t dc_aux_bytes = 0;
  // eigenvector / scaled-eigenvector buffers of pinvh_device (covariance properties), alive across its eigensolve
  void* pinv_ws = nullptr;
  size_t pinv_ws_bytes = 0;

  // host -> device uploads of descriptor tables without a stream synchronisation: the tables are copied into this
  // pinned arena first (sc_stage_upload) and the arena is recycled once the event recorded at the end of the solve
  // (sc_stage_end) has passed
  // (two arenas taking turns, so that the host may enqueue solve k + 1 while solve k still runs)
  char* h_stage[2] = {nullptr, nullptr};
  size_t h_stage_bytes[2] = {0, 0}, h_stage_off[2] = {0, 0};
  hipEvent_t h_stage_done[2] = {nullptr, nullptr};
  bool h_stage_pending[2] = {false, false};
  int h_stage_cur = 0;
  // deferred status of the device-pointer eigensolver entries (which do not synchronise): [0] = 1 + index of a matrix
  // with a NaN / Inf entry, [1] = the tridiagonal QL iteration failed.  Read and cleared by sc_deferred_status.
  // Round 6, events a solve no longer looks at from the host (kSpStatusWords words; sc_collect_events adds them to the
  // counters below at the next synchronising call): [2] persistent chases that k_chase_finish had to finish, [3] of
  // those: after a time-out (not the test hook), [4] sweeps the persistent kernels finished, [5] panels factored by
  // k_panel_serial after k_panel_coop gave up, [6] chases left incomplete without a raised flag (an XCD that owns
  // matrices received no workgroup), [7] tridiagonalisations k_sytrd_takeover did after k_sytrd_resident gave up ([8] of
  // those after an aborted roll call, [9] after a wait lost in mid-run, [10] where).
  unsigned long long* d_status = nullptr;

  int two_stage = -1;   // eigensolver path: -1 automatic, 0 one-stage, 1 two-stage tridiagonalisation
  // persistent bulge chase (twostage.hip): chase_ok = 0 once a chase of this context ran into its time-out (never
  // expected; the per-wavefront launches take over from then on); chase_mode / chase_give_up are set through the debug
  // entry sc_dbg_set_chase (-1: SPRINGCRAFT_BULGE_PERSISTENT or the size rule; 0 / 1 / 2 as that variable).
  int chase_ok = -1;
  int chase_mode = -1, chase_give_up = 0;
  int chase_form = -1;   // debug entry: 1 = pair form, 0 = one sweep per workgroup, 2 = that with a matrix' workgroups on all XCDs, -1 = by size
  // XCDs of this device as a probe launch saw them (distinct XCC_ID values; 0 = not probed yet), and whether
  // k_bulge_pair's dynamic LDS size has been raised on THIS context's device (-1 = not tried, 0 = refused, 1 = set):
  // the attribute is per device, a process may own contexts on several
  int nxcd = 0;
  int pair_attr = -1;
  long long cnt_pair_fallbacks = 0;   // pair launches that were refused and re-issued as k_bulge_chase
  // k_gemm3 (gemm3.hip): whether its dynamic LDS size has been raised on this context's device (-1 not tried, 0 refused,
  // 1 set), and the launches it took
  int gemm3_attr = -1;
  // k_symm3 (symm3.hip): a page of zeros its loaders point pieces at that must not count, and the launches it took
  void* d_zeros = nullptr;
  long long cnt_symm3_launches = 0;
  bool gemm3_side_by_side = false;   // the caller runs parts of the batch on several streams (band reduction): see gemm3_would_take
  long long cnt_gemm3_launches = 0;
  // k_panel_coop (twostage.hip): dynamic LDS attribute of this device (-1 not tried, 0 refused, 1 set); coop_ok = 0
  // once a wait between its workgroups timed out (the context then keeps to the chunked panel launches)
  int coop_attr = -1, coop_ok = -1;
  int coop_min_rows = -1;   // debug entry sc_dbg_set_panel_coop: rows from which a panel takes it (0 never, -1 default rule)
  int coop_fail_panel = -1; // debug entry sc_dbg_set_panel_coop_fail: panel from which the abort flags are raised (test hook)
  // k_sytrd_resident (tridiag.hip): resident_ok = 0 once a wait was lost or three roll calls failed (the launches per column from
  // then on); resident_mode / _hook / _wgs through the debug entry sc_dbg_set_resident (-1 / 0 / 0: the default rule)
  int resident_ok = -1, resident_mode = -1, resident_hook = 0, resident_wgs = 0;
  long long cnt_resident_launches = 0, cnt_resident_takeovers = 0, cnt_resident_rollcalls = 0, cnt_resident_lost = 0;
  long long resident_lost_at = -1;   // (step << 32 | workgroup) of the most recent lost wait / arrivals at the last failed roll call
  int resident_strikes = 0;          // failed roll calls since the kernel was (re-)armed: the context gives it up at the third
  int* last_chase_ctl = nullptr;   // control block of the most recent persistent chase (in dc_aux), read by sc_collect_events
  long long cnt_coop_launches = 0, cnt_coop_timeouts = 0;
  // event counters since the context was created (sc_ctx_get_counter)
  long long cnt_chase_launches = 0, cnt_chase_timeouts = 0, cnt_chase_incomplete = 0, cnt_chase_resumed = 0,
            cnt_chase_sweeps = 0, cnt_stepwise_chases = 0, cnt_pair_launches = 0;
  int chase_tickets[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // workgroups per XCD of the most recent chase launch
  int chase_wait[3] = {-1, -1, -1};                  // (matrix, sweep, task) of the wait that timed out last
  bool profiling = false;
  double last_timings[6] = {0, 0, 0, 0, 0, 0};
  // named kernel-group durations of the most recent profiled eigensolve (sc_last_eigh_phase_ms)
  std::vector<std::pair<std::string, double>> phases;
};

// Sum of the device time between start() / stop() pairs on one stream (HIP events; profiling runs only).  The events are
// resolved in finish(), which synchronises on the last one; destroyed with the object on every return path.
class PhaseTimer {
 public:
  PhaseTimer(sc_ctx* ctx, const char* name, hipStream_t st) : ctx_(ctx), name_(name), st_(st), on_(ctx->profiling) {}
  ~PhaseTimer() { for (hipEvent_t e : ev_) (void)hipEventDestroy(e); }
  void start() { mark(); }
  void stop() { mark(); }
  // stores the sum under `name` in ctx->phases; call after everything timed has been enqueued
  void finish() {
    if (!on_ || ev_.size() < 2) return;
    double total = 0.0;
    if (hipEventSynchronize(ev_.back()) != hipSuccess) return;
    for (size_t i = 0; i + 1 < ev_.size(); i += 2) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev_[i], ev_[i + 1]) == hipSuccess) total += ms;
    }
    for (auto& p : ctx_->phases)
      if (p.first == name_) { p.second = total; return; }
    ctx_->phases.emplace_back(name_, total);
  }
 private:
  void mark() {
    if (!on_) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { on_ = false; return; }
    (void)hipEventRecord(e, st_);
    ev_.push_back(e);
  }
  sc_ctx* ctx_;
  std::string name_;
  hipStream_t st_;
  bool on_;
  std::vector<hipEvent_t> ev_;
};

// N timing events that are destroyed on every return path (profiling runs only create them).
template <int N>
struct ScopedEvents {
  hipEvent_t e[N];
  ScopedEvents() { for (auto& x : e) x = nullptr; }
  ~ScopedEvents() { for (auto& x : e) if (x) (void)hipEventDestroy(x); }
  ScopedEvents(const ScopedEvents&) = delete;
  ScopedEvents& operator=(const ScopedEvents&) = delete;
  hipEvent_t& operator[](int i) { return e[i]; }
  hipEvent_t* begin() { return e; }
  hipEvent_t* end() { return e + N; }
};

int sc_set_error(sc_ctx* ctx, int code, const char* fmt, ...);

#define SC_HIP(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess)                                                             \
      return sc_set_error((ctx), e__ == hipErrorOutOfMemory ? SC_ERR_NOMEM : SC_ERR_HIP, \
                          "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__),      \
                          __FILE__, __LINE__);                                         \
  } while (0)

#define SC_TRY(expr)            \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != SC_OK) return rc__; \
  } while (0)

// Grow-only cached allocations.
int sc_reserve_ws(sc_ctx* ctx, size_t bytes);
int sc_aux_stream(sc_ctx* ctx);   // creates aux_stream / aux_fork / aux_join if needed
int sc_side_streams(sc_ctx* ctx, int count);   // makes sure side_streams / side_joins hold `count` entries
int sc_reserve_scratch(sc_ctx* ctx, size_t bytes);
int sc_reserve_dc_aux(sc_ctx* ctx, size_t bytes);
int sc_reserve_pinv(sc_ctx* ctx, size_t bytes);
// d_dst <- bytes at h_src, enqueued on ctx->stream; h_src may be released as soon as the call returns
int sc_stage_upload(sc_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
// end of a solve that used sc_stage_upload: the arena may be reused when everything enqueued so far has run
int sc_stage_end(sc_ctx* ctx);
// after a synchronisation of ctx->stream: SC_ERR_NOCONV (and the flags cleared) if a solve since the last call met
// non-finite input or a QL failure, else SC_OK
constexpr int kSpStatusWords = 16;
int sc_deferred_status(sc_ctx* ctx);
// adds the device-side event words to the context's counters (the context's stream must be idle)
int sc_collect_events(sc_ctx* ctx);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- assembly (assembly.hip) ----------------------------------------------------------------
// Device-side patch tables (all device pointers; null when there are no patches).
struct PatchDev {
  const uint8_t* shut;     // (N) 1 = every contact of this atom is switched off
  const int32_t* row_ptr;  // (N+1) CSR over override entries, both directions stored
  const int32_t* col;      // (nnz)
  const int8_t* flag;      // (nnz) 0 = force off, 1 = force on
  const double* gam;       // (nnz) override force constant, NaN = use the base force field
  int mask_gamma;          // PatchedForceField: base gamma is 0 beyond the base cutoff
};

int launch_kirchhoff(sc_ctx* ctx, const double* d_coord, int64_t n, int64_t batch,
                     const sc_ff_desc& ff, const PatchDev* patch, const double* d_w,
                     double* d_k, int64_t* d_counts);
int launch_hessian(sc_ctx* ctx, const double* d_coord, int64_t n, int64_t batch,
                   const sc_ff_desc& ff, const PatchDev* patch, const double* d_w, double* d_h);
int launch_contact_counts(sc_ctx* ctx, const double* d_coord, int64_t n, const sc_ff_desc& ff,
                          const PatchDev* patch, int64_t* d_counts);
// d_offsets: exclusive scan of counts (N+1 entries, last = k)
int launch_pair_fill(sc_ctx* ctx, const double* d_coord, int64_t n, const sc_ff_desc& ff,
                     const PatchDev* patch, const int64_t* d_offsets, int64_t* d_pairs,
                     double* d_sqdist);
int launch_exclusive_scan_i64(sc_ctx* ctx, const int64_t* d_in, int64_t n, int64_t* d_out);
int launch_kirchhoff_from_pairs(sc_ctx* ctx, int64_t n, const int64_t* d_pairs, int64_t k,
                                const double* d_gamma, double* d_k);
int launch_hessian_from_pairs(sc_ctx* ctx, const double* d_coord, int64_t n,
                              const int64_t* d_pairs, int64_t k, const double* d_gamma,
                              double* d_h);
// Ragged / decorated batches (sc_batch_plan, api.hip): records of one structure each, opaque outside assembly.hip.
size_t asm_item_bytes();
// ff_dev: descriptor whose .tab (if any) holds DEVICE pointers; patch: device tables (empty tables when unpatched)
void asm_item_fill(void* dst, long long atom_off, int n, int ld, const sc_ff_desc& ff_dev, const PatchDev& patch);
// d_items: `count` records in device memory.  Writes (count, ld, ld) matrices: the structure's matrix in the leading
// dim * n rows / columns of its slot, the rest padded (see k_pad_fill).  d_bound_bits: count uint64 of scratch.
int launch_assemble_items(sc_ctx* ctx, int dim, const void* d_items, int64_t count, int max_atoms, bool any_patch,
                          bool any_pad, const double* d_coord, const double* d_w, double* d_matrix,
                          unsigned long long* d_bound_bits);
// Host-callback force fields for a whole plan (sc_batch_plan_contacts / _pairs / _fill_from_pairs_f64): contact counts of
// all atoms back to back; ordered pair lists (local indices) behind the exclusive scan of those counts; all slots from
// pairs + gamma.
int launch_items_counts(sc_ctx* ctx, const void* d_items, int64_t count, int max_atoms, bool any_patch,
                        const double* d_coord, int64_t* d_counts);
int launch_items_pair_fill(sc_ctx* ctx, const void* d_items, int64_t count, int max_atoms, bool any_patch,
                           const double* d_coord, const int64_t* d_offsets, int64_t* d_pairs, double* d_sqdist);
int launch_items_from_pairs(sc_ctx* ctx, int dim, const void* d_items, int64_t count, int max_atoms, int64_t order,
                            bool any_pad, const double* d_coord, const int64_t* d_pair_off, int64_t k_total,
                            const int64_t* d_pairs, const double* d_gamma, const double* d_w, double* d_matrix,
                            unsigned long long* d_bound_bits);
// In-place transpose-free symmetric "row-major == column-major" note: the eigensolver reads the
// LOWER triangle in column-major order, i.e. the UPPER triangle of the row-major matrix the
// assembly writes; the matrices are symmetric so both views agree.

// ---- eigensolver (eigh.hip and friends) ----------------------------------------------------
// d_a: (batch, n, n) symmetric (destroyed), d_w: (batch, n), d_v: nullptr or (batch, n, n) rows = modes.
// eigh_batched synchronises the stream and returns this solve's errors (host-pointer entry points); the _async form only
// enqueues: non-finite input / a QL failure then surface through sc_deferred_status at the next synchronising call.
int eigh_batched(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, double* d_w, double* d_v);
int eigh_batched_async(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, double* d_w, double* d_v);
size_t eigh_workspace_bytes(int64_t n, int64_t batch, bool want_vectors);
// Partial spectrum: eigenvalues il..iu (0-based, inclusive): d_w (batch, m), d_v nullptr or (batch, m, n).
int eigh_range_batched(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, int64_t il, int64_t iu,
                       double* d_w, double* d_v);
int eigh_range_batched_async(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, int64_t il, int64_t iu,
                             double* d_w, double* d_v);
// Hermitian pseudo-inverse of the (n,n) matrix d_a (destroyed) into d_out, numpy.linalg.pinv(hermitian=True) rule.
int pinvh_device(sc_ctx* ctx, double* d_a, int64_t n, double rcond, double* d_out);

// ---- mode-subset consumers on device-resident eigenpairs (consumers.hip) ------------------------------
// what: 0 = msf, 1 = dcc, 2 = prs
size_t modes_scratch_bytes(int64_t n, int dim, int64_t nsel, int what);
int modes_msf_device(sc_ctx* ctx, const double* d_v, const double* d_w, int64_t n, int dim, const int* d_sel,
                     int64_t nsel, char* scratch, double* d_out);
int modes_dcc_device(sc_ctx* ctx, const double* d_v, const double* d_w, int64_t n, int dim, const int* d_sel,
                     int64_t nsel, int norm, char* scratch, double* d_out);
int modes_prs_device(sc_ctx* ctx, const double* d_v, const double* d_w, int64_t n, double rcond, int norm,
                     char* scratch, double* d_out);

// Raises a kernel's dynamic LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (device, kernel).  The
// attribute belongs to the device that is current when it is set: a function-local static done-flag (rounds 2-5) served
// the first device a process used and left contexts on a second GPU with refused launches.  Returns
// whether the limit is in place on the CURRENT device; the caller has made its context's device current.
inline bool sc_raise_dyn_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, bool> done;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> g(mu);
  const auto key = std::make_pair(dev, fn);
  const auto it = done.find(key);
  if (it != done.end()) return it->second;
  const bool ok = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  done[key] = ok;
  return ok;
}
