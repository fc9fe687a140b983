// C-ABI entry points of libspringcraft_hip.so (declared in include/springcraft_hip.h).
// Host-side plumbing only: argument validation, patch-table construction, transfers and
// kernel sequencing.  All arithmetic lives in assembly.hip / the eigensolver units.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <map>
#include <new>
#include <utility>

#include "common.h"
#include "host_logic.h"

int sc_set_error(sc_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

int sc_reserve_ws(sc_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return SC_OK;
  if (ctx->ws) {
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  SC_HIP(ctx, hipMalloc(&ctx->ws, bytes));
  ctx->ws_bytes = bytes;
  return SC_OK;
}

int sc_reserve_scratch(sc_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return SC_OK;
  if (ctx->scratch) {
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
  }
  SC_HIP(ctx, hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return SC_OK;
}

int sc_stage_upload(sc_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  if (bytes == 0) return SC_OK;
  const int c = ctx->h_stage_cur;
  if (ctx->h_stage_pending[c]) {   // this arena was last used two solves ago: its copies must have left it
    SC_HIP(ctx, hipEventSynchronize(ctx->h_stage_done[c]));
    ctx->h_stage_pending[c] = false;
    ctx->h_stage_off[c] = 0;
  }
  const size_t need = align_up(ctx->h_stage_off[c] + bytes, 256);
  if (need > ctx->h_stage_bytes[c]) {
    // grow: copies of this solve may still be reading the old arena
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_stage[c]) SC_HIP(ctx, hipHostFree(ctx->h_stage[c]));
    ctx->h_stage[c] = nullptr;
    ctx->h_stage_bytes[c] = 0;
    ctx->h_stage_off[c] = 0;
    const size_t cap = std::max<size_t>(2 * need, (size_t)1 << 20);
    SC_HIP(ctx, hipHostMalloc((void**)&ctx->h_stage[c], cap, hipHostMallocDefault));
    ctx->h_stage_bytes[c] = cap;
  }
  if (!ctx->h_stage_done[c]) SC_HIP(ctx, hipEventCreateWithFlags(&ctx->h_stage_done[c], hipEventDisableTiming));
  char* slot = ctx->h_stage[c] + ctx->h_stage_off[c];
  memcpy(slot, h_src, bytes);
  ctx->h_stage_off[c] = align_up(ctx->h_stage_off[c] + bytes, 256);
  SC_HIP(ctx, hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
  return SC_OK;
}

int sc_stage_end(sc_ctx* ctx) {
  const int c = ctx->h_stage_cur;
  if (!ctx->h_stage_done[c] || ctx->h_stage_off[c] == 0) return SC_OK;
  SC_HIP(ctx, hipEventRecord(ctx->h_stage_done[c], ctx->stream));
  ctx->h_stage_pending[c] = true;
  ctx->h_stage_cur = 1 - c;
  return SC_OK;
}

int sc_collect_events(sc_ctx* ctx) {
  if (!ctx->d_status) return SC_OK;
  unsigned long long h[kSpStatusWords] = {0};
  SC_HIP(ctx, hipMemcpy(h, ctx->d_status, sizeof(h), hipMemcpyDeviceToHost));
  bool any = false;
  for (int i = 2; i < kSpStatusWords; ++i) any = any || h[i] != 0;
  if (any) SC_HIP(ctx, hipMemset(ctx->d_status + 2, 0, sizeof(unsigned long long) * (kSpStatusWords - 2)));
  ctx->cnt_chase_resumed += (long long)h[2];
  ctx->cnt_chase_timeouts += (long long)h[3];
  ctx->cnt_chase_sweeps += (long long)h[4];
  ctx->cnt_coop_timeouts += (long long)h[5];
  ctx->cnt_chase_incomplete += (long long)h[6];
  ctx->cnt_resident_takeovers += (long long)h[7];
  ctx->cnt_resident_rollcalls += (long long)h[8];
  ctx->cnt_resident_lost += (long long)h[9];
  if (h[7]) ctx->resident_lost_at = (long long)h[10];
  // (one failed roll call can be a launch whose workgroups were dispatched late; the context gives the kernel up at the third)
  ctx->resident_strikes += (int)h[8];
  if (h[9] || (h[8] && ctx->resident_strikes >= 3)) ctx->resident_ok = 0;
  // a context whose persistent kernels ran into a bound keeps to the launch-per-wavefront / chunked forms from here on
  if (h[3] || h[6]) ctx->chase_ok = 0;
  if (h[5]) ctx->coop_ok = 0;
  if (ctx->last_chase_ctl) {   // the most recent chase's control block: tickets per XCD, where a wait timed out
    int c[32] = {0};
    SC_HIP(ctx, hipMemcpy(c, ctx->last_chase_ctl, sizeof(c), hipMemcpyDeviceToHost));
    for (int x = 0; x < 8; ++x) ctx->chase_tickets[x] = c[2 + x];
    if (c[0] && c[1] == 0) { ctx->chase_wait[0] = c[10]; ctx->chase_wait[1] = c[11]; ctx->chase_wait[2] = c[12]; }
    ctx->last_chase_ctl = nullptr;
  }
  return SC_OK;
}

int sc_deferred_status(sc_ctx* ctx) {
  if (!ctx->d_status) return SC_OK;
  SC_TRY(sc_collect_events(ctx));
  unsigned long long h[2] = {0, 0};
  SC_HIP(ctx, hipMemcpy(h, ctx->d_status, sizeof(h), hipMemcpyDeviceToHost));
  if (h[0] == 0 && h[1] == 0) return SC_OK;
  SC_HIP(ctx, hipMemset(ctx->d_status, 0, sizeof(h)));
  if (h[0])   // np.linalg.eigh raises LinAlgError("Eigenvalues did not converge") for such input (nma.py:61)
    return sc_set_error(ctx, SC_ERR_NOCONV, "Eigenvalues did not converge: matrix %llu of the batch contains NaN or Inf",
                        h[0] - 1ull);
  return sc_set_error(ctx, SC_ERR_NOCONV, "tridiagonal QL iteration did not converge");
}

int sc_reserve_pinv(sc_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pinv_ws_bytes) return SC_OK;
  if (ctx->pinv_ws) {
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_HIP(ctx, hipFree(ctx->pinv_ws));
    ctx->pinv_ws = nullptr;
    ctx->pinv_ws_bytes = 0;
  }
  SC_HIP(ctx, hipMalloc(&ctx->pinv_ws, bytes));
  ctx->pinv_ws_bytes = bytes;
  return SC_OK;
}

int sc_reserve_dc_aux(sc_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->dc_aux_bytes) return SC_OK;
  if (ctx->dc_aux) {
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SC_TRY(sc_collect_events(ctx));   // (the last chase's control block lives in the buffer that goes away)
    SC_HIP(ctx, hipFree(ctx->dc_aux));
    ctx->dc_aux = nullptr;
    ctx->dc_aux_bytes = 0;
  }
  SC_HIP(ctx, hipMalloc(&ctx->dc_aux, bytes));
  ctx->dc_aux_bytes = bytes;
  return SC_OK;
}

int sc_aux_stream(sc_ctx* ctx) {
  if (ctx->aux_stream) return SC_OK;
  SC_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
  SC_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
  SC_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
  return SC_OK;
}

int sc_side_streams(sc_ctx* ctx, int count) {
  while ((int)ctx->side_streams.size() < count) {
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    SC_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(s);
      return sc_set_error(ctx, SC_ERR_HIP, "hipEventCreateWithFlags failed");
    }
    ctx->side_streams.push_back(s);
    ctx->side_joins.push_back(e);
  }
  return SC_OK;
}

namespace {

int ctx_create_impl(int device, void* stream, bool own, sc_ctx** out) {
  if (!out) return SC_ERR_INVALID_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return SC_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SC_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SC_ERR_NO_DEVICE;  // gfx950 code objects only
  if (hipSetDevice(device) != hipSuccess) return SC_ERR_NO_DEVICE;
  sc_ctx* ctx = new sc_ctx();
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount;
  if (own) {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return SC_ERR_HIP;
    }
    ctx->own_stream = true;
  } else {
    ctx->stream = (hipStream_t)stream;
  }
  // (d_zeros: the page of zeros k_symm3's loaders read -- symm3.hip --, zeroed here, long before any stream uses it)
  if (hipMalloc((void**)&ctx->d_status, kSpStatusWords * sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(ctx->d_status, 0, kSpStatusWords * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void**)&ctx->d_zeros, 16384) != hipSuccess || hipMemset(ctx->d_zeros, 0, 16384) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->d_zeros) (void)hipFree(ctx->d_zeros);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return SC_ERR_NOMEM;
  }
  *out = ctx;
  return SC_OK;
}

int check_ff(sc_ctx* ctx, const sc_ff_desc* ff) {
  std::string err;
  const int rc = sc_host::check_ff(ff, err);
  return rc == SC_OK ? SC_OK : sc_set_error(ctx, rc, "%s", err.c_str());
}

struct Bump {  // carves sub-buffers out of ctx->scratch
  char* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* p = reinterpret_cast<T*>(base + off);
    off += count * sizeof(T);
    return p;
  }
};

using sc_host::HostPatch;

int build_patch(sc_ctx* ctx, const sc_patch_desc* pd, int64_t n, HostPatch& hp) {
  std::string err;
  const int rc = sc_host::build_patch(pd, n, hp, err);
  return rc == SC_OK ? SC_OK : sc_set_error(ctx, rc, "%s", err.c_str());
}

int upload_patch(sc_ctx* ctx, const HostPatch& hp, Bump& bump, PatchDev& dev) {
  uint8_t* d_shut = bump.take<uint8_t>(hp.shut.size());
  int32_t* d_rp = bump.take<int32_t>(hp.row_ptr.size());
  int32_t* d_col = bump.take<int32_t>(hp.col.size() + 1);
  int8_t* d_flag = bump.take<int8_t>(hp.flag.size() + 1);
  double* d_gam = bump.take<double>(hp.gam.size() + 1);
  SC_HIP(ctx, hipMemcpyAsync(d_shut, hp.shut.data(), hp.shut.size(), hipMemcpyHostToDevice, ctx->stream));
  SC_HIP(ctx, hipMemcpyAsync(d_rp, hp.row_ptr.data(), hp.row_ptr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  if (!hp.col.empty()) {
    SC_HIP(ctx, hipMemcpyAsync(d_col, hp.col.data(), hp.col.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(d_flag, hp.flag.data(), hp.flag.size(), hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(d_gam, hp.gam.data(), hp.gam.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  // the source vectors must outlive the async copies
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  dev = PatchDev{d_shut, d_rp, d_col, d_flag, d_gam, hp.mask_gamma};
  return SC_OK;
}

int check_coord_args(sc_ctx* ctx, const double* coord, int64_t n) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n < 0 || n > 2000000) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad atom count %lld", (long long)n);
  if (n > 0 && !coord) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "coord is NULL");
  return SC_OK;
}

// Shared body of sc_kirchhoff_f64 / sc_hessian_f64 / sc_contacts / fused eigen entry points:
// stages coord (+ weights, patches) into scratch, returns device pointers.
struct Staged {
  double* d_coord = nullptr;
  double* d_w = nullptr;
  PatchDev patch{};
  bool has_patch = false;
  HostPatch hp;  // keeps host vectors alive until the stream is synchronised
  sc_tab_desc tab_dev{};   // SC_FF_TABULATED: same fields, DEVICE pointers
  sc_ff_desc ff_dev{};     // copy of the caller's descriptor with .tab -> &tab_dev
};

size_t tab_device_bytes(const sc_ff_desc* ff, int64_t n) {
  if (!ff || ff->kind != SC_FF_TABULATED || !ff->tab) return 0;
  const size_t nb = (size_t)ff->tab->n_bins;
  return align_up(nb * 8, 256) + align_up(3 * 400 * nb * 4, 256) + 2 * align_up((size_t)n * 4, 256) +
         align_up((size_t)n, 256) + 2048;
}

// Upload the tabulated force-field tables; st.ff_dev is what the kernel launchers must be given.
int stage_tab(sc_ctx* ctx, const sc_ff_desc* ff, int64_t n, Staged& st, Bump& bump) {
  st.ff_dev = *ff;
  if (ff->kind != SC_FF_TABULATED) return SC_OK;
  const sc_tab_desc* t = ff->tab;
  const size_t nb = (size_t)t->n_bins;
  for (int64_t i = 0; i < n; ++i)
    if (t->atom_type[i] < 0 || t->atom_type[i] >= 20)
      return sc_set_error(ctx, SC_ERR_INDEX, "amino-acid type %d of atom %lld out of range", t->atom_type[i],
                          (long long)i);
  double* d_edges = bump.take<double>(nb);
  float* d_tab = bump.take<float>(3 * 400 * nb);
  int32_t* d_type = bump.take<int32_t>((size_t)n);
  int32_t* d_chain = bump.take<int32_t>((size_t)n);
  uint8_t* d_bond = bump.take<uint8_t>((size_t)n);
  hipStream_t s = ctx->stream;
  if (t->edges_sq) SC_HIP(ctx, hipMemcpyAsync(d_edges, t->edges_sq, nb * 8, hipMemcpyHostToDevice, s));
  SC_HIP(ctx, hipMemcpyAsync(d_tab, t->bonded, 400 * nb * 4, hipMemcpyHostToDevice, s));
  SC_HIP(ctx, hipMemcpyAsync(d_tab + 400 * nb, t->intra_chain, 400 * nb * 4, hipMemcpyHostToDevice, s));
  SC_HIP(ctx, hipMemcpyAsync(d_tab + 800 * nb, t->inter_chain, 400 * nb * 4, hipMemcpyHostToDevice, s));
  SC_HIP(ctx, hipMemcpyAsync(d_type, t->atom_type, (size_t)n * 4, hipMemcpyHostToDevice, s));
  SC_HIP(ctx, hipMemcpyAsync(d_chain, t->chain, (size_t)n * 4, hipMemcpyHostToDevice, s));
  SC_HIP(ctx, hipMemcpyAsync(d_bond, t->bonded_next, (size_t)n, hipMemcpyHostToDevice, s));
  SC_HIP(ctx, hipStreamSynchronize(s));
  st.tab_dev = *t;
  st.tab_dev.edges_sq = d_edges;
  st.tab_dev.bonded = d_tab;
  st.tab_dev.intra_chain = d_tab + 400 * nb;
  st.tab_dev.inter_chain = d_tab + 800 * nb;
  st.tab_dev.atom_type = d_type;
  st.tab_dev.chain = d_chain;
  st.tab_dev.bonded_next = d_bond;
  st.ff_dev.tab = &st.tab_dev;
  return SC_OK;
}

int stage_inputs(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff, const sc_patch_desc* pd,
                 const double* inv_sqrt_mass, size_t extra_bytes, Staged& st, Bump& bump) {
  SC_TRY(build_patch(ctx, pd, n, st.hp));
  size_t need = align_up((size_t)n * 24, 256) + align_up((size_t)n * 8, 256) + 1024 + extra_bytes +
                tab_device_bytes(ff, n);
  if (st.hp.any) need += st.hp.device_bytes();
  SC_TRY(sc_reserve_scratch(ctx, need));
  bump.base = (char*)ctx->scratch;
  bump.off = 0;
  st.d_coord = bump.take<double>((size_t)n * 3);
  SC_HIP(ctx, hipMemcpyAsync(st.d_coord, coord, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
  if (inv_sqrt_mass) {
    st.d_w = bump.take<double>((size_t)n);
    SC_HIP(ctx, hipMemcpyAsync(st.d_w, inv_sqrt_mass, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  if (st.hp.any) {
    SC_TRY(upload_patch(ctx, st.hp, bump, st.patch));
    st.has_patch = true;
  }
  SC_TRY(stage_tab(ctx, ff, n, st, bump));
  return SC_OK;
}

}  // namespace

extern "C" {

int sc_ctx_create(int device, sc_ctx** out) { return ctx_create_impl(device, nullptr, true, out); }

int sc_ctx_create_on_stream(int device, void* hip_stream, sc_ctx** out) {
  return ctx_create_impl(device, hip_stream, false, out);
}

void sc_ctx_destroy(sc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->dc_aux) (void)hipFree(ctx->dc_aux);
  if (ctx->pinv_ws) (void)hipFree(ctx->pinv_ws);
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  if (ctx->d_zeros) (void)hipFree(ctx->d_zeros);
  for (int c = 0; c < 2; ++c) {
    if (ctx->h_stage[c]) (void)hipHostFree(ctx->h_stage[c]);
    if (ctx->h_stage_done[c]) (void)hipEventDestroy(ctx->h_stage_done[c]);
  }
  if (ctx->aux_stream) {
    (void)hipStreamSynchronize(ctx->aux_stream);
    (void)hipStreamDestroy(ctx->aux_stream);
    (void)hipEventDestroy(ctx->aux_fork);
    (void)hipEventDestroy(ctx->aux_join);
  }
  for (size_t i = 0; i < ctx->side_streams.size(); ++i) {
    (void)hipStreamSynchronize(ctx->side_streams[i]);
    (void)hipStreamDestroy(ctx->side_streams[i]);
    (void)hipEventDestroy(ctx->side_joins[i]);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* sc_last_error(sc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int sc_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return SC_ERR_INVALID_ARG;
  *out = nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
    (void)hipGetLastError();
    return SC_ERR_NOMEM;
  }
  *out = p;
  return SC_OK;
}

int sc_host_free(void* p) {
  if (!p) return SC_OK;
  return hipHostFree(p) == hipSuccess ? SC_OK : SC_ERR_HIP;
}

int sc_ctx_synchronize(sc_ctx* ctx) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return sc_deferred_status(ctx);   // errors of the device-pointer solves enqueued since the last call
}

int sc_device_info(sc_ctx* ctx, char* buf, size_t buflen) {
  if (!ctx || !buf || buflen == 0) return SC_ERR_INVALID_ARG;
  hipDeviceProp_t prop;
  SC_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
  snprintf(buf, buflen, "%s %s, %d CUs, %.0f GiB", prop.gcnArchName, prop.name,
           prop.multiProcessorCount, (double)prop.totalGlobalMem / (1 << 30));
  return SC_OK;
}

int sc_ctx_set_profiling(sc_ctx* ctx, int enabled) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  ctx->profiling = enabled != 0;
  return SC_OK;
}

int sc_last_eigh_timings(sc_ctx* ctx, double* out6) {
  if (!ctx || !out6) return SC_ERR_INVALID_ARG;
  for (int i = 0; i < 6; ++i) out6[i] = ctx->last_timings[i];
  return SC_OK;
}

int sc_last_eigh_phase_ms(sc_ctx* ctx, const char* name, double* ms) {
  if (!ctx || !name || !ms) return SC_ERR_INVALID_ARG;
  for (const auto& p : ctx->phases)
    if (p.first == name) { *ms = p.second; return SC_OK; }
  // a phase the last solve did not run is an answer, not a failure: the context's error string is left alone
  *ms = 0.0;
  return SC_ERR_INVALID_ARG;
}

int sc_ctx_get_counter(sc_ctx* ctx, const char* name, int64_t* value) {
  if (!ctx || !name || !value) return SC_ERR_INVALID_ARG;
  const std::string k(name);
  // (round 6: what the persistent kernels report is collected on the device while solves are enqueued; a diagnostic call
  // may wait for the stream)
  SC_HIP(ctx, hipSetDevice(ctx->device));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SC_TRY(sc_collect_events(ctx));
  int tmin = ctx->chase_tickets[0], tmax = ctx->chase_tickets[0];
  for (int x = 1; x < 8; ++x) {
    tmin = std::min(tmin, ctx->chase_tickets[x]);
    tmax = std::max(tmax, ctx->chase_tickets[x]);
  }
  if (k == "chase_launches") *value = ctx->cnt_chase_launches;
  else if (k == "chase_pair_launches") *value = ctx->cnt_pair_launches;
  else if (k == "chase_pair_fallbacks") *value = ctx->cnt_pair_fallbacks;
  else if (k == "xcd_count") *value = ctx->nxcd;
  else if (k == "gemm3_launches") *value = ctx->cnt_gemm3_launches;
  else if (k == "symm3_launches") *value = ctx->cnt_symm3_launches;
  else if (k == "resident_launches") *value = ctx->cnt_resident_launches;
  else if (k == "resident_takeovers") *value = ctx->cnt_resident_takeovers;
  else if (k == "resident_rollcall_failures") *value = ctx->cnt_resident_rollcalls;
  else if (k == "resident_lost_waits") *value = ctx->cnt_resident_lost;
  else if (k == "resident_lost_at") *value = ctx->resident_lost_at;
  else if (k == "panel_coop_launches") *value = ctx->cnt_coop_launches;
  else if (k == "panel_coop_timeouts") *value = ctx->cnt_coop_timeouts;
  else if (k == "chase_timeouts") *value = ctx->cnt_chase_timeouts;
  else if (k == "chase_incomplete") *value = ctx->cnt_chase_incomplete;
  else if (k == "chase_resumed") *value = ctx->cnt_chase_resumed;
  else if (k == "chase_sweeps") *value = ctx->cnt_chase_sweeps;
  else if (k == "stepwise_chases") *value = ctx->cnt_stepwise_chases;
  else if (k == "chase_xcd_min") *value = tmin;
  else if (k == "chase_xcd_max") *value = tmax;
  else if (k == "chase_wait_matrix") *value = ctx->chase_wait[0];
  else if (k == "chase_wait_sweep") *value = ctx->chase_wait[1];
  else if (k == "chase_wait_task") *value = ctx->chase_wait[2];
  else { *value = 0; return SC_ERR_INVALID_ARG; }
  return SC_OK;
}

// debugging entry point (include/springcraft_hip_debug.h)
int sc_dbg_set_chase(sc_ctx* ctx, int mode, int give_up_after) {
  if (!ctx || mode < -1 || mode > 5 || give_up_after < 0) return SC_ERR_INVALID_ARG;
  // 3 / 4 / 5: persistent always, in the pair form / with one sweep per workgroup, a matrix on one XCD / with one sweep per
  // workgroup and the workgroups of a matrix on all XCDs ("spread": fewer matrices than XCDs only); 2: always, the form by size
  ctx->chase_form = mode == 3 ? 1 : (mode == 4 ? 0 : (mode == 5 ? 2 : -1));
  if (mode > 2) mode = 2;
  ctx->chase_mode = mode;
  ctx->chase_give_up = give_up_after;
  if (mode >= 0) ctx->chase_ok = -1;
  return SC_OK;
}

int sc_dbg_set_panel_coop_fail(sc_ctx* ctx, int panel) {
  if (!ctx || panel < -1) return SC_ERR_INVALID_ARG;
  ctx->coop_fail_panel = panel;
  if (panel >= 0) ctx->coop_ok = -1;
  return SC_OK;
}

int sc_dbg_set_resident(sc_ctx* ctx, int mode, int hook, int workgroups) {
  if (!ctx || mode < -1 || mode > 1 || hook < 0 || workgroups < 0 || workgroups > 256) return SC_ERR_INVALID_ARG;
  ctx->resident_mode = mode;
  ctx->resident_hook = hook;
  ctx->resident_wgs = workgroups;
  if (mode != 0) { ctx->resident_ok = -1; ctx->resident_strikes = 0; }
  return SC_OK;
}

int sc_dbg_set_panel_coop(sc_ctx* ctx, int min_rows) {
  if (!ctx || min_rows < -1 || (min_rows > 0 && min_rows < 128)) return SC_ERR_INVALID_ARG;
  ctx->coop_min_rows = min_rows;
  if (min_rows != 0) ctx->coop_ok = -1;
  return SC_OK;
}

int sc_contacts(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                const sc_patch_desc* patch, int64_t* counts, int64_t* n_pairs) {
  SC_TRY(check_coord_args(ctx, coord, n));
  SC_TRY(check_ff(ctx, ff));
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n_pairs) *n_pairs = 0;
  if (n == 0) return SC_OK;
  Staged st;
  Bump bump{};
  SC_TRY(stage_inputs(ctx, coord, n, ff, patch, nullptr, (size_t)n * 8 + 256, st, bump));
  int64_t* d_counts = bump.take<int64_t>((size_t)n);
  SC_TRY(launch_contact_counts(ctx, st.d_coord, n, st.ff_dev, st.has_patch ? &st.patch : nullptr, d_counts));
  std::vector<int64_t> h((size_t)n);
  SC_HIP(ctx, hipMemcpyAsync(h.data(), d_counts, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int64_t total = 0;
  for (int64_t i = 0; i < n; ++i) total += h[(size_t)i];
  if (counts) std::memcpy(counts, h.data(), (size_t)n * 8);
  if (n_pairs) *n_pairs = total;
  return SC_OK;
}

int sc_pairs(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
             const sc_patch_desc* patch, int64_t capacity, int64_t* pairs, double* sq_dist,
             int64_t* n_pairs) {
  SC_TRY(check_coord_args(ctx, coord, n));
  SC_TRY(check_ff(ctx, ff));
  if (capacity < 0 || (capacity > 0 && !pairs))
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "pairs buffer is NULL");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n_pairs) *n_pairs = 0;
  if (n == 0) return SC_OK;
  Staged st;
  Bump bump{};
  const size_t extra = (size_t)(2 * n + 2) * 8 + (size_t)capacity * 24 + 2048;
  SC_TRY(stage_inputs(ctx, coord, n, ff, patch, nullptr, extra, st, bump));
  int64_t* d_counts = bump.take<int64_t>((size_t)n);
  int64_t* d_off = bump.take<int64_t>((size_t)n + 1);
  int64_t* d_pairs = bump.take<int64_t>((size_t)capacity * 2 + 2);
  double* d_sq = sq_dist ? bump.take<double>((size_t)capacity + 1) : nullptr;
  const PatchDev* pdev = st.has_patch ? &st.patch : nullptr;
  SC_TRY(launch_contact_counts(ctx, st.d_coord, n, st.ff_dev, pdev, d_counts));
  // offsets of the rows in the ordered pair list: exclusive scan on the device, only the total comes back
  SC_TRY(launch_exclusive_scan_i64(ctx, d_counts, n, d_off));
  int64_t k = 0;
  SC_HIP(ctx, hipMemcpyAsync(&k, d_off + n, 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (n_pairs) *n_pairs = k;
  if (k > capacity)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "pair buffer too small: %lld > %lld", (long long)k,
                        (long long)capacity);
  if (k == 0) return SC_OK;
  SC_TRY(launch_pair_fill(ctx, st.d_coord, n, st.ff_dev, pdev, d_off, d_pairs, d_sq));
  SC_HIP(ctx, hipMemcpyAsync(pairs, d_pairs, (size_t)k * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (sq_dist)
    SC_HIP(ctx, hipMemcpyAsync(sq_dist, d_sq, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

static int assemble_host(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                         const sc_patch_desc* patch, const double* inv_sqrt_mass, double* out,
                         int dim) {
  SC_TRY(check_coord_args(ctx, coord, n));
  SC_TRY(check_ff(ctx, ff));
  if (n > 0 && !out) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "output matrix is NULL");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return SC_OK;
  const size_t elems = (size_t)n * n * dim * dim;
  Staged st;
  Bump bump{};
  SC_TRY(stage_inputs(ctx, coord, n, ff, patch, inv_sqrt_mass, elems * 8 + 512, st, bump));
  double* d_m = bump.take<double>(elems);
  const PatchDev* pdev = st.has_patch ? &st.patch : nullptr;
  if (dim == 1)
    SC_TRY(launch_kirchhoff(ctx, st.d_coord, n, 1, st.ff_dev, pdev, st.d_w, d_m, nullptr));
  else
    SC_TRY(launch_hessian(ctx, st.d_coord, n, 1, st.ff_dev, pdev, st.d_w, d_m));
  SC_HIP(ctx, hipMemcpyAsync(out, d_m, elems * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_kirchhoff_f64(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                     const sc_patch_desc* patch, const double* inv_sqrt_mass, double* kirchhoff) {
  return assemble_host(ctx, coord, n, ff, patch, inv_sqrt_mass, kirchhoff, 1);
}

int sc_hessian_f64(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                   const sc_patch_desc* patch, const double* inv_sqrt_mass, double* hessian) {
  return assemble_host(ctx, coord, n, ff, patch, inv_sqrt_mass, hessian, 3);
}

static int check_pairs(sc_ctx* ctx, int64_t n, const int64_t* pairs, int64_t k, const double* gamma) {
  if (k < 0 || (k > 0 && (!pairs || !gamma)))
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "pairs / gamma is NULL");
  for (int64_t p = 0; p < 2 * k; ++p)
    if (pairs[p] < 0 || pairs[p] >= n)
      return sc_set_error(ctx, SC_ERR_INDEX, "pair index %lld out of range for %lld atoms",
                          (long long)pairs[p], (long long)n);
  return SC_OK;
}

int sc_kirchhoff_from_pairs_f64(sc_ctx* ctx, int64_t n, const int64_t* pairs, int64_t k,
                                const double* gamma, double* kirchhoff) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && !kirchhoff)) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_TRY(check_pairs(ctx, n, pairs, k, gamma));
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return SC_OK;
  const size_t elems = (size_t)n * n;
  SC_TRY(sc_reserve_scratch(ctx, elems * 8 + (size_t)k * 24 + 4096));
  Bump bump{(char*)ctx->scratch};
  double* d_m = bump.take<double>(elems);
  int64_t* d_pairs = bump.take<int64_t>((size_t)k * 2 + 2);
  double* d_g = bump.take<double>((size_t)k + 1);
  if (k > 0) {
    SC_HIP(ctx, hipMemcpyAsync(d_pairs, pairs, (size_t)k * 16, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(d_g, gamma, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  SC_TRY(launch_kirchhoff_from_pairs(ctx, n, d_pairs, k, d_g, d_m));
  SC_HIP(ctx, hipMemcpyAsync(kirchhoff, d_m, elems * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_hessian_from_pairs_f64(sc_ctx* ctx, const double* coord, int64_t n, const int64_t* pairs,
                              int64_t k, const double* gamma, double* hessian) {
  SC_TRY(check_coord_args(ctx, coord, n));
  if (n > 0 && !hessian) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "output matrix is NULL");
  SC_TRY(check_pairs(ctx, n, pairs, k, gamma));
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return SC_OK;
  const size_t elems = (size_t)n * n * 9;
  SC_TRY(sc_reserve_scratch(ctx, elems * 8 + (size_t)k * 24 + (size_t)n * 24 + 4096));
  Bump bump{(char*)ctx->scratch};
  double* d_m = bump.take<double>(elems);
  double* d_coord = bump.take<double>((size_t)n * 3);
  int64_t* d_pairs = bump.take<int64_t>((size_t)k * 2 + 2);
  double* d_g = bump.take<double>((size_t)k + 1);
  SC_HIP(ctx, hipMemcpyAsync(d_coord, coord, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
  if (k > 0) {
    SC_HIP(ctx, hipMemcpyAsync(d_pairs, pairs, (size_t)k * 16, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(d_g, gamma, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  SC_TRY(launch_hessian_from_pairs(ctx, d_coord, n, d_pairs, k, d_g, d_m));
  SC_HIP(ctx, hipMemcpyAsync(hessian, d_m, elems * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_eigh_f64(sc_ctx* ctx, const double* a, int64_t n, double* w, double* v) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!a || !w))) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return SC_OK;
  const size_t elems = (size_t)n * n;
  SC_TRY(sc_reserve_scratch(ctx, elems * 8 * (v ? 2 : 1) + (size_t)n * 8 + 4096));
  Bump bump{(char*)ctx->scratch};
  double* d_a = bump.take<double>(elems);
  double* d_w = bump.take<double>((size_t)n);
  double* d_v = v ? bump.take<double>(elems) : nullptr;
  SC_HIP(ctx, hipMemcpyAsync(d_a, a, elems * 8, hipMemcpyHostToDevice, ctx->stream));
  SC_TRY(eigh_batched(ctx, d_a, n, 1, d_w, d_v));
  SC_HIP(ctx, hipMemcpyAsync(w, d_w, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (v) SC_HIP(ctx, hipMemcpyAsync(v, d_v, elems * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_pinvh_f64(sc_ctx* ctx, const double* a, int64_t n, double rcond, double* out) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && (!a || !out))) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return SC_OK;
  const size_t elems = (size_t)n * n;
  SC_TRY(sc_reserve_scratch(ctx, 2 * elems * 8 + 4096));
  Bump bump{(char*)ctx->scratch};
  double* d_a = bump.take<double>(elems);
  double* d_out = bump.take<double>(elems);
  SC_HIP(ctx, hipMemcpyAsync(d_a, a, elems * 8, hipMemcpyHostToDevice, ctx->stream));
  SC_TRY(pinvh_device(ctx, d_a, n, rcond, d_out));
  SC_HIP(ctx, hipMemcpyAsync(out, d_out, elems * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_eigh_range_f64(sc_ctx* ctx, const double* a, int64_t n, int64_t il, int64_t iu, double* w, double* v) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n <= 0 || !a || !w || il < 0 || iu < il || iu >= n)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t m = iu - il + 1;
  const size_t elems = (size_t)n * n;
  SC_TRY(sc_reserve_scratch(ctx, elems * 8 + (size_t)m * 8 + (v ? (size_t)m * n * 8 : 0) + 4096));
  Bump bump{(char*)ctx->scratch};
  double* d_a = bump.take<double>(elems);
  double* d_w = bump.take<double>((size_t)m);
  double* d_v = v ? bump.take<double>((size_t)m * n) : nullptr;
  SC_HIP(ctx, hipMemcpyAsync(d_a, a, elems * 8, hipMemcpyHostToDevice, ctx->stream));
  SC_TRY(eigh_range_batched(ctx, d_a, n, 1, il, iu, d_w, d_v));
  SC_HIP(ctx, hipMemcpyAsync(w, d_w, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (v) SC_HIP(ctx, hipMemcpyAsync(v, d_v, (size_t)m * n * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_anm_eigen_range_f64(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                           const sc_patch_desc* patch, const double* inv_sqrt_mass, int64_t il, int64_t iu,
                           double* w, double* v) {
  SC_TRY(check_coord_args(ctx, coord, n));
  SC_TRY(check_ff(ctx, ff));
  const int64_t dim3n = 3 * n;
  if (n <= 0 || !w || il < 0 || iu < il || iu >= dim3n)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t m = iu - il + 1;
  const size_t elems = (size_t)dim3n * dim3n;
  Staged st;
  Bump bump{};
  SC_TRY(stage_inputs(ctx, coord, n, ff, patch, inv_sqrt_mass,
                      elems * 8 + (size_t)m * 8 + (v ? (size_t)m * dim3n * 8 : 0) + 4096, st, bump));
  double* d_m = bump.take<double>(elems);
  double* d_w = bump.take<double>((size_t)m);
  double* d_v = v ? bump.take<double>((size_t)m * dim3n) : nullptr;
  SC_TRY(launch_hessian(ctx, st.d_coord, n, 1, st.ff_dev, st.has_patch ? &st.patch : nullptr, st.d_w, d_m));
  SC_TRY(eigh_range_batched(ctx, d_m, dim3n, 1, il, iu, d_w, d_v));
  SC_HIP(ctx, hipMemcpyAsync(w, d_w, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (v) SC_HIP(ctx, hipMemcpyAsync(v, d_v, (size_t)m * dim3n * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_dev_eigh_range_f64(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, int64_t il, int64_t iu,
                          double* d_w, double* d_v) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n <= 0 || batch <= 0 || !d_a || !d_w) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  return eigh_range_batched_async(ctx, d_a, n, batch, il, iu, d_w, d_v);
}

static int enm_eigen_host(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                          const sc_patch_desc* patch, const double* inv_sqrt_mass, double* w,
                          double* v, int dim) {
  SC_TRY(check_coord_args(ctx, coord, n));
  SC_TRY(check_ff(ctx, ff));
  if (n > 0 && !w) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "w is NULL");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0) return SC_OK;
  const int64_t m = n * dim;
  const size_t elems = (size_t)m * m;
  Staged st;
  Bump bump{};
  SC_TRY(stage_inputs(ctx, coord, n, ff, patch, inv_sqrt_mass,
                      elems * 8 * (v ? 2 : 1) + (size_t)m * 8 + 2048, st, bump));
  double* d_m = bump.take<double>(elems);
  double* d_w = bump.take<double>((size_t)m);
  double* d_v = v ? bump.take<double>(elems) : nullptr;
  const PatchDev* pdev = st.has_patch ? &st.patch : nullptr;
  if (dim == 1)
    SC_TRY(launch_kirchhoff(ctx, st.d_coord, n, 1, st.ff_dev, pdev, st.d_w, d_m, nullptr));
  else
    SC_TRY(launch_hessian(ctx, st.d_coord, n, 1, st.ff_dev, pdev, st.d_w, d_m));
  SC_TRY(eigh_batched(ctx, d_m, m, 1, d_w, d_v));
  SC_HIP(ctx, hipMemcpyAsync(w, d_w, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (v) SC_HIP(ctx, hipMemcpyAsync(v, d_v, elems * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_anm_eigen_f64(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                     const sc_patch_desc* patch, const double* inv_sqrt_mass, double* w, double* v) {
  return enm_eigen_host(ctx, coord, n, ff, patch, inv_sqrt_mass, w, v, 3);
}

int sc_gnm_eigen_f64(sc_ctx* ctx, const double* coord, int64_t n, const sc_ff_desc* ff,
                     const sc_patch_desc* patch, const double* inv_sqrt_mass, double* w, double* v) {
  return enm_eigen_host(ctx, coord, n, ff, patch, inv_sqrt_mass, w, v, 1);
}

int sc_dev_kirchhoff_f64(sc_ctx* ctx, const double* d_coord, int64_t n, int64_t batch,
                         const sc_ff_desc* ff, const double* d_inv_sqrt_mass, double* d_matrix) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  SC_TRY(check_ff(ctx, ff));
  if (ff->kind == SC_FF_TABULATED)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "tabulated force fields are not supported on the batched device path");
  if (n < 0 || batch < 0 || (n > 0 && batch > 0 && (!d_coord || !d_matrix)))
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  return launch_kirchhoff(ctx, d_coord, n, batch, *ff, nullptr, d_inv_sqrt_mass, d_matrix, nullptr);
}

int sc_dev_hessian_f64(sc_ctx* ctx, const double* d_coord, int64_t n, int64_t batch,
                       const sc_ff_desc* ff, const double* d_inv_sqrt_mass, double* d_matrix) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  SC_TRY(check_ff(ctx, ff));
  if (ff->kind == SC_FF_TABULATED)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "tabulated force fields are not supported on the batched device path");
  if (n < 0 || batch < 0 || (n > 0 && batch > 0 && (!d_coord || !d_matrix)))
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  return launch_hessian(ctx, d_coord, n, batch, *ff, nullptr, d_inv_sqrt_mass, d_matrix);
}

int sc_dev_eigh_f64(sc_ctx* ctx, double* d_a, int64_t n, int64_t batch, double* d_w, double* d_v) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n < 0 || batch < 0 || (n > 0 && batch > 0 && (!d_a || !d_w)))
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (n == 0 || batch == 0) return SC_OK;
  return eigh_batched_async(ctx, d_a, n, batch, d_w, d_v);
}

int sc_ctx_set_two_stage(sc_ctx* ctx, int mode) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (mode < -1 || mode > 1) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "mode must be -1, 0 or 1");
  ctx->two_stage = mode;
  return SC_OK;
}

int64_t sc_eigh_workspace_bytes(int64_t n, int64_t batch, int want_vectors) {
  if (n <= 0 || batch <= 0) return 0;
  return (int64_t)eigh_workspace_bytes(n, batch, want_vectors != 0);
}

// ---- ragged / decorated batches ---------------------------------------------------------------------------------
}  // extern "C"

// Everything a batch of structures needs on the device besides coordinates: per-structure records (size, slot order,
// force-field descriptor with device table pointers, patch override tables), uploaded once at plan creation.
struct sc_batch_plan {
  sc_ctx* ctx = nullptr;
  int dim = 3;
  int64_t count = 0, order = 0;
  int max_atoms = 0;
  bool any_patch = false, any_pad = false;
  char* d_blob = nullptr;
  size_t off_items = 0, off_bound = 0;
  std::vector<int64_t> atom_off;   // count + 1: first atom of every structure in the packed buffers
};

namespace {

// Host image of the plan's device blob: take() reserves an aligned region and returns its offset.
struct Blob {
  std::vector<char> bytes;
  size_t take(size_t n) {
    const size_t off = align_up(bytes.size(), 256);
    bytes.resize(off + n, 0);
    return off;
  }
  template <typename T>
  size_t put(const T* src, size_t count) {
    const size_t off = take(count * sizeof(T) + 16);   // (+16: the kernels may read one element past empty tables)
    if (count) memcpy(bytes.data() + off, src, count * sizeof(T));
    return off;
  }
};

}  // namespace

extern "C" {

int sc_batch_plan_create(sc_ctx* ctx, int dim, const sc_structure_desc* structures, int64_t count, int64_t order,
                         sc_batch_plan** out) {
  if (!ctx || !out) return SC_ERR_INVALID_ARG;
  *out = nullptr;
  if ((dim != 1 && dim != 3) || count <= 0 || !structures)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments (dim %d, %lld structures)", dim, (long long)count);
  int64_t max_atoms = 0;
  for (int64_t b = 0; b < count; ++b) {
    if (structures[b].n_atoms <= 0 || structures[b].n_atoms > 2000000)
      return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad atom count %lld of structure %lld",
                          (long long)structures[b].n_atoms, (long long)b);
    SC_TRY(check_ff(ctx, structures[b].ff));
    max_atoms = std::max(max_atoms, structures[b].n_atoms);
  }
  if (order == 0) order = dim * max_atoms;
  if (order < dim * max_atoms || order > 46000)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "slot order %lld does not hold %d x %lld rows (or exceeds 46000)",
                        (long long)order, dim, (long long)max_atoms);
  SC_HIP(ctx, hipSetDevice(ctx->device));

  Blob blob;
  const size_t isz = asm_item_bytes();
  const size_t off_items = blob.take(isz * (size_t)count);
  const size_t off_bound = blob.take(8 * (size_t)count);
  const size_t off_zero = blob.take(4 * ((size_t)max_atoms + 2));   // empty patch tables: shut = 0, row_ptr = 0
  struct Rec {
    sc_ff_desc ff; sc_tab_desc tab; bool has_tab = false;
    size_t o_edges = 0, o_tables = 0, o_type = 0, o_chain = 0, o_bond = 0;
    bool has_patch = false; int mask_gamma = 0;
    size_t o_shut = 0, o_rp = 0, o_col = 0, o_flag = 0, o_gam = 0;
  };
  std::vector<Rec> recs((size_t)count);
  // parameter tables are shared by every structure that points at the same host arrays
  std::map<std::pair<const void*, int>, std::pair<size_t, size_t>> tables;   // (bonded ptr, bins) -> (edges, tables)
  bool any_patch = false, any_pad = false;
  for (int64_t b = 0; b < count; ++b) {
    const sc_structure_desc& sd = structures[b];
    Rec& r = recs[(size_t)b];
    const size_t n = (size_t)sd.n_atoms;
    r.ff = *sd.ff;
    if (dim * sd.n_atoms < order) any_pad = true;
    if (sd.ff->kind == SC_FF_TABULATED) {
      const sc_tab_desc* t = sd.ff->tab;
      const size_t nb = (size_t)t->n_bins;
      for (size_t i = 0; i < n; ++i)
        if (t->atom_type[i] < 0 || t->atom_type[i] >= 20)
          return sc_set_error(ctx, SC_ERR_INDEX, "amino-acid type %d of atom %lld of structure %lld out of range",
                              t->atom_type[i], (long long)i, (long long)b);
      r.has_tab = true;
      r.tab = *t;
      const auto key = std::make_pair((const void*)t->bonded, t->n_bins);
      auto it = tables.find(key);
      if (it == tables.end()) {
        const size_t oe = blob.take(nb * 8 + 16);
        if (t->edges_sq) memcpy(blob.bytes.data() + oe, t->edges_sq, nb * 8);
        const size_t ot = blob.take(3 * 400 * nb * 4);
        memcpy(blob.bytes.data() + ot, t->bonded, 400 * nb * 4);
        memcpy(blob.bytes.data() + ot + 400 * nb * 4, t->intra_chain, 400 * nb * 4);
        memcpy(blob.bytes.data() + ot + 800 * nb * 4, t->inter_chain, 400 * nb * 4);
        it = tables.emplace(key, std::make_pair(oe, ot)).first;
      }
      r.o_edges = it->second.first;
      r.o_tables = it->second.second;
      r.o_type = blob.put(t->atom_type, n);
      r.o_chain = blob.put(t->chain, n);
      r.o_bond = blob.put(t->bonded_next, n);
    }
    HostPatch hp;
    SC_TRY(build_patch(ctx, sd.patch, sd.n_atoms, hp));
    if (hp.any) {
      any_patch = true;
      r.has_patch = true;
      r.mask_gamma = hp.mask_gamma;
      r.o_shut = blob.put(hp.shut.data(), hp.shut.size());
      r.o_rp = blob.put(hp.row_ptr.data(), hp.row_ptr.size());
      r.o_col = blob.put(hp.col.data(), hp.col.size());
      r.o_flag = blob.put(hp.flag.data(), hp.flag.size());
      r.o_gam = blob.put(hp.gam.data(), hp.gam.size());
    }
  }
  char* d_blob = nullptr;
  SC_HIP(ctx, hipMalloc((void**)&d_blob, align_up(blob.bytes.size(), 256)));
  // the item records hold device addresses: fill them now that the blob's base is known
  long long atom_off = 0;
  for (int64_t b = 0; b < count; ++b) {
    Rec& r = recs[(size_t)b];
    const size_t nb = r.has_tab ? (size_t)r.tab.n_bins : 0;
    sc_ff_desc ffd = r.ff;
    sc_tab_desc td{};
    if (r.has_tab) {
      td = r.tab;
      td.edges_sq = reinterpret_cast<const double*>(d_blob + r.o_edges);
      td.bonded = reinterpret_cast<const float*>(d_blob + r.o_tables);
      td.intra_chain = td.bonded + 400 * nb;
      td.inter_chain = td.bonded + 800 * nb;
      td.atom_type = reinterpret_cast<const int32_t*>(d_blob + r.o_type);
      td.chain = reinterpret_cast<const int32_t*>(d_blob + r.o_chain);
      td.bonded_next = reinterpret_cast<const uint8_t*>(d_blob + r.o_bond);
      ffd.tab = &td;
    }
    PatchDev pdv{};
    if (r.has_patch) {
      pdv = PatchDev{reinterpret_cast<const uint8_t*>(d_blob + r.o_shut), reinterpret_cast<const int32_t*>(d_blob + r.o_rp),
                     reinterpret_cast<const int32_t*>(d_blob + r.o_col), reinterpret_cast<const int8_t*>(d_blob + r.o_flag),
                     reinterpret_cast<const double*>(d_blob + r.o_gam), r.mask_gamma};
    } else {
      pdv = PatchDev{reinterpret_cast<const uint8_t*>(d_blob + off_zero), reinterpret_cast<const int32_t*>(d_blob + off_zero),
                     reinterpret_cast<const int32_t*>(d_blob + off_zero), reinterpret_cast<const int8_t*>(d_blob + off_zero),
                     reinterpret_cast<const double*>(d_blob + off_zero), 0};
    }
    asm_item_fill(blob.bytes.data() + off_items + isz * (size_t)b, atom_off, (int)structures[b].n_atoms, (int)order, ffd, pdv);
    atom_off += structures[b].n_atoms;
  }
  if (hipMemcpy(d_blob, blob.bytes.data(), blob.bytes.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d_blob);
    return sc_set_error(ctx, SC_ERR_HIP, "upload of the batch plan failed");
  }
  sc_batch_plan* plan = new (std::nothrow) sc_batch_plan();
  if (!plan) { (void)hipFree(d_blob); return SC_ERR_NOMEM; }
  plan->ctx = ctx; plan->dim = dim; plan->count = count; plan->order = order; plan->max_atoms = (int)max_atoms;
  plan->any_patch = any_patch; plan->any_pad = any_pad;
  plan->d_blob = d_blob; plan->off_items = off_items; plan->off_bound = off_bound;
  plan->atom_off.assign((size_t)count + 1, 0);
  for (int64_t b = 0; b < count; ++b) plan->atom_off[(size_t)b + 1] = plan->atom_off[(size_t)b] + structures[b].n_atoms;
  *out = plan;
  return SC_OK;
}

int sc_batch_plan_assemble_f64(sc_batch_plan* plan, const double* d_coord, const double* d_inv_sqrt_mass,
                               double* d_matrix) {
  if (!plan) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = plan->ctx;
  if (!d_coord || !d_matrix) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  return launch_assemble_items(ctx, plan->dim, plan->d_blob + plan->off_items, plan->count, plan->max_atoms,
                               plan->any_patch, plan->any_pad, d_coord, d_inv_sqrt_mass, d_matrix,
                               reinterpret_cast<unsigned long long*>(plan->d_blob + plan->off_bound));
}

// Contact scan of every structure: counts per atom + their exclusive scan in the context's scratch; the per-structure
// offsets of the pair list (count + 1) come back to `pair_off`.  Leaves d_off (atoms + 1 entries) valid in scratch.
static int plan_scan(sc_batch_plan* plan, const double* d_coord, size_t extra_bytes, Bump& bump, int64_t** d_off_out,
                     std::vector<int64_t>& pair_off) {
  sc_ctx* ctx = plan->ctx;
  const int64_t atoms = plan->atom_off.back();
  SC_TRY(sc_reserve_scratch(ctx, (size_t)(2 * atoms + 2) * 8 + extra_bytes + 4096));
  bump = Bump{(char*)ctx->scratch};
  int64_t* d_counts = bump.take<int64_t>((size_t)atoms);
  int64_t* d_off = bump.take<int64_t>((size_t)atoms + 1);
  SC_TRY(launch_items_counts(ctx, plan->d_blob + plan->off_items, plan->count, plan->max_atoms, plan->any_patch, d_coord,
                             d_counts));
  SC_TRY(launch_exclusive_scan_i64(ctx, d_counts, atoms, d_off));
  std::vector<int64_t> off((size_t)atoms + 1);
  SC_HIP(ctx, hipMemcpyAsync(off.data(), d_off, ((size_t)atoms + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  pair_off.resize((size_t)plan->count + 1);
  for (int64_t b = 0; b <= plan->count; ++b) pair_off[(size_t)b] = off[(size_t)plan->atom_off[(size_t)b]];
  *d_off_out = d_off;
  return SC_OK;
}

int sc_batch_plan_contacts(sc_batch_plan* plan, const double* d_coord, int64_t* n_pairs) {
  if (!plan) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = plan->ctx;
  if (!d_coord || !n_pairs) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  Bump bump{};
  int64_t* d_off = nullptr;
  std::vector<int64_t> pair_off;
  SC_TRY(plan_scan(plan, d_coord, 0, bump, &d_off, pair_off));
  for (int64_t b = 0; b < plan->count; ++b) n_pairs[b] = pair_off[(size_t)b + 1] - pair_off[(size_t)b];
  return SC_OK;
}

int sc_batch_plan_pairs(sc_batch_plan* plan, const double* d_coord, int64_t capacity, int64_t* pairs, double* sq_dist,
                        int64_t* pair_off) {
  if (!plan) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = plan->ctx;
  if (!d_coord || !pair_off || capacity < 0 || (capacity > 0 && !pairs))
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  Bump bump{};
  int64_t* d_off = nullptr;
  std::vector<int64_t> off;
  SC_TRY(plan_scan(plan, d_coord, (size_t)capacity * 24 + 1024, bump, &d_off, off));
  for (int64_t b = 0; b <= plan->count; ++b) pair_off[b] = off[(size_t)b];
  const int64_t k = off.back();
  if (k > capacity)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "pair buffer too small: %lld > %lld", (long long)k, (long long)capacity);
  if (k == 0) return SC_OK;
  int64_t* d_pairs = bump.take<int64_t>((size_t)capacity * 2 + 2);
  double* d_sq = sq_dist ? bump.take<double>((size_t)capacity + 1) : nullptr;
  SC_TRY(launch_items_pair_fill(ctx, plan->d_blob + plan->off_items, plan->count, plan->max_atoms, plan->any_patch, d_coord,
                                d_off, d_pairs, d_sq));
  SC_HIP(ctx, hipMemcpyAsync(pairs, d_pairs, (size_t)k * 16, hipMemcpyDeviceToHost, ctx->stream));
  if (sq_dist) SC_HIP(ctx, hipMemcpyAsync(sq_dist, d_sq, (size_t)k * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_batch_plan_fill_from_pairs_f64(sc_batch_plan* plan, const double* d_coord, const int64_t* pairs,
                                      const int64_t* pair_off, const double* gamma, const double* d_inv_sqrt_mass,
                                      double* d_matrix) {
  if (!plan) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = plan->ctx;
  if (!d_coord || !pair_off || !d_matrix) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  if (pair_off[0] != 0) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "pair_off[0] must be 0");
  for (int64_t b = 0; b < plan->count; ++b) {
    if (pair_off[b + 1] < pair_off[b]) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "pair_off must not decrease");
    const int64_t nb = plan->atom_off[(size_t)b + 1] - plan->atom_off[(size_t)b];
    if (pair_off[b + 1] > pair_off[b] && (!pairs || !gamma))
      return sc_set_error(ctx, SC_ERR_INVALID_ARG, "pairs / gamma is NULL");
    for (int64_t p = 2 * pair_off[b]; p < 2 * pair_off[b + 1]; ++p)
      if (pairs[p] < 0 || pairs[p] >= nb)
        return sc_set_error(ctx, SC_ERR_INDEX, "pair index %lld out of range for the %lld atoms of structure %lld",
                            (long long)pairs[p], (long long)nb, (long long)b);
  }
  const int64_t k = pair_off[plan->count];
  SC_HIP(ctx, hipSetDevice(ctx->device));
  SC_TRY(sc_reserve_scratch(ctx, (size_t)k * 24 + ((size_t)plan->count + 1) * 8 + 4096));
  Bump bump{(char*)ctx->scratch};
  int64_t* d_poff = bump.take<int64_t>((size_t)plan->count + 1);
  int64_t* d_pairs = bump.take<int64_t>((size_t)k * 2 + 2);
  double* d_g = bump.take<double>((size_t)k + 1);
  // (pageable host memory: the copies are staged by the runtime before the call returns)
  SC_HIP(ctx, hipMemcpyAsync(d_poff, pair_off, ((size_t)plan->count + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
  if (k > 0) {
    SC_HIP(ctx, hipMemcpyAsync(d_pairs, pairs, (size_t)k * 16, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipMemcpyAsync(d_g, gamma, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  SC_TRY(launch_items_from_pairs(ctx, plan->dim, plan->d_blob + plan->off_items, plan->count, plan->max_atoms, plan->order,
                                 plan->any_pad, d_coord, d_poff, k, d_pairs, d_g, d_inv_sqrt_mass, d_matrix,
                                 reinterpret_cast<unsigned long long*>(plan->d_blob + plan->off_bound)));
  // the scratch arena is reused by the next call: the kernels that read it must have run
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int64_t sc_batch_plan_order(const sc_batch_plan* plan) { return plan ? plan->order : 0; }

void sc_batch_plan_destroy(sc_batch_plan* plan) {
  if (!plan) return;
  (void)hipSetDevice(plan->ctx->device);
  (void)hipStreamSynchronize(plan->ctx->stream);
  if (plan->d_blob) (void)hipFree(plan->d_blob);
  delete plan;
}

// ---- device-resident eigenpairs --------------------------------------------------------------------------
}  // extern "C"

struct sc_modes {
  sc_ctx* ctx = nullptr;
  int64_t n = 0;
  int dim = 1;
  double* d_w = nullptr;
  double* d_v = nullptr;
};

namespace {
int modes_alloc(sc_ctx* ctx, int64_t n, int dim, sc_modes** out) {
  sc_modes* m = new (std::nothrow) sc_modes();
  if (!m) return sc_set_error(ctx, SC_ERR_NOMEM, "out of host memory");
  m->ctx = ctx; m->n = n; m->dim = dim;
  if (hipMalloc(&m->d_w, sizeof(double) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess ||
      hipMalloc(&m->d_v, sizeof(double) * (size_t)std::max<int64_t>(n * n, 1)) != hipSuccess) {
    (void)hipFree(m->d_w);
    delete m;
    return sc_set_error(ctx, SC_ERR_NOMEM, "cannot allocate %lld x %lld eigenvectors on the device", (long long)n,
                        (long long)n);
  }
  *out = m;
  return SC_OK;
}

// mode list -> validated int32 list on the device (in scratch)
int stage_mode_list(sc_modes* m, const int64_t* idx, int64_t k, int* d_sel) {
  sc_ctx* ctx = m->ctx;
  std::vector<int> h((size_t)k);
  for (int64_t i = 0; i < k; ++i) {
    int64_t v = idx[i];
    if (v < 0) v += m->n;   // NumPy-style negative indices
    if (v < 0 || v >= m->n)
      return sc_set_error(ctx, SC_ERR_INDEX, "mode index %lld out of bounds for %lld modes", (long long)idx[i],
                          (long long)m->n);
    h[(size_t)i] = (int)v;
  }
  if (k > 0) {
    SC_HIP(ctx, hipMemcpyAsync(d_sel, h.data(), (size_t)k * 4, hipMemcpyHostToDevice, ctx->stream));
    SC_HIP(ctx, hipStreamSynchronize(ctx->stream));   // h goes out of scope
  }
  return SC_OK;
}
}  // namespace

extern "C" {

int sc_modes_from_coord(sc_ctx* ctx, const double* coord, int64_t n, int dim, const sc_ff_desc* ff,
                        const sc_patch_desc* patch, const double* inv_sqrt_mass, sc_modes** out) {
  SC_TRY(check_coord_args(ctx, coord, n));
  SC_TRY(check_ff(ctx, ff));
  if (!out || (dim != 1 && dim != 3)) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  const int64_t m = n * dim;
  const size_t elems = (size_t)m * m;
  sc_modes* md = nullptr;
  SC_TRY(modes_alloc(ctx, m, dim, &md));
  int rc = SC_OK;
  if (n > 0) {
    Staged st;
    Bump bump{};
    rc = stage_inputs(ctx, coord, n, ff, patch, inv_sqrt_mass, elems * 8 + 2048, st, bump);
    if (rc == SC_OK) {
      double* d_m = bump.take<double>(elems);
      const PatchDev* pdev = st.has_patch ? &st.patch : nullptr;
      rc = dim == 1 ? launch_kirchhoff(ctx, st.d_coord, n, 1, st.ff_dev, pdev, st.d_w, d_m, nullptr)
                    : launch_hessian(ctx, st.d_coord, n, 1, st.ff_dev, pdev, st.d_w, d_m);
      if (rc == SC_OK) rc = eigh_batched(ctx, d_m, m, 1, md->d_w, md->d_v);
      if (rc == SC_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = sc_set_error(ctx, SC_ERR_HIP, "eigensolve failed");
    }
  }
  if (rc != SC_OK) { sc_modes_destroy(md); return rc; }
  *out = md;
  return SC_OK;
}

int sc_modes_from_matrix(sc_ctx* ctx, const double* a, int64_t n, int dim, sc_modes** out) {
  if (!ctx) return SC_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && !a) || !out || (dim != 1 && dim != 3) || n % dim != 0)
    return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  sc_modes* md = nullptr;
  SC_TRY(modes_alloc(ctx, n, dim, &md));
  int rc = SC_OK;
  if (n > 0) {
    const size_t elems = (size_t)n * n;
    rc = sc_reserve_scratch(ctx, elems * 8 + 1024);
    if (rc == SC_OK) {
      double* d_a = reinterpret_cast<double*>(ctx->scratch);
      if (hipMemcpyAsync(d_a, a, elems * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = sc_set_error(ctx, SC_ERR_HIP, "upload failed");
      if (rc == SC_OK) rc = eigh_batched(ctx, d_a, n, 1, md->d_w, md->d_v);
      if (rc == SC_OK && hipStreamSynchronize(ctx->stream) != hipSuccess)
        rc = sc_set_error(ctx, SC_ERR_HIP, "eigensolve failed");
    }
  }
  if (rc != SC_OK) { sc_modes_destroy(md); return rc; }
  *out = md;
  return SC_OK;
}

void sc_modes_destroy(sc_modes* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipFree(m->d_w);
  (void)hipFree(m->d_v);
  delete m;
}

int64_t sc_modes_order(const sc_modes* m) { return m ? m->n : -1; }

int sc_modes_get(sc_modes* m, double* w, double* v) {
  if (!m) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = m->ctx;
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (m->n == 0) return SC_OK;
  if (w) SC_HIP(ctx, hipMemcpyAsync(w, m->d_w, (size_t)m->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (v) SC_HIP(ctx, hipMemcpyAsync(v, m->d_v, (size_t)m->n * m->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_modes_msf(sc_modes* m, const int64_t* mode_idx, int64_t k, double* out) {
  if (!m) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = m->ctx;
  if (k < 0 || (k > 0 && !mode_idx) || (m->n > 0 && !out)) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (m->n == 0) return SC_OK;
  const size_t N = (size_t)(m->n / m->dim);
  const size_t work = modes_scratch_bytes(m->n, m->dim, k, 0);
  SC_TRY(sc_reserve_scratch(ctx, work + align_up((size_t)k * 4, 256) + align_up(N * 8, 256) + 1024));
  Bump bump{(char*)ctx->scratch};
  int* d_sel = bump.take<int>((size_t)std::max<int64_t>(k, 1));
  double* d_out = bump.take<double>(N);
  char* scratch = bump.take<char>(work);
  SC_TRY(stage_mode_list(m, mode_idx, k, d_sel));
  SC_TRY(modes_msf_device(ctx, m->d_v, m->d_w, m->n, m->dim, d_sel, k, scratch, d_out));
  SC_HIP(ctx, hipMemcpyAsync(out, d_out, N * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_modes_dcc(sc_modes* m, const int64_t* mode_idx, int64_t k, int norm, double* out) {
  if (!m) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = m->ctx;
  if (k < 0 || (k > 0 && !mode_idx) || (m->n > 0 && !out)) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (m->n == 0) return SC_OK;
  const size_t N = (size_t)(m->n / m->dim);
  const size_t work = modes_scratch_bytes(m->n, m->dim, k, 1);
  SC_TRY(sc_reserve_scratch(ctx, work + align_up((size_t)k * 4, 256) + align_up(N * N * 8, 256) + 1024));
  Bump bump{(char*)ctx->scratch};
  int* d_sel = bump.take<int>((size_t)std::max<int64_t>(k, 1));
  double* d_out = bump.take<double>(N * N);
  char* scratch = bump.take<char>(work);
  SC_TRY(stage_mode_list(m, mode_idx, k, d_sel));
  SC_TRY(modes_dcc_device(ctx, m->d_v, m->d_w, m->n, m->dim, d_sel, k, norm, scratch, d_out));
  SC_HIP(ctx, hipMemcpyAsync(out, d_out, N * N * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

int sc_modes_prs(sc_modes* m, double rcond, int norm, double* out) {
  if (!m) return SC_ERR_INVALID_ARG;
  sc_ctx* ctx = m->ctx;
  if (m->dim != 3) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "perturbation response scanning needs an ANM (dim 3)");
  if (m->n > 0 && !out) return sc_set_error(ctx, SC_ERR_INVALID_ARG, "bad arguments");
  SC_HIP(ctx, hipSetDevice(ctx->device));
  if (m->n == 0) return SC_OK;
  const size_t N = (size_t)(m->n / 3);
  const size_t work = modes_scratch_bytes(m->n, 3, m->n, 2);
  SC_TRY(sc_reserve_scratch(ctx, work + align_up(N * N * 8, 256) + 1024));
  Bump bump{(char*)ctx->scratch};
  double* d_out = bump.take<double>(N * N);
  char* scratch = bump.take<char>(work);
  SC_TRY(modes_prs_device(ctx, m->d_v, m->d_w, m->n, rcond, norm, scratch, d_out));
  SC_HIP(ctx, hipMemcpyAsync(out, d_out, N * N * 8, hipMemcpyDeviceToHost, ctx->stream));
  SC_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SC_OK;
}

}  // extern "C"
