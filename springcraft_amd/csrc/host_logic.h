// Host-side logic of the C ABI that needs no HIP: the contact-patch override table and the force-field descriptor
// check.  Header-only so that tests/host_sanitize/ can compile exactly this code with g++ -fsanitize=address,undefined
// (sanitizers run on the CPU build only); api.hip includes it and forwards the messages to
// sc_set_error.
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "springcraft_hip.h"

namespace sc_host {

static inline size_t align_up_sz(size_t x, size_t a) { return (x + a - 1) / a * a; }

// status code + message, printf style
static inline int fail(std::string& err, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  err = buf;
  return code;
}

static inline int check_ff(const sc_ff_desc* ff, std::string& err) {
  if (!ff) return fail(err, SC_ERR_INVALID_ARG, "force-field descriptor is NULL");
  if (ff->kind < SC_FF_INVARIANT || ff->kind > SC_FF_TABULATED)
    return fail(err, SC_ERR_INVALID_ARG, "unknown force-field kind %d", ff->kind);
  if (ff->kind == SC_FF_TABULATED) {
    const sc_tab_desc* t = ff->tab;
    if (!t || t->n_bins < 1 || !t->bonded || !t->intra_chain || !t->inter_chain || !t->atom_type || !t->chain ||
        !t->bonded_next || (t->n_bins > 1 && !t->edges_sq))
      return fail(err, SC_ERR_INVALID_ARG, "incomplete tabulated force-field descriptor");
  }
  if (ff->kind == SC_FF_INVARIANT && !ff->has_cutoff)
    return fail(err, SC_ERR_INVALID_ARG, "Cutoff distance must be a float");  // forcefield.py:277-281
  return SC_OK;
}

struct HostPatch {
  bool any = false;
  std::vector<uint8_t> shut;
  std::vector<int32_t> row_ptr, col;
  std::vector<int8_t> flag;
  std::vector<double> gam;
  int mask_gamma = 0;
  size_t device_bytes() const {
    return align_up_sz(shut.size(), 256) + align_up_sz(row_ptr.size() * 4, 256) +
           align_up_sz(col.size() * 4 + 4, 256) + align_up_sz(flag.size() + 1, 256) +
           align_up_sz(gam.size() * 8 + 8, 256) + 2048;
  }
};

// Restates _patch_adjacency_matrix (interaction.py:193-213) + the patch-matrix construction of
// PatchedForceField.force_constant (forcefield.py:199-224) as a per-row override table.
static inline int build_patch(const sc_patch_desc* pd, int64_t n, HostPatch& hp, std::string& err) {
  hp.any = false;
  if (!pd || (pd->n_shutdown == 0 && pd->n_pair_off == 0 && pd->n_pair_on == 0)) return SC_OK;
  if (pd->n_shutdown < 0 || pd->n_pair_off < 0 || pd->n_pair_on < 0 ||
      (pd->n_shutdown && !pd->shutdown) || (pd->n_pair_off && !pd->pair_off) ||
      (pd->n_pair_on && !pd->pair_on))
    return fail(err, SC_ERR_INVALID_ARG, "inconsistent patch descriptor");
  hp.any = true;
  hp.mask_gamma = pd->base_cutoff_masks_gamma;
  hp.shut.assign((size_t)n, 0);
  auto in_range = [&](int64_t v) { return v >= 0 && v < n; };
  for (int64_t s = 0; s < pd->n_shutdown; ++s) {
    if (!in_range(pd->shutdown[s]))
      return fail(err, SC_ERR_INDEX, "Index %lld is out of bounds for a structure of length %lld",
                          (long long)pd->shutdown[s], (long long)n);
    hp.shut[(size_t)pd->shutdown[s]] = 1;
  }
  struct Ov { int8_t flag; double gam; };
  std::map<std::pair<int32_t, int32_t>, Ov> ov;
  auto touch = [&](int64_t i, int64_t j) -> Ov& {
    auto key = std::make_pair((int32_t)i, (int32_t)j);
    auto it = ov.find(key);
    if (it == ov.end()) it = ov.emplace(key, Ov{0, -1.0}).first;
    return it->second;
  };
  for (int64_t p = 0; p < pd->n_pair_off; ++p) {
    const int64_t i = pd->pair_off[2 * p], j = pd->pair_off[2 * p + 1];
    if (!in_range(i) || !in_range(j))
      return fail(err, SC_ERR_INDEX, "Index %lld is out of bounds for a structure of length %lld",
                          (long long)(in_range(i) ? j : i), (long long)n);
    touch(i, j).flag = 0;
    touch(j, i).flag = 0;
  }
  for (int64_t p = 0; p < pd->n_pair_on; ++p) {
    const int64_t i = pd->pair_on[2 * p], j = pd->pair_on[2 * p + 1];
    if (!in_range(i) || !in_range(j))
      return fail(err, SC_ERR_INDEX, "Index %lld is out of bounds for a structure of length %lld",
                          (long long)(in_range(i) ? j : i), (long long)n);
    if (i == j)
      return fail(err, SC_ERR_SELF_PAIR, "Cannot turn on interaction of an atom with itself");
  }
  // numpy assignment order: matrix[i, j] = v for all rows, then matrix[j, i] = v for all rows
  for (int pass = 0; pass < 2; ++pass)
    for (int64_t p = 0; p < pd->n_pair_on; ++p) {
      const int64_t i = pd->pair_on[2 * p + pass], j = pd->pair_on[2 * p + 1 - pass];
      Ov& o = touch(i, j);
      o.flag = 1;
      if (pd->on_force_constants) o.gam = pd->on_force_constants[p];
    }
  hp.row_ptr.assign((size_t)n + 1, 0);
  for (auto& kv : ov) hp.row_ptr[(size_t)kv.first.first + 1]++;
  for (int64_t i = 0; i < n; ++i) hp.row_ptr[i + 1] += hp.row_ptr[i];
  hp.col.reserve(ov.size());
  hp.flag.reserve(ov.size());
  hp.gam.reserve(ov.size());
  for (auto& kv : ov) {  // std::map iterates sorted by (i, j): already CSR order
    hp.col.push_back(kv.first.second);
    hp.flag.push_back(kv.second.flag);
    hp.gam.push_back(kv.second.gam == -1.0 ? std::nan("") : kv.second.gam);  // forcefield.py:221-224
  }
  return SC_OK;
}


// ---- k_sytrd_resident (tridiag.hip): which part of a one-matrix, one-stage solve it takes and with what launch shape.
// Orders up to kResidentMaxLds keep the rows of the matrix in LDS (<= 8 rows of <= 2048 doubles per workgroup), up to
// kResidentMaxReg in registers (<= 12 rows of 12 x 256 entries per thread; 256 workgroups, so a device with fewer CUs stays
// with the LDS form); a larger matrix hands over its trailing columns at the first panel boundary (nb columns) from which
// the order fits.  want_wgs: 0, or a number of workgroups to raise the minimum to (a power of two, <= 256).
constexpr int kResidentRowsLds = 8, kResidentRowsReg = 12;
constexpr int kResidentMaxLds = 2048, kResidentMaxReg = 3072;
constexpr int kResidentSmallDoubles = 160;     // LDS besides the rows
constexpr int kResidentMinOrder = 128;
struct ResidentShape {
  int off, m;        // the trailing matrix: rows / columns off .. off + m - 1
  int P, logP;       // workgroups (a power of two)
  int Q;             // template parameter: chunks of 256 columns per thread (1, 2, 4, 6, 8 | 10, 12)
  bool reg;          // rows in registers
  size_t lds_bytes;  // dynamic LDS of the launch
};
static inline bool resident_shape(int n, int nb, int cus, int max_order, int want_wgs, ResidentShape* out) {
  if (n < kResidentMinOrder || nb <= 0) return false;
  int max_m = cus >= 256 ? kResidentMaxReg : kResidentMaxLds;
  if (max_order >= kResidentMinOrder && max_order < max_m) max_m = max_order;
  ResidentShape R{};
  R.off = n > max_m ? (n - max_m + nb - 1) / nb * nb : 0;
  R.m = n - R.off;
  if (R.m < kResidentMinOrder) return false;
  R.reg = R.m > kResidentMaxLds;
  const int rows = R.reg ? kResidentRowsReg : kResidentRowsLds;
  int P = 32, lp = 5;
  while ((R.m + P - 1) / P > rows) { P *= 2; ++lp; }
  while (P < want_wgs && 2 * P <= 256) { P *= 2; ++lp; }
  if (P > cus || P > 256) return false;
  R.P = P;
  R.logP = lp;
  const int qn = (R.m + 255) / 256;
  R.Q = R.reg ? (qn <= 10 ? 10 : 12) : (qn <= 2 ? qn : (qn <= 4 ? 4 : (qn <= 6 ? 6 : 8)));
  const size_t rmax = (size_t)(R.m - 1) / P + 1;
  R.lds_bytes = sizeof(double) * ((R.reg ? 0 : rmax * 256 * R.Q) + kResidentSmallDoubles);
  *out = R;
  return true;
}

}  // namespace sc_host
