// Host-only logic of the C ABI (springcraft_amd/csrc/host_logic.h) under AddressSanitizer + UBSan:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I include -I springcraft_amd/csrc \
//       tests/host_sanitize/test_host_logic.cpp -o /tmp/test_host_logic && /tmp/test_host_logic
// (built and run by tests/test_abi_and_host.py::test_host_logic_under_sanitizers).  Expected values restate
// _patch_adjacency_matrix (interaction.py:193-213) and PatchedForceField.force_constant (forcefield.py:199-224).
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "host_logic.h"

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); std::exit(1); } } while (0)

static sc_patch_desc desc(const std::vector<int64_t>& shut, const std::vector<int64_t>& off, const std::vector<int64_t>& on,
                          const std::vector<double>* gam) {
  sc_patch_desc d{};
  d.n_shutdown = (int64_t)shut.size(); d.shutdown = shut.empty() ? nullptr : shut.data();
  d.n_pair_off = (int64_t)off.size() / 2; d.pair_off = off.empty() ? nullptr : off.data();
  d.n_pair_on = (int64_t)on.size() / 2; d.pair_on = on.empty() ? nullptr : on.data();
  d.on_force_constants = gam ? gam->data() : nullptr;
  return d;
}

int main() {
  using sc_host::HostPatch;
  std::string err;
  // ---- no patch at all
  {
    HostPatch hp;
    CHECK(sc_host::build_patch(nullptr, 10, hp, err) == SC_OK && !hp.any);
    sc_patch_desc d{};
    CHECK(sc_host::build_patch(&d, 10, hp, err) == SC_OK && !hp.any);
  }
  // ---- shutdown + off + on with overrides, duplicates and both orientations
  {
    const std::vector<int64_t> shut = {3, 3, 0}, off = {1, 2, 4, 5, 2, 1}, on = {1, 2, 6, 7, 7, 6};
    const std::vector<double> gam = {2.5, 1.0, 9.0};
    sc_patch_desc d = desc(shut, off, on, &gam);
    HostPatch hp;
    CHECK(sc_host::build_patch(&d, 8, hp, err) == SC_OK && hp.any);
    CHECK(hp.shut.size() == 8 && hp.shut[0] == 1 && hp.shut[3] == 1 && hp.shut[1] == 0);
    CHECK(hp.row_ptr.size() == 9 && hp.row_ptr[8] == (int32_t)hp.col.size());
    auto find = [&](int i, int j) -> int {
      for (int p = hp.row_ptr[i]; p < hp.row_ptr[i + 1]; ++p) if (hp.col[p] == j) return p;
      return -1;
    };
    // pair_on beats pair_off (applied last), symmetric
    CHECK(find(1, 2) >= 0 && hp.flag[find(1, 2)] == 1 && hp.gam[find(1, 2)] == 2.5);
    CHECK(find(2, 1) >= 0 && hp.flag[find(2, 1)] == 1 && hp.gam[find(2, 1)] == 2.5);
    CHECK(find(4, 5) >= 0 && hp.flag[find(4, 5)] == 0 && find(5, 4) >= 0 && hp.flag[find(5, 4)] == 0);
    // numpy assignment order: m[i, j] = v for all rows, then m[j, i] = v for all rows -> (6,7),(7,6) listed twice:
    // first pass writes m[6,7] = 1.0 then m[7,6] = 9.0, second pass m[7,6] = 1.0 then m[6,7] = 9.0
    CHECK(hp.gam[find(6, 7)] == 9.0 && hp.gam[find(7, 6)] == 1.0);
    for (int i = 0; i < 8; ++i)
      for (int p = hp.row_ptr[i] + 1; p < hp.row_ptr[i + 1]; ++p) CHECK(hp.col[p - 1] < hp.col[p]);   // CSR rows sorted
    CHECK(hp.device_bytes() >= hp.shut.size() + hp.col.size() * 4);
  }
  // ---- pair_on without force constants: NaN marks "use the base force field" (forcefield.py:221-224)
  {
    const std::vector<int64_t> on = {0, 1};
    sc_patch_desc d = desc({}, {}, on, nullptr);
    HostPatch hp;
    CHECK(sc_host::build_patch(&d, 2, hp, err) == SC_OK);
    CHECK(hp.col.size() == 2 && std::isnan(hp.gam[0]) && std::isnan(hp.gam[1]) && hp.flag[0] == 1);
  }
  // ---- errors: out of range (either column, negative), self pair, inconsistent descriptor
  {
    HostPatch hp;
    const std::vector<int64_t> bad_shut = {8};
    sc_patch_desc d = desc(bad_shut, {}, {}, nullptr);
    CHECK(sc_host::build_patch(&d, 8, hp, err) == SC_ERR_INDEX && err.find("Index 8 is out of bounds") != std::string::npos);
    const std::vector<int64_t> bad_off = {1, -1};
    d = desc({}, bad_off, {}, nullptr);
    CHECK(sc_host::build_patch(&d, 8, hp, err) == SC_ERR_INDEX && err.find("Index -1") != std::string::npos);
    const std::vector<int64_t> bad_on = {9, 1};
    d = desc({}, {}, bad_on, nullptr);
    CHECK(sc_host::build_patch(&d, 8, hp, err) == SC_ERR_INDEX && err.find("Index 9") != std::string::npos);
    const std::vector<int64_t> self = {2, 2};
    d = desc({}, {}, self, nullptr);
    CHECK(sc_host::build_patch(&d, 8, hp, err) == SC_ERR_SELF_PAIR);
    sc_patch_desc e{};
    e.n_pair_on = 3;   // count without pointer
    CHECK(sc_host::build_patch(&e, 8, hp, err) == SC_ERR_INVALID_ARG);
    e = sc_patch_desc{};
    e.n_shutdown = -1;
    CHECK(sc_host::build_patch(&e, 8, hp, err) == SC_ERR_INVALID_ARG);
  }
  // ---- a large random patch: every override lands in its row, nothing out of bounds (ASan / UBSan watch the rest)
  {
    const int64_t n = 5000;
    std::vector<int64_t> shut, off, on;
    std::vector<double> gam;
    unsigned long long s = 12345;
    auto rnd = [&](int64_t m) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (int64_t)((s >> 33) % (unsigned long long)m); };
    for (int i = 0; i < 300; ++i) shut.push_back(rnd(n));
    for (int i = 0; i < 20000; ++i) { off.push_back(rnd(n)); off.push_back(rnd(n)); }
    for (int i = 0; i < 20000; ++i) { int64_t a = rnd(n), b = rnd(n); if (a == b) b = (b + 1) % n; on.push_back(a); on.push_back(b); gam.push_back((double)i); }
    sc_patch_desc d = desc(shut, off, on, &gam);
    HostPatch hp;
    CHECK(sc_host::build_patch(&d, n, hp, err) == SC_OK);
    CHECK((int64_t)hp.row_ptr.size() == n + 1 && hp.row_ptr[0] == 0);
    for (int64_t i = 0; i < n; ++i) {
      CHECK(hp.row_ptr[i] <= hp.row_ptr[i + 1]);
      for (int p = hp.row_ptr[i]; p < hp.row_ptr[i + 1]; ++p) CHECK(hp.col[p] >= 0 && hp.col[p] < n);
    }
    CHECK(hp.col.size() == hp.flag.size() && hp.col.size() == hp.gam.size());
  }
  // ---- force-field descriptor checks
  {
    CHECK(sc_host::check_ff(nullptr, err) == SC_ERR_INVALID_ARG);
    sc_ff_desc f{};
    f.kind = SC_FF_INVARIANT; f.has_cutoff = 0;
    CHECK(sc_host::check_ff(&f, err) == SC_ERR_INVALID_ARG && err == "Cutoff distance must be a float");
    f.has_cutoff = 1; f.cutoff = 7.0; f.cutoff_sq = 49.0;
    CHECK(sc_host::check_ff(&f, err) == SC_OK);
    f.kind = 17;
    CHECK(sc_host::check_ff(&f, err) == SC_ERR_INVALID_ARG);
    f.kind = SC_FF_TABULATED; f.tab = nullptr;
    CHECK(sc_host::check_ff(&f, err) == SC_ERR_INVALID_ARG);
  }
  // ---- launch shape of k_sytrd_resident (one matrix resident on the chip: tridiag.hip)
  {
    using sc_host::ResidentShape;
    ResidentShape S{};
    CHECK(!sc_host::resident_shape(127, 64, 256, 0, 0, &S));                      // too small
    CHECK(sc_host::resident_shape(128, 64, 256, 0, 0, &S) && S.off == 0 && S.m == 128 && S.P == 32 && S.Q == 1 && !S.reg);
    CHECK(sc_host::resident_shape(300, 64, 256, 0, 0, &S) && S.P == 64 && S.logP == 6 && S.Q == 2);
    CHECK(sc_host::resident_shape(300, 64, 256, 0, 256, &S) && S.P == 256 && S.logP == 8);      // more workgroups on request
    CHECK(sc_host::resident_shape(1536, 64, 256, 0, 0, &S) && S.P == 256 && S.Q == 6 && !S.reg &&
          S.lds_bytes == 8 * (size_t)(6 * 1536 + sc_host::kResidentSmallDoubles));
    CHECK(sc_host::resident_shape(2048, 64, 256, 0, 0, &S) && S.Q == 8 && !S.reg && S.lds_bytes <= 160 * 1024);   // the LDS of a CU
    CHECK(sc_host::resident_shape(2049, 64, 256, 0, 0, &S) && S.reg && S.Q == 10 && S.P == 256 && S.off == 0 &&
          S.lds_bytes == 8 * (size_t)sc_host::kResidentSmallDoubles);
    CHECK(sc_host::resident_shape(2561, 64, 256, 0, 0, &S) && S.reg && S.Q == 12);
    CHECK(sc_host::resident_shape(3072, 64, 256, 0, 0, &S) && S.reg && S.off == 0 && (S.m + S.P - 1) / S.P <= sc_host::kResidentRowsReg);
    // larger: the trailing matrix from the first panel boundary at which it fits
    CHECK(sc_host::resident_shape(3073, 64, 256, 0, 0, &S) && S.off == 64 && S.m == 3009);
    CHECK(sc_host::resident_shape(6000, 64, 256, 0, 0, &S) && S.off == 2944 && S.m == 3056 && S.reg);
    CHECK(sc_host::resident_shape(6000, 64, 256, 2048, 0, &S) && S.off == 3968 && S.m == 2032 && !S.reg);   // SPRINGCRAFT_RESIDENT_MAX
    // a device with fewer CUs has no register form and may not fit the workgroups at all
    CHECK(sc_host::resident_shape(3000, 64, 128, 0, 0, &S) == false || (S.m <= 2048 && S.P <= 128));
    CHECK(sc_host::resident_shape(1000, 64, 128, 0, 0, &S) && S.P == 128);
    CHECK(!sc_host::resident_shape(2000, 64, 128, 0, 0, &S));                      // needs 256 workgroups
    for (int n = 128; n <= 7000; n += 37) {
      if (!sc_host::resident_shape(n, 64, 256, 0, 0, &S)) { CHECK(false); continue; }
      const int rows = (S.m - 1) / S.P + 1;
      CHECK(S.off % 64 == 0 && S.off + S.m == n && S.m <= 3072 && (1 << S.logP) == S.P && S.P <= 256);
      CHECK(rows <= (S.reg ? 12 : 8) && 256 * S.Q >= S.m && S.lds_bytes <= 160 * 1024);
    }
  }
  std::puts("host logic ok");
  return 0;
}
