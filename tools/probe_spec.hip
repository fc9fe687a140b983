// Follow-up to probe_mix.hip: is the starvation of memory waves next to f64 MFMA waves a per-SIMD effect?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_spec.hip -o /tmp/probe_spec && /tmp/probe_spec
// One workgroup per CU (100 KB of LDS), 4 waves = one per SIMD.  MFMA waves run the k_gemm2-like loop without barriers;
// memory waves read-modify-write a private slice of a large buffer with 32 x 1 KB loads in flight.  Configurations:
//   3+1   waves 0-2 MFMA, wave 3 memory (the memory wave has a SIMD of its own)
//   4+4   8 waves: 4 MFMA + 4 memory, i.e. every SIMD holds one of each (the probe_mix situation inside one workgroup),
//         without and with s_setprio 3 in the memory waves
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
struct Stamp { unsigned long long c0, r0, c1, r1, role; };
__device__ __forceinline__ void stamp(unsigned long long& c, unsigned long long& r) {
  c = __builtin_amdgcn_s_memtime();
  r = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
}

__global__ void __launch_bounds__(512, 1) k_spec(const double* __restrict__ src, double* out, Stamp* st, int iters, d2* big,
                                                 long long chunk_d2, int passes, int n_mfma_waves, int do_mfma, int do_mem, int prio) {
  extern __shared__ double lds[];
  constexpr int LD = 144;
  double* sa = lds;
  double* sb = lds + 16 * LD;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * LD; i += blockDim.x) { sa[i] = src[(gid + i) & 65535]; sb[i] = src[(gid + 3 * i) & 65535]; }
  __syncthreads();
  unsigned long long c0 = 0, r0 = 0, c1 = 0, r1 = 0;
  const bool mfma_role = wave < n_mfma_waves;
  if (mfma_role) {
    if (!do_mfma) return;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
    const int fr = lane & 15, fk = lane >> 4;
    const double* a_s = sa + fk * LD + (wave & 1) * 64 + fr;
    const double* b_s = sb + fk * LD + ((wave >> 1) & 1) * 64 + fr;
    stamp(c0, r0);
    for (int it = 0; it < iters; ++it) {
      double af[2][4], bf[2][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { af[0][t] = a_s[t * 16]; bf[0][t] = b_s[t * 16]; }
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int cur = k4 & 1, nxt = cur ^ 1;
        if (k4 < 3) {
#pragma unroll
          for (int t = 0; t < 4; ++t) { af[nxt][t] = a_s[(k4 + 1) * 4 * LD + t * 16]; bf[nxt][t] = b_s[(k4 + 1) * 4 * LD + t * 16]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[cur][i], af[cur][j], acc[i][j], 0, 0, 0);
      }
    }
    stamp(c1, r1);
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[gid] = s;
  } else {
    if (!do_mem) return;
    const int n_mem_waves = (blockDim.x >> 6) - n_mfma_waves;
    const long long w = (long long)blockIdx.x * n_mem_waves + (wave - n_mfma_waves);
    d2* base = big + w * chunk_d2;
    if (prio) __builtin_amdgcn_s_setprio(3);   // the SIMD's arbiter serves this wave first whenever it has an instruction ready
    stamp(c0, r0);
    for (int p = 0; p < passes; ++p)
      for (long long off = 0; off + 32 * 64 <= chunk_d2; off += 32 * 64) {
        d2 v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = base[off + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < 32; ++u) base[off + u * 64 + lane] = v[u] + 1.0;
      }
    stamp(c1, r1);
  }
  if (lane == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1, mfma_role ? 1ull : 2ull};
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double *d_rand, *d_out; Stamp* d_st; d2* big;
  std::vector<double> hr(65536);
  srand(1);
  for (auto& v : hr) v = 2.0 * rand() / RAND_MAX - 1.0;
  CK(hipMalloc(&d_rand, 65536 * 8)); CK(hipMemcpy(d_rand, hr.data(), 65536 * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_out, (size_t)cus * 512 * 8));
  CK(hipMalloc(&d_st, sizeof(Stamp) * cus * 8));
  const long long total_d2 = (8ll << 30) / 16;   // 8 GiB buffer
  CK(hipMalloc(&big, (size_t)total_d2 * 16)); CK(hipMemset(big, 0, (size_t)total_d2 * 16));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spec), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const int iters = 4000, passes = 2;
  printf("device: %s  CUs=%d; one workgroup per CU; MFMA waves: %d k-steps x 64 MFMAs; memory waves: 2 passes of read-modify-write over their share of 8 GiB\n", p.name, cus, iters);
  struct Cfg { const char* name; int threads, n_mfma, prio; };
  const Cfg cfgs[3] = {{"3+1 (memory wave on its own SIMD)", 256, 3, 0}, {"4+4 (every SIMD holds one of each)", 512, 4, 0},
                       {"4+4, memory waves at s_setprio 3", 512, 4, 1}};
  for (const Cfg& c : cfgs)
    for (int mode = 0; mode < 3; ++mode) {
      const int do_mfma = mode != 1, do_mem = mode != 0;
      const int n_mem = c.threads / 64 - c.n_mfma;
      const long long chunk_d2 = total_d2 / ((long long)cus * n_mem) / (32 * 64) * (32 * 64);
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d_st, 0, sizeof(Stamp) * cus * 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_spec, dim3(cus), dim3(c.threads), 100 * 1024, 0, d_rand, d_out, d_st, iters, big, chunk_d2, passes,
                           c.n_mfma, do_mfma, do_mem, c.prio);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep < 2) continue;
        std::vector<Stamp> h(cus * 8);
        CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * cus * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk, mem_s;
        for (auto& x : h) {
          const double dc = (double)(x.c1 - x.c0), dr = (double)(x.r1 - x.r0);
          if (dr <= 0) continue;
          if (x.role == 1) { cyc.push_back(dc / (iters * 64.0)); clk.push_back(dc / dr * 0.1); }
          else if (x.role == 2) mem_s.push_back(dr * 1e-8);
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        const double t_mem = med(mem_s), mc = med(cyc), g = med(clk);
        printf("%-36s %-12s kernel %8.3f ms", c.name, mode == 0 ? "MFMA only" : (mode == 1 ? "memory only" : "both"), ms);
        if (do_mfma) printf("   MFMA: %.1f cycles per MFMA at %.3f GHz = %.1f TFLOP/s from %d waves per CU", mc, g,
                            (double)cus * c.n_mfma * 2048.0 * g / mc * 1e-3, c.n_mfma);
        if (do_mem) printf("   memory: median wave %.3f ms -> %.2f TB/s (read + write)", t_mem * 1e3,
                           (double)cus * n_mem * chunk_d2 * 16.0 * passes * 2.0 / t_mem * 1e-12);
        printf("\n");
      }
    }
  return 0;
}
