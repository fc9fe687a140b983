// f64 MFMA / VALU ceiling with the clock the chip actually holds under load (gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_clock.hip -o /tmp/probe_clock && /tmp/probe_clock
// Every loop is stamped with s_memtime (shader cycles) and s_memrealtime (100 MHz) by every wave, after >= 2 s of
// back-to-back launches of the same kernel on the same data (MI355X_MICROARCH.md, "DVFS give-back" item 6); the
// in-kernel clock is d(memtime) / d(memrealtime) * 100 MHz (median over waves) and cycles per instruction are
// d(memtime) / instructions per SIMD.  The stamps go to a buffer of their own.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long c0, r0, c1, r1; };

__device__ __forceinline__ void stamp(unsigned long long& c, unsigned long long& r) {
  c = __builtin_amdgcn_s_memtime();
  r = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
}

// operands come from memory (random or zero data, decided by the host)
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma16(const double* __restrict__ src, double* out, Stamp* st, int iters) {
  d4 acc[NACC];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const double a = src[gid & 65535], b = src[(gid + 12345) & 65535];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  unsigned long long c0, r0, c1, r1;
  stamp(c0, r0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  stamp(c1, r1);
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[gid] = s;
  if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1};
}

// v_mfma_f64_4x4x4_4b_f64: four 4x4x4 blocks, one result per lane (512 flop per instruction)
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma4(const double* __restrict__ src, double* out, Stamp* st, int iters) {
  double acc[NACC];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const double a = src[gid & 65535], b = src[(gid + 12345) & 65535];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
  unsigned long long c0, r0, c1, r1;
  stamp(c0, r0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  stamp(c1, r1);
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[gid] = s;
  if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1};
}

// 16 independent f64 FMA chains per lane
__global__ void __launch_bounds__(256) k_fma(const double* __restrict__ src, double* out, Stamp* st, int iters) {
  double x[16];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const double m = 1.0 + 1e-7 * src[gid & 65535], c = 1e-9 * src[(gid + 999) & 65535];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = src[(gid + 17 * i) & 65535];
  unsigned long long c0, r0, c1, r1;
  stamp(c0, r0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fma(x[i], m, c);
  }
  stamp(c1, r1);
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[gid] = s;
  if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1};
}

// MFMA waves and VALU-FMA waves side by side on every SIMD (waves 0-3 of a 512-thread block: MFMA, 4-7: FMA)
__global__ void __launch_bounds__(512) k_mixed(const double* __restrict__ src, double* out, Stamp* st, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c0, r0, c1, r1;
  double s = 0;
  if (threadIdx.x < 256) {
    d4 acc[8];
    const double a = src[gid & 65535], b = src[(gid + 12345) & 65535];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    stamp(c0, r0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    stamp(c1, r1);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x[16];
    const double m = 1.0 + 1e-7 * src[gid & 65535], c = 1e-9 * src[(gid + 999) & 65535];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = src[(gid + 17 * i) & 65535];
    stamp(c0, r0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = fma(x[i], m, c);
    }
    stamp(c1, r1);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
  }
  out[gid] = s;
  if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1};
}

struct Result { double ms, clock_ghz, cyc; };

template <class Launch>
static Result run(Launch launch, Stamp* d_st, int nwaves, double soak_s, int first_wave = 0, int wave_stride = 1) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // soak: back-to-back launches for soak_s seconds so that the chip sits at the clock it holds under this load
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float one = 0; CK(hipEventElapsedTime(&one, e0, e1));
  const int reps = std::max(1, (int)(soak_s * 1e3 / std::max(one, 0.01f)));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> h(nwaves);
  CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * nwaves, hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (int w = first_wave; w < nwaves; w += wave_stride) {
    const double dc = (double)(h[w].c1 - h[w].c0), dr = (double)(h[w].r1 - h[w].r0);
    if (dr > 0) { clk.push_back(dc / dr * 0.1); cyc.push_back(dc); }
  }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return Result{ms, clk[clk.size() / 2], cyc[cyc.size() / 2]};
}

int main(int argc, char** argv) {
  const double soak = argc > 1 ? atof(argv[1]) : 2.0;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device: %s  CUs=%d  clockRate=%d kHz   soak %.1f s per line\n", p.name, cus, p.clockRate, soak);
  double *d_rand, *d_zero, *d_out; Stamp* d_st;
  std::vector<double> hr(65536);
  srand(1);
  for (auto& v : hr) v = 2.0 * rand() / RAND_MAX - 1.0;
  CK(hipMalloc(&d_rand, 65536 * 8)); CK(hipMalloc(&d_zero, 65536 * 8));
  CK(hipMemcpy(d_rand, hr.data(), 65536 * 8, hipMemcpyHostToDevice));
  CK(hipMemset(d_zero, 0, 65536 * 8));
  const int max_threads = cus * 8 * 256;
  CK(hipMalloc(&d_out, (size_t)max_threads * 8));
  CK(hipMalloc(&d_st, sizeof(Stamp) * (max_threads / 64)));
  const int iters = 20000;

  printf("\n# v_mfma_f64_16x16x4_f64 (2048 flop), operands in registers; cyc = shader cycles per MFMA per SIMD\n");
  for (const char* data : {"random", "zero"}) {
    const double* src = data[0] == 'r' ? d_rand : d_zero;
    for (int wps : {1, 2, 4}) {
      for (int nacc : {4, 8}) {
        const int blocks = cus * wps;
        auto launch = [&]() {
          if (nacc == 4) hipLaunchKernelGGL(k_mfma16<4>, dim3(blocks), dim3(256), 0, 0, src, d_out, d_st, iters);
          else hipLaunchKernelGGL(k_mfma16<8>, dim3(blocks), dim3(256), 0, 0, src, d_out, d_st, iters);
        };
        const Result r = run(launch, d_st, blocks * 4, soak);
        const double n_per_simd = (double)iters * nacc * wps;
        const double flops = (double)blocks * 4 * iters * nacc * 2048.0;
        printf("mfma16 %-6s waves/SIMD=%d nacc=%d  %8.3f ms  %6.1f TFLOP/s  clock %.3f GHz  %.1f cyc/mfma/SIMD\n", data,
               wps, nacc, r.ms, flops / r.ms * 1e-9, r.clock_ghz, r.cyc * wps / n_per_simd);
      }
    }
  }
  printf("\n# v_mfma_f64_4x4x4_4b_f64 (512 flop)\n");
  for (int wps : {1, 2, 4}) {
    const int blocks = cus * wps;
    auto launch = [&]() { hipLaunchKernelGGL(k_mfma4<8>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); };
    const Result r = run(launch, d_st, blocks * 4, soak);
    const double flops = (double)blocks * 4 * iters * 8 * 512.0;
    printf("mfma4  random waves/SIMD=%d nacc=8  %8.3f ms  %6.1f TFLOP/s  clock %.3f GHz  %.1f cyc/mfma/SIMD\n", wps, r.ms,
           flops / r.ms * 1e-9, r.clock_ghz, r.cyc * wps / ((double)iters * 8 * wps));
  }
  printf("\n# v_fma_f64, 16 independent chains per lane (128 flop per wave-instruction)\n");
  for (const char* data : {"random", "zero"}) {
    const double* src = data[0] == 'r' ? d_rand : d_zero;
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = cus * wps;
      auto launch = [&]() { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, src, d_out, d_st, iters); };
      const Result r = run(launch, d_st, blocks * 4, soak);
      const double flops = (double)blocks * 256 * iters * 16 * 2.0;
      printf("fma    %-6s waves/SIMD=%d          %8.3f ms  %6.1f TFLOP/s  clock %.3f GHz  %.2f cyc/fma/SIMD\n", data, wps,
             r.ms, flops / r.ms * 1e-9, r.clock_ghz, r.cyc * wps / ((double)iters * 16 * wps));
    }
  }
  printf("\n# MFMA waves (nacc 8) and FMA waves on the same SIMDs (1 + 1 and 2 + 2 waves per SIMD)\n");
  for (int wps : {1, 2}) {
    const int blocks = cus * wps;
    auto launch = [&]() { hipLaunchKernelGGL(k_mixed, dim3(blocks), dim3(512), 0, 0, d_rand, d_out, d_st, iters); };
    // waves 0-3 of each block are MFMA, 4-7 FMA: the MFMA waves end later or earlier; report both medians
    Result rm{}, rf{};
    {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float one; CK(hipEventElapsedTime(&one, e0, e1));
      const int reps = std::max(1, (int)(soak * 1e3 / one));
      for (int r = 0; r < reps; ++r) launch();
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<Stamp> h(blocks * 8);
      CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * h.size(), hipMemcpyDeviceToHost));
      std::vector<double> cm, cf, km;
      for (size_t w = 0; w < h.size(); ++w) {
        const double dc = (double)(h[w].c1 - h[w].c0), dr = (double)(h[w].r1 - h[w].r0);
        if ((w & 7) < 4) { cm.push_back(dc); km.push_back(dc / dr * 0.1); } else cf.push_back(dc);
      }
      std::sort(cm.begin(), cm.end()); std::sort(cf.begin(), cf.end()); std::sort(km.begin(), km.end());
      rm = Result{ms, km[km.size() / 2], cm[cm.size() / 2]};
      rf = Result{ms, 0, cf[cf.size() / 2]};
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    const double f_m = (double)blocks * 4 * iters * 8 * 2048.0, f_f = (double)blocks * 256 * iters * 16 * 2.0;
    const double t_m = rm.cyc / (rm.clock_ghz * 1e9), t_f = rf.cyc / (rm.clock_ghz * 1e9);
    printf("mixed  waves/SIMD=%d+%d  kernel %8.3f ms  clock %.3f GHz  mfma part: %.1f cyc/mfma/SIMD, %.1f TFLOP/s while it ran; "
           "fma part: %.2f cyc/fma/SIMD, %.1f TFLOP/s while it ran\n", wps, wps, rm.ms, rm.clock_ghz,
           rm.cyc / ((double)iters * 8), f_m / t_m * 1e-12, rf.cyc / ((double)iters * 16), f_f / t_f * 1e-12);
  }
  return 0;
}
