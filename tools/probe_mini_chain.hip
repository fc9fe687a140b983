// How fast can ONE wave (and two waves per SIMD) issue the MFMA pattern of k_bt2_apply's mini-diamonds (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mini_chain.hip -o /tmp/probe_mini_chain && /tmp/probe_mini_chain
// Per mini: 20 MFMAs  W += A Z[rt][r]  accumulating into NCH accumulators that take turns (a dot-product chain), the
// accumulators summed, then 20 MFMAs  Z[rt] += A W[r]  over five row tiles that take turns.  No memory traffic in the
// loop: this isolates the dependent-issue behaviour of v_mfma_f64_16x16x4_f64 (64 cycles per instruction when the pipe
// is kept full).  Variants: NCH = 1, 2, 4; product 2 in the order (r outer, rt inner) as the kernel has it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NCH, bool P2FIRSTHALF>
__global__ __launch_bounds__(256, 2) void k_chain(const double* __restrict__ src, double* out, unsigned long long* cyc, int iters) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  d4 z[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) z[t] = d4{src[gid & 1023], src[(gid + t) & 1023], 0.5, 0.25};
  double a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = src[(gid + 7 * j) & 1023] * 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 3; st >= 0; --st) {
      d4 w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) w[c] = d4{0, 0, 0, 0};
#pragma unroll
      for (int p = 0; p < 20; ++p) {
        const int rt = st + p / 4, r = p % 4;
        w[p % NCH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p & 7], z[rt][r], w[p % NCH], 0, 0, 0);
      }
      d4 ww = w[0];
#pragma unroll
      for (int c = 1; c < NCH; ++c) ww = ww + w[c];
#pragma unroll
      for (int p = 0; p < 20; ++p) {
        const int r = p / 5, rt = st + p % 5;
        z[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(p + 3) & 7], ww[r], z[rt], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) s += z[t][0] + z[t][1] + z[t][2] + z[t][3];
  out[gid] = s;
  if ((threadIdx.x & 63) == 0) cyc[gid >> 6] = t1 - t0;
}

template <int NCH>
void run(const double* d_src, double* d_out, unsigned long long* d_cyc, int waves_per_simd) {
  const int iters = 200;
  const int threads = 256, blocks = 256 * waves_per_simd;   // one or two 4-wave workgroups per CU
  hipLaunchKernelGGL((k_chain<NCH, false>), dim3(blocks), dim3(threads), 0, 0, d_src, d_out, d_cyc, iters);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_chain<NCH, false>), dim3(blocks), dim3(threads), 0, 0, d_src, d_out, d_cyc, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)blocks * 4);
  CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= (double)h.size();
  const double mfmas_per_wave = 160.0 * iters;
  printf("chains %d  waves/SIMD %d: %8.1f cycles per MFMA per wave (s_memtime ticks x 24 = shader cycles at 2.4 GHz / 100 MHz? raw %.1f)  %.1f TFLOP/s\n",
         NCH, waves_per_simd, mean / mfmas_per_wave, mean / mfmas_per_wave,
         2048.0 * mfmas_per_wave * blocks * 4 / (ms * 1e-3) / 1e12);
}

int main() {
  double *d_src, *d_out;
  unsigned long long* d_cyc;
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0.001 * (i % 37) - 0.01;
  CK(hipMalloc(&d_src, 8192));
  CK(hipMalloc(&d_out, 8 * 256 * 512));
  CK(hipMalloc(&d_cyc, 8 * 4 * 512));
  CK(hipMemcpy(d_src, h.data(), 8192, hipMemcpyHostToDevice));
  for (int wps : {1, 2}) {
    run<1>(d_src, d_out, d_cyc, wps);
    run<2>(d_src, d_out, d_cyc, wps);
    run<4>(d_src, d_out, d_cyc, wps);
  }
  return 0;
}
