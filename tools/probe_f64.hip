// Hardware probe for gfx950: f64 MFMA operand layout + issue rate, f64 VALU FMA rate,
// HBM streaming copy rate.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_f64.hip -o probe
// Output is plain text; numbers feed DESIGN.md (roofline peaks) and the GEMM tile design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// ---- layout check: C = A(16x4) * B(4x16) with distinct integer entries -------------
__global__ void k_layout(const double* A, const double* B, double* C) {
  int l = threadIdx.x;             // one wave
  double a = A[(l & 15) * 4 + (l >> 4)];      // A[i=l&15][k=l>>4], row-major 16x4
  double b = B[(l >> 4) * 16 + (l & 15)];     // B[k=l>>4][j=l&15], row-major 4x16
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  // guide: col = lane&15, row = (lane>>4) + 4*reg
  for (int r = 0; r < 4; ++r) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

// ---- MFMA throughput ----------------------------------------------------------------
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma_rate(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- VALU f64 FMA throughput --------------------------------------------------------
__global__ void __launch_bounds__(256) k_fma_rate(double* out, int iters) {
  double x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
  double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = fma(x[i], m, c);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- f64 divide / sqrt rate ----------------------------------------------------------
__global__ void __launch_bounds__(256) k_div_rate(double* out, int iters) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0 + threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 1.5 / x[i] + 0.25;
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- streaming copy / write / read ---------------------------------------------------
__global__ void k_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i];
}
__global__ void k_write(double2* __restrict__ out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = double2{1.0, 2.0};
}
__global__ void k_read(const double2* __restrict__ in, double* out, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n; i += stride) { double2 v = in[i]; s += v.x + v.y; }
  if (s == 123.456) out[0] = s;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  L2=%d  mem=%zu GB\n", p.name, p.multiProcessorCount,
         p.clockRate, p.l2CacheSize, p.totalGlobalMem >> 30);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  // layout
  {
    std::vector<double> A(64), B(64), C(256), R(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 4 + k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 100 + 7 * k + 13 * j + (k * j) % 5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j)
      for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dC;
    CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8));
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dC);
    CK(hipMemcpy(C.data(), dC, 256 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (C[i] != R[i]) ++bad;
    printf("mfma_f64_16x16x4 layout check (A[l&15][l>>4], B[l>>4][l&15], C row=(l>>4)+4r col=l&15): %s (%d mismatches)\n",
           bad ? "FAIL" : "OK", bad);
  }

  double* dout; CK(hipMalloc(&dout, 2048 * 256 * 8 * 4));
  // MFMA rate: blocks = CUs * k, 256 threads (1 wave / SIMD) and 512 threads (2 waves/SIMD)
  for (int wavesPerSimd = 1; wavesPerSimd <= 2; ++wavesPerSimd) {
    int iters = 20000;
    int blocks = p.multiProcessorCount * wavesPerSimd;
    for (int nacc = 1; nacc <= 8; nacc *= 2) {
      auto launch = [&]() {
        if (nacc == 1) k_mfma_rate<1><<<blocks, 256>>>(dout, iters);
        if (nacc == 2) k_mfma_rate<2><<<blocks, 256>>>(dout, iters);
        if (nacc == 4) k_mfma_rate<4><<<blocks, 256>>>(dout, iters);
        if (nacc == 8) k_mfma_rate<8><<<blocks, 256>>>(dout, iters);
      };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1);
      double flops = (double)blocks * 4 * iters * nacc * 2048.0;
      double cyc_per_mfma = ms * 1e-3 * 2.4e9 / ((double)iters * nacc * wavesPerSimd);
      printf("mfma f64 16x16x4: waves/SIMD=%d nacc=%d  %.3f ms  %.1f TFLOP/s  (~%.1f cyc/mfma/SIMD @2.4GHz)\n",
             wavesPerSimd, nacc, ms, flops / ms * 1e-9, cyc_per_mfma);
    }
  }
  // VALU FMA rate
  for (int w = 1; w <= 4; w *= 2) {
    int iters = 20000, blocks = p.multiProcessorCount * w;
    k_fma_rate<<<blocks, 256>>>(dout, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k_fma_rate<<<blocks, 256>>>(dout, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = time_ms(e0, e1);
    double flops = (double)blocks * 256 * iters * 16 * 2.0;
    printf("valu f64 fma: waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", w, ms, flops / ms * 1e-9);
  }
  for (int w = 1; w <= 4; w *= 4) {
    int iters = 4000, blocks = p.multiProcessorCount * w;
    k_div_rate<<<blocks, 256>>>(dout, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k_div_rate<<<blocks, 256>>>(dout, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = time_ms(e0, e1);
    double divs = (double)blocks * 256 * iters * 8;
    printf("valu f64 div: waves/SIMD=%d  %.3f ms  %.2f Tdiv/s\n", w, ms, divs / ms * 1e-9);
  }
  // streaming
  {
    size_t bytes = (size_t)2 << 30;  // 2 GiB each
    double2 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    size_t n = bytes / 16;
    for (int blocks : {2048, 8192}) {
      k_copy<<<blocks, 256>>>(a, b, n); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) k_copy<<<blocks, 256>>>(a, b, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1) / 5;
      printf("copy   2GiB->2GiB blocks=%d: %.3f ms  %.2f TB/s (read+write)\n", blocks, ms, 2.0 * bytes / ms * 1e-9);
      CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) k_write<<<blocks, 256>>>(b, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      ms = time_ms(e0, e1) / 5;
      printf("write  2GiB blocks=%d: %.3f ms  %.2f TB/s\n", blocks, ms, 1.0 * bytes / ms * 1e-9);
      CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) k_read<<<blocks, 256>>>(a, dout, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      ms = time_ms(e0, e1) / 5;
      printf("read   2GiB blocks=%d: %.3f ms  %.2f TB/s\n", blocks, ms, 1.0 * bytes / ms * 1e-9);
    }
    // re-read of a 64 MiB / 144 MiB buffer (MALL resident?)
    for (size_t mb : {16, 64, 144, 288}) {
      size_t nb = mb << 20; size_t nn = nb / 16;
      k_read<<<2048, 256>>>(a, dout, nn); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 20; ++r) k_read<<<2048, 256>>>(a, dout, nn);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = time_ms(e0, e1) / 20;
      printf("re-read %zu MiB: %.4f ms  %.2f TB/s\n", mb, ms, (double)nb / ms * 1e-9);
    }
    // launch overhead: empty-ish kernel chain
    CK(hipEventRecord(e0)); for (int r = 0; r < 1000; ++r) k_write<<<1, 64>>>(b, 64);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    printf("1000 tiny dependent launches: %.3f ms (%.2f us each)\n", time_ms(e0, e1), time_ms(e0, e1));
  }
  return 0;
}
