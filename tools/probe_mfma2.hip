// What keeps v_mfma_f64_16x16x4_f64 from its 64-cycle issue rate in a real loop?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma2.hip -o /tmp/probe_mfma2 && /tmp/probe_mfma2
// All kernels keep their accumulators in VGPRs (__launch_bounds__(256, 2) caps the kernel at 256 registers, so hipcc
// does not move them to AGPRs and back every iteration as it did in tools/probe_f64.hip).  Stamps as in probe_clock.hip.
//   regs      operands in registers, NACC independent accumulators
//   lds_a     A operand re-read from LDS for every MFMA (ds_read_b64, conflict-free), B in registers: the pattern of the
//             stage-2 back-transformation (fragments in LDS, the eigenvector window as the B operand)
//   gemm      per k-step 4 + 4 fragment reads and 16 MFMAs (4 x 4 accumulator tiles): the k_gemm2 inner loop without
//             staging and barriers
//   gemm_bar  the same with a workgroup barrier every 4 k-steps
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

template <int NACC>
__global__ void __launch_bounds__(256, 2) k_regs(const double* __restrict__ src, double* out, Stamp* st, int iters) {
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

// 8 accumulators, A from LDS per MFMA (16 fragments of 512 B walked round and round), B in registers
__global__ void __launch_bounds__(256, 2) k_lds_a(const double* __restrict__ src, double* out, Stamp* st, int iters) {
  __shared__ double frag[16 * 64 * 4];   // 4 waves x 16 fragments
  d4 acc[8];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 64 * 4; i += 256) frag[i] = src[(gid + i) & 65535];
  const double b0 = src[gid & 65535], b1 = src[(gid + 777) & 65535];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  const double* f = frag + wave * 16 * 64 + lane;
  unsigned long long c0, r0, c1, r1;
  stamp(c0, r0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      acc[i & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[i * 64], (i & 1) ? b1 : b0, acc[i & 7], 0, 0, 0);
  }
  stamp(c1, r1);
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[gid] = s;
  if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1};
}

// GEMM inner loop: 16 accumulators, per k-step 4 A + 4 B fragment reads ([k][144] image), fragments of the next k-step
// read while the MFMAs of this one issue
template <bool BARRIER>
__global__ void __launch_bounds__(256, 2) k_gemm(const double* __restrict__ src, double* out, Stamp* st, int iters) {
  constexpr int LD = 144;
  __shared__ double sa[16 * LD], sb[16 * LD];
  d4 acc[4][4];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * LD; i += 256) { sa[i] = src[(gid + i) & 65535]; sb[i] = src[(gid + 3 * i) & 65535]; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
  const int fr = lane & 15, fk = lane >> 4;
  const double* a_s = sa + fk * LD + (wave & 1) * 64 + fr;
  const double* b_s = sb + fk * LD + (wave >> 1) * 64 + fr;
  unsigned long long c0, r0, c1, r1;
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
    if (BARRIER) __syncthreads();
  }
  stamp(c1, r1);
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[gid] = s;
  if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1};
}

template <class Launch>
static void run(const char* name, int wps, double mfma_per_wave, Launch launch, Stamp* d_st, int nwaves, double soak_s) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
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
  for (int w = 0; w < nwaves; ++w) {
    const double dc = (double)(h[w].c1 - h[w].c0), dr = (double)(h[w].r1 - h[w].r0);
    if (dr > 0) { clk.push_back(dc / dr * 0.1); cyc.push_back(dc); }
  }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double c = cyc[cyc.size() / 2], g = clk[clk.size() / 2];
  printf("%-9s waves/SIMD=%d  %8.3f ms  %6.1f TFLOP/s (kernel)  clock %.3f GHz  %.1f cyc per MFMA per SIMD (wave stamps)\n", name,
         wps, ms, (double)nwaves * mfma_per_wave * 2048.0 / ms * 1e-9, g, c / mfma_per_wave / wps);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const double soak = argc > 1 ? atof(argv[1]) : 1.0;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device: %s  CUs=%d   soak %.1f s per line\n", p.name, cus, soak);
  double *d_rand, *d_out; Stamp* d_st;
  std::vector<double> hr(65536);
  srand(1);
  for (auto& v : hr) v = 2.0 * rand() / RAND_MAX - 1.0;
  CK(hipMalloc(&d_rand, 65536 * 8));
  CK(hipMemcpy(d_rand, hr.data(), 65536 * 8, hipMemcpyHostToDevice));
  const int max_threads = cus * 8 * 256;
  CK(hipMalloc(&d_out, (size_t)max_threads * 8));
  CK(hipMalloc(&d_st, sizeof(Stamp) * (max_threads / 64)));
  const int iters = 4000;
  for (int wps : {1, 2, 4}) {
    const int blocks = cus * wps;
    run("regs x1", wps, iters * 1.0, [&]() { hipLaunchKernelGGL(k_regs<1>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
    run("regs x2", wps, iters * 2.0, [&]() { hipLaunchKernelGGL(k_regs<2>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
    run("regs x4", wps, iters * 4.0, [&]() { hipLaunchKernelGGL(k_regs<4>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
    run("regs x8", wps, iters * 8.0, [&]() { hipLaunchKernelGGL(k_regs<8>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
    run("regs x16", wps, iters * 16.0, [&]() { hipLaunchKernelGGL(k_regs<16>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
    run("lds_a", wps, iters * 16.0, [&]() { hipLaunchKernelGGL(k_lds_a, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
    run("gemm", wps, iters * 64.0, [&]() { hipLaunchKernelGGL(k_gemm<false>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
    run("gemm_bar", wps, iters * 64.0, [&]() { hipLaunchKernelGGL(k_gemm<true>, dim3(blocks), dim3(256), 0, 0, d_rand, d_out, d_st, iters); }, d_st, blocks * 4, soak);
  }
  return 0;
}
