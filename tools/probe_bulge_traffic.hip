// How long does the memory traffic of the bulge chase take by itself?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_bulge_traffic.hip -o /tmp/probe_bulge && /tmp/probe_bulge [n] [batch]
// Same launch structure, workgroup shape and addresses as k_bulge_step (twostage.hip): launch t runs the tasks (s, k)
// with 2 s + k = t, one 256-thread workgroup each; a task reads its 64 x 64 off-diagonal block E and the lower triangle
// of its diagonal block D from the band storage (128 x n doubles per matrix, column-major), and writes both back.  Here
// NOTHING is computed in between (the values go through registers unchanged, kept alive by an add of a kernel argument
// that is zero), so the time is that of the traffic pattern + the launches alone.  Modes: 0 = loads + stores as the
// kernel issues them, 1 = additionally 4 workgroup barriers and an LDS round trip of E (the data path without the
// arithmetic).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kB = 64, kLdab = 128;
__host__ __device__ inline int chase_len(int n, int s) { return (n - 1 - s + kB - 1) / kB; }

template <int MODE>
__global__ __launch_bounds__(256) void k_traffic(double* __restrict__ ab_all, int n, int t, double zero) {
  constexpr int LD = kB + 1;
  __shared__ double E[kB * LD];
  const int k = (t & 1) + 2 * (int)blockIdx.x;
  const int s = (t - k) / 2;
  if (s < 0 || s > n - 3 || k >= chase_len(n, s)) return;
  double* ab = ab_all + (size_t)blockIdx.y * kLdab * n;
  const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
  const int r0 = s + 1 + k * kB;
  const int L = min(kB, n - r0);
  double d16[16], t16[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int jc = min(q * 16 + u, L - 1);
    const int ic = min(max(i, jc), L - 1);
    d16[u] = ab[(size_t)(ic - jc) + (size_t)(r0 + jc) * kLdab];
  }
  if (k > 0) {
    const int c0 = r0 - kB;
    const int ic = min(i, L - 1);
#pragma unroll
    for (int u = 0; u < 16; ++u) t16[u] = ab[(size_t)(kB + ic - (q * 16 + u)) + (size_t)(c0 + q * 16 + u) * kLdab];
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) E[i * LD + q * 16 + u] = t16[u];
      __syncthreads();
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 16; ++u) t16[u] = E[i * LD + q * 16 + u];
      __syncthreads();
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (i < L) ab[(size_t)(kB + i - (q * 16 + u)) + (size_t)(c0 + q * 16 + u) * kLdab] = t16[u] + zero;
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int jj = q * 16 + u;
    if (i >= jj && i < L) ab[(size_t)(i - jj) + (size_t)(r0 + jj) * kLdab] = d16[u] + zero;
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 6000, batch = argc > 2 ? atoi(argv[2]) : 64;
  double* ab;
  CK(hipMalloc(&ab, (size_t)batch * kLdab * n * 8)); CK(hipMemset(ab, 0, (size_t)batch * kLdab * n * 8));
  const int t_max = 2 * (n - 3) + chase_len(n, n - 3) - 1, gx = chase_len(n, 0) / 2 + 1;
  double tasks = 0;
  for (int s = 0; s <= n - 3; ++s) tasks += chase_len(n, s);
  printf("n = %d, %d matrices: %d launches of (%d, %d) workgroups, %.3g tasks, %.1f GB read + written (96 KB per task)\n", n, batch,
         t_max + 1, gx, batch, tasks * batch, tasks * batch * 96e3 * 1e-9);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int t = 0; t <= t_max; ++t) {
        if (mode == 0) hipLaunchKernelGGL(k_traffic<0>, dim3(gx, batch), dim3(256), 0, 0, ab, n, t, 0.0);
        else hipLaunchKernelGGL(k_traffic<1>, dim3(gx, batch), dim3(256), 0, 0, ab, n, t, 0.0);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1)
        printf("%-34s %8.1f ms  %.2f TB/s  %.1f us per launch\n", mode == 0 ? "loads + stores only" : "+ LDS round trip and 4 barriers", ms,
               tasks * batch * 96e3 / ms * 1e-9, ms * 1e3 / (t_max + 1));
    }
  }
  return 0;
}
