// Calibration of the PMC counters FETCH_SIZE / WRITE_SIZE for 8-byte-per-lane accesses (MI355X_MICROARCH.md calibrates
// them for 16-byte-per-lane streams only): three kernels move a known number of bytes of a 2 GiB buffer (> Infinity
// Cache), reading / writing 8 B per lane, 16 B per lane, and 8 B per lane in 128-byte segments strided like the
// Z window of k_bt2_fused (16 consecutive lanes per column).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_fetch_width.hip -o build/probe_fetch_width
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- build/probe_fetch_width   (and WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read8(const double* __restrict__ a, size_t n, double* out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s == 1234.5) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read16(const double2* __restrict__ a, size_t n2, double* out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 1234.5) out[0] = s;
}
// 8 B per lane, 16 lanes contiguous (128 B), the 4 lane groups of a wave in 4 different columns of pitch `ld`
__global__ __launch_bounds__(256) void k_read8_seg(const double* __restrict__ a, size_t rows, size_t ld, size_t cols, double* out) {
  double s = 0.0;
  const int fr = threadIdx.x & 15, g = threadIdx.x >> 4;   // 16 groups of 16 lanes
  for (size_t c0 = (size_t)blockIdx.x * 16; c0 < cols; c0 += (size_t)gridDim.x * 16)
    for (size_t r = 0; r < rows; r += 16) s += a[(c0 + g) * ld + r + fr];
  if (s == 1234.5) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write8(double* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = 1.0;
}
__global__ __launch_bounds__(256) void k_write8_seg(double* __restrict__ a, size_t rows, size_t ld, size_t cols) {
  const int fr = threadIdx.x & 15, g = threadIdx.x >> 4;
  for (size_t c0 = (size_t)blockIdx.x * 16; c0 < cols; c0 += (size_t)gridDim.x * 16)
    for (size_t r = 0; r < rows; r += 16) a[(c0 + g) * ld + r + fr] = 2.0;
}

int main() {
  const size_t n = (size_t)1 << 28;   // 2^28 doubles = 2 GiB
  double* a; CK(hipMalloc(&a, n * 8)); CK(hipMemset(a, 0, n * 8));
  double* out; CK(hipMalloc(&out, 64));
  const size_t ld = 6000, rows = 5984, cols = (n / ld) / 16 * 16;   // 6000-pitch columns; rows and cols multiples of 16 (the kernels assume it)
  hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, a, n, out);
  hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2*)a, n / 2, out);
  hipLaunchKernelGGL(k_read8_seg, dim3(4096), dim3(256), 0, 0, a, rows, ld, cols, out);
  hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, a, n);
  hipLaunchKernelGGL(k_write8_seg, dim3(4096), dim3(256), 0, 0, a, rows, ld, cols);
  CK(hipDeviceSynchronize());
  printf("bytes: read8 %zu read16 %zu read8_seg %zu write8 %zu write8_seg %zu\n", n * 8, n * 8, rows * cols * 8, n * 8, rows * cols * 8);
  return 0;
}
