// How fast can a GEMM workgroup read-modify-write its C tile?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_ctile.hip -o /tmp/probe_ctile && /tmp/probe_ctile
// 64 column-major matrices of 6000 x 6000 doubles (18.4 GB); every workgroup (256 threads, 2 per CU as k_gemm2 at
// 128 x 128) reads one 128 x 128 tile, adds 1, writes it back.  Lane maps:
//   mfma8   the accumulator layout of k_gemm2: lane (fr, fk) moves 8 B of row (16 mi + fr) of column (4 r + fk): one
//           wave instruction = 4 columns x 128 B
//   row16   16 B per lane, 8 lanes per 128-B column segment: one wave instruction = 8 columns x 128 B
//   row16x  the same, a whole 1 KB column (128 rows) per 64 lanes: one wave instruction = 1 column x 1 KB
// in the variants  r (read only), w (write only), rw (read, then write after all reads of the tile have landed) and, for
// mfma8, global_atomic_add_f64 without return (the memory side does the read-modify-write; counted as rw bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MAP, int MODE>   // MODE 0 r, 1 w, 2 rw
__global__ __launch_bounds__(256, 2) void k_tile(double* __restrict__ a, int n, int tiles_per_dim, long long stride, double* sink) {
  const int t = blockIdx.x, b = blockIdx.y;
  const int tm = t % tiles_per_dim, tn = t / tiles_per_dim;
  double* C = a + (size_t)b * stride + (size_t)(tn * 128) * n + tm * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double s = 0.0;
  if (MAP == 0) {
    const int fr = lane & 15, fk = lane >> 4, wm = w & 1, wn = w >> 1;
    double v[4][4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          double* p = C + (size_t)(wn * 64 + ni * 16 + fk + 4 * r) * n + wm * 64 + mi * 16 + fr;
          v[ni][r][mi] = (MODE == 1 || MODE == 3) ? 1.0 : *p;
        }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          double* p = C + (size_t)(wn * 64 + ni * 16 + fk + 4 * r) * n + wm * 64 + mi * 16 + fr;
          if (MODE == 0) s += v[ni][r][mi];
          else if (MODE == 3) (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)p, v[ni][r][mi]);
          else *p = v[ni][r][mi] + 1.0;
        }
  } else if (MAP == 1) {
    // wave w: columns 32 w .. 32 w + 31; instruction: 8 columns x 16 rows; lane: column l >> 3, rows 2 (l & 7), + 1
    const int c8 = lane >> 3, r2 = (lane & 7) * 2;
    d2 v[4][8];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
      for (int rg = 0; rg < 8; ++rg) {
        d2* p = (d2*)(C + (size_t)(32 * w + 8 * cg + c8) * n + 16 * rg + r2);
        v[cg][rg] = MODE == 1 ? d2{1.0, 1.0} : *p;
      }
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
      for (int rg = 0; rg < 8; ++rg) {
        d2* p = (d2*)(C + (size_t)(32 * w + 8 * cg + c8) * n + 16 * rg + r2);
        if (MODE == 0) s += v[cg][rg][0] + v[cg][rg][1]; else *p = v[cg][rg] + 1.0;
      }
  } else {
    // instruction: one column, 128 rows (1 KB): lane rows 2 l, 2 l + 1; wave w: columns 32 w ..
    d2 v[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      d2* p = (d2*)(C + (size_t)(32 * w + c) * n + 2 * lane);
      v[c] = MODE == 1 ? d2{1.0, 1.0} : *p;
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      d2* p = (d2*)(C + (size_t)(32 * w + c) * n + 2 * lane);
      if (MODE == 0) s += v[c][0] + v[c][1]; else *p = v[c] + 1.0;
    }
  }
  if (MODE == 0 && s == 1234.5) sink[0] = s;
}

template <int MAP, int MODE>
static void run(const char* name, double* a, int n, int batch, double* sink) {
  const int tpd = n / 128;   // whole tiles only
  dim3 grid(tpd * tpd, batch);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_tile<MAP, MODE>), grid, dim3(256), 0, 0, a, n, tpd, (long long)n * n, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_tile<MAP, MODE>), grid, dim3(256), 0, 0, a, n, tpd, (long long)n * n, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  const double bytes = (double)tpd * tpd * batch * 128 * 128 * 8 * (MODE >= 2 ? 2 : 1);
  printf("%-34s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / ms * 1e-9);
}

int main() {
  const int n = 6000, batch = 64;
  double *a, *sink;
  CK(hipMalloc(&a, (size_t)n * n * batch * 8)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, (size_t)n * n * batch * 8));
  run<0, 0>("mfma8 r", a, n, batch, sink); run<0, 1>("mfma8 w", a, n, batch, sink); run<0, 2>("mfma8 rw", a, n, batch, sink);
  run<0, 3>("mfma8 atomic add (counted as rw)", a, n, batch, sink);
  run<1, 0>("row16 r", a, n, batch, sink); run<1, 1>("row16 w", a, n, batch, sink); run<1, 2>("row16 rw", a, n, batch, sink);
  run<2, 0>("row16x r", a, n, batch, sink); run<2, 1>("row16x w", a, n, batch, sink); run<2, 2>("row16x rw", a, n, batch, sink);
  return 0;
}
