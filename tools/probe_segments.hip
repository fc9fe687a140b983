// Probe: HBM read bandwidth for a column-major tile walk — contiguous segments of SEG bytes separated by a
// 48 KB stride (the column pitch of a 6000 x 6000 f64 matrix), the access pattern of the SYMV tiles.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_segments.hip -o build/probe_segments
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// Each block reads a tile of `rows` x `cols` doubles: thread layout (rows/2) x (256 / (rows/2)); 16-byte loads.
template <int ROWS>
__global__ __launch_bounds__(256, 4) void k_tiles(const double* __restrict__ a, int n, int tiles_r, int tiles_c,
                                                   double* out, int total_tiles) {
  constexpr int COLS = 64 * 64 / ROWS;          // same 32 KB per tile
  constexpr int TR = ROWS / 2;                  // threads along rows
  constexpr int TC = 256 / TR;                  // column groups
  const int lr = threadIdx.x % TR, kq = threadIdx.x / TR;
  double s = 0.0;
  for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    const int tr = t % tiles_r, tc = t / tiles_r;
    const double* base = a + (size_t)(tc * COLS) * n + tr * ROWS + 2 * lr;
    double2 v[COLS / TC];
#pragma unroll
    for (int p = 0; p < COLS / TC; ++p) v[p] = *reinterpret_cast<const double2*>(base + (size_t)(kq + TC * p) * n);
#pragma unroll
    for (int p = 0; p < COLS / TC; ++p) s += v[p].x + v[p].y;
  }
  if (s == 1234.5) out[0] = s;
}

int main() {
  const int n = 6000;
  const size_t bytes = (size_t)n * n * 8;
  const int NMAT = 8;  // > Infinity Cache
  double* a; CK(hipMalloc(&a, bytes * NMAT)); CK(hipMemset(a, 0, bytes * NMAT));
  double* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int rows, int grid) {
    const int cols = 64 * 64 / rows;
    const int tiles_r = n / rows, tiles_c = n / cols;  // full tiles only
    const int total = tiles_r * tiles_c;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int m = 0; m < NMAT; ++m) {
        const double* am = a + (size_t)m * n * n;
        if (rows == 64) k_tiles<64><<<grid, 256>>>(am, n, tiles_r, tiles_c, out, total);
        if (rows == 128) k_tiles<128><<<grid, 256>>>(am, n, tiles_r, tiles_c, out, total);
        if (rows == 256) k_tiles<256><<<grid, 256>>>(am, n, tiles_r, tiles_c, out, total);
        if (rows == 512) k_tiles<512><<<grid, 256>>>(am, n, tiles_r, tiles_c, out, total);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double gb = (double)tiles_r * rows * tiles_c * cols * 8.0 * NMAT / 1e9;
    printf("tile %3d rows x %3d cols (segment %4d B) grid=%5d: %.3f ms  %.2f TB/s\n", rows, cols, rows * 8, grid,
           best, gb / best);
  };
  for (int grid : {1024, 4096}) for (int rows : {64, 128, 256, 512}) run(rows, grid);
  return 0;
}
