// Cost per node of a chain of dependent small kernels (gfx950): plain stream launches against a replayed hipGraph.
// The one-stage tridiagonalisation issues 3 dependent launches per column (k_col_update, k_symv_tiles, k_w_reduce);
// for ONE structure the solve is bound by that chain (~18 000 launches at n = 6000), so what a captured graph saves
// per node decides whether graph capture is worth building.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_graph_chain.hip -o /tmp/probe_graph_chain && /tmp/probe_graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_a(double* x, int n, int step) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = x[i] * 0.999 + step * 1e-9;
}
__global__ __launch_bounds__(256) void k_b(const double* x, double* y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = x[n - 1 - i] + 1.0;
}
__global__ __launch_bounds__(256) void k_c(const double* y, double* x, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += 0.5 * y[i];
}

// grid barrier: monotone counter, agent-scope release / acquire (what a fused multi-workgroup per-column kernel needs
// between its phases: the phases exchange data across XCDs)
__global__ __launch_bounds__(256) void k_persist(double* x, double* y, int n, int steps, unsigned* bar) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned nb = gridDim.x;
  unsigned target = 0;
  for (int s = 0; s < steps; ++s) {
    if (i < n) x[i] = x[i] * 0.999 + s * 1e-9;
    // barrier
    __syncthreads();
    target += nb;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1L << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (i < n) y[i] = x[n - 1 - i] + 1.0;
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int n = 6000, cols = 2000;   // 3 * cols nodes
  double *x, *y;
  CK(hipMalloc(&x, sizeof(double) * n));
  CK(hipMalloc(&y, sizeof(double) * n));
  CK(hipMemset(x, 0, sizeof(double) * n));
  CK(hipMemset(y, 0, sizeof(double) * n));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const dim3 g((n + 255) / 256), b(256);
  auto chain = [&](hipStream_t s) {
    for (int c = 0; c < cols; ++c) {
      hipLaunchKernelGGL(k_a, g, b, 0, s, x, n, c);
      hipLaunchKernelGGL(k_b, g, b, 0, s, x, y, n);
      hipLaunchKernelGGL(k_c, g, b, 0, s, y, x, n);
    }
  };
  // ---- plain launches
  chain(st);
  CK(hipStreamSynchronize(st));
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    chain(st);
    const double t1 = now();
    CK(hipStreamSynchronize(st));
    const double t2 = now();
    printf("stream launches : issue %.2f us/node, end to end %.2f us/node\n", (t1 - t0) / (3.0 * cols) * 1e6,
           (t2 - t0) / (3.0 * cols) * 1e6);
  }
  // ---- captured graph
  hipGraph_t graph;
  hipGraphExec_t exec;
  const double c0 = now();
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  chain(st);
  CK(hipStreamEndCapture(st, &graph));
  const double c1 = now();
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  const double c2 = now();
  printf("capture %.1f ms, instantiate %.1f ms (%d nodes)\n", (c1 - c0) * 1e3, (c2 - c1) * 1e3, 3 * cols);
  CK(hipGraphLaunch(exec, st));
  CK(hipStreamSynchronize(st));
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    CK(hipGraphLaunch(exec, st));
    const double t1 = now();
    CK(hipStreamSynchronize(st));
    const double t2 = now();
    printf("graph replay    : issue %.2f us/node, end to end %.2f us/node\n", (t1 - t0) / (3.0 * cols) * 1e6,
           (t2 - t0) / (3.0 * cols) * 1e6);
  }
  // ---- one persistent kernel with a grid barrier per "node" (what a fused per-column kernel would pay)
  unsigned* bar;
  CK(hipMalloc(&bar, 256));
  for (int wgs : {24, 64, 256, 512}) {
    CK(hipMemset(bar, 0, 256));
    const int nn = wgs * 256, steps = 2000;
    double *xx, *yy;
    CK(hipMalloc(&xx, sizeof(double) * nn));
    CK(hipMalloc(&yy, sizeof(double) * nn));
    CK(hipMemset(xx, 0, sizeof(double) * nn));
    void* args[] = {(void*)&xx, (void*)&yy, (void*)&nn, (void*)&steps, (void*)&bar};
    const double t0 = now();
    CK(hipLaunchCooperativeKernel((const void*)k_persist, dim3(wgs), dim3(256), args, 0, st));
    CK(hipStreamSynchronize(st));
    const double t1 = now();
    printf("persistent kernel, %3d workgroups: %.2f us per grid barrier (agent-scope release / acquire)\n", wgs,
           (t1 - t0) / steps * 1e6);
    CK(hipFree(xx));
    CK(hipFree(yy));
  }
  printf("done\n");
  return 0;
}
