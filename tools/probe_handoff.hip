// Cost of a producer -> consumer hand-off between workgroups inside one kernel (gfx950), the building block of a
// persistent bulge chase with per-sweep progress counters.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_handoff.hip -o /tmp/probe_handoff && /tmp/probe_handoff
// Chains of C workgroups (all resident: grid <= 4 per CU); link r of a chain repeatedly waits until link r-1 has
// published step i, reads the 32 KB block link r-1 wrote, adds 1, writes its own block and publishes step i.  The
// pipeline is full after C steps, so time / steps = the hand-off latency of one link (wait + read + write + publish).
//   mode 0: agent-scope release / acquire atomics (what the memory model asks for across XCDs)
//   mode 1: the chain's workgroups sit on ONE XCD (blockIdx & 7), relaxed atomics, stores drained with vmcnt(0) before
//           the publish, data loads with sc1 (bypass the CU's L1, hit the XCD's L2)
// Every spin is bounded; a time-out sets a flag that all links see, so the kernel always ends.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double load_sc1(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_chain(double* data, int* progress, int* abort_flag, int chain_len, int steps, int* xcd_bad) {
  // chain c = all workgroups with the same (blockIdx & 7, blockIdx >> 3 / chain_len); link r inside it
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chain = xcd + 8 * (slot / chain_len), r = slot % chain_len;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if ((int)(id & 7) != xcd) atomicExch(xcd_bad, 1);
  }
  double* mine = data + ((size_t)chain * chain_len + r) * 4096;
  const double* prev = data + ((size_t)chain * chain_len + (r + chain_len - 1) % chain_len) * 4096;
  int* my_prog = progress + chain * chain_len + r;
  const int* prev_prog = progress + chain * chain_len + (r + chain_len - 1) % chain_len;
  __shared__ int s_ok;
  for (int i = 1; i <= steps; ++i) {
    // link 0 of step i follows the LAST link of step i - 1 (a ring): wait for the predecessor's step
    const int need = r == 0 ? i - 1 : i;
    if (threadIdx.x == 0) {
      int ok = 1;
      long spins = 0;
      for (;;) {
        int p = MODE == 0 ? __hip_atomic_load(prev_prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                          : __hip_atomic_load(prev_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p >= need) break;
        if (++spins > (1L << 22) || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = MODE == 0 ? prev[u * 256 + threadIdx.x] : load_sc1(prev + u * 256 + threadIdx.x);
#pragma unroll
    for (int u = 0; u < 16; ++u) mine[u * 256 + threadIdx.x] = v[u] + 1.0;
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) __hip_atomic_store(my_prog, i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(my_prog, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int chain_len = 16, chains = cus * 4 / chain_len / 8 * 8, grid = chains * chain_len, steps = 2000;
  double* data; int *progress, *abort_flag, *xcd_bad;
  CK(hipMalloc(&data, (size_t)grid * 4096 * 8)); CK(hipMalloc(&progress, grid * 4)); CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&xcd_bad, 4));
  printf("%d chains of %d workgroups (grid %d on %d CUs), %d steps, 32 KB read + written per hand-off\n", chains, chain_len, grid, cus, steps);
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(data, 0, (size_t)grid * 4096 * 8)); CK(hipMemset(progress, 0, grid * 4)); CK(hipMemset(abort_flag, 0, 4)); CK(hipMemset(xcd_bad, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(grid), dim3(256), 0, 0, data, progress, abort_flag, chain_len, steps, xcd_bad);
    else hipLaunchKernelGGL(k_chain<1>, dim3(grid), dim3(256), 0, 0, data, progress, abort_flag, chain_len, steps, xcd_bad);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int h_abort = 0, h_bad = 0; std::vector<double> h(4096);
    CK(hipMemcpy(&h_abort, abort_flag, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_bad, xcd_bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h.data(), data + (size_t)(chain_len - 1) * 4096, 4096 * 8, hipMemcpyDeviceToHost));
    // the ring adds 1 per link per step: the last link of chain 0 holds steps * chain_len
    printf("%-58s %8.2f ms  %.2f us per hand-off  value %.0f (expected %d)  abort %d  xcd mismatch %d\n",
           mode == 0 ? "agent-scope release / acquire" : "one XCD per chain, relaxed + vmcnt(0) + sc1 loads", ms,
           ms * 1e3 / ((double)steps * chain_len), h[0], steps * chain_len, h_abort, h_bad);
  }
  return 0;
}
