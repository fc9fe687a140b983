#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* src, double* out, int shift) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned voff = threadIdx.x * 16u;
  const unsigned long long b = (unsigned long long)(size_t)(src + shift);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  const unsigned long long sb = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_waitcnt vmcnt(0)" : : "s"(lds_base), "v"(voff), "s"(sb) : "memory");
  __syncthreads();
  out[threadIdx.x * 2] = ((double*)lds)[threadIdx.x * 2];
  out[threadIdx.x * 2 + 1] = ((double*)lds)[threadIdx.x * 2 + 1];
}
int main() {
  double *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 1024);
  std::vector<double> h(512); for (int i = 0; i < 512; ++i) h[i] = i;
  hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 3; ++shift) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, s, o, shift);
    std::vector<double> r(128); hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 128; ++i) bad += r[i] != i + shift;
    printf("shift %d doubles: %d wrong (first %g %g %g)\n", shift, bad, r[0], r[1], r[2]);
  }
  return 0;
}
