// Round 6: would an integer-MFMA emulation of the long-K f64 products pay on this chip?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_i8.hip -o /tmp/probe_i8 && /tmp/probe_i8
// A bounded probe of the two rates that bound any such scheme, measured on the box, for the shape of the Q1
// back-transformation's W = V^T Z (256 x 6000 by 6000 x 6000, K = 6000; k_gemm3 runs it at 0.82 of the f64 peak):
//   (1) the int8 matrix pipe: v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8 issued back to back from registers, every
//       CU busy, 1 / 2 waves per SIMD -- an UPPER bound for any int8 GEMM kernel (no operand traffic at all);
//   (2) the split pass: a 6000 x 6000 f64 operand read once and written as s planes of int8 (row-scaled, error-free slices of
//       7 bits) -- what an Ozaki-type scheme must do to the operand that changes between calls (Z changes with every
//       reflector block), a LOWER bound for the conversion (no scaling reduction pass, no second operand);
//   (3) the same for a residue scheme ("Ozaki II": N moduli, one int8 plane per modulus).
// The model and the conclusion are printed from these numbers (profiles/r06_probe_i8_emulation.txt); the error of the slice
// scheme for s slices is emulated in NumPy (tools/models/ozaki_error.py) on operands shaped like the solver's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256, 2) k_i8_32(const int* __restrict__ src, int* out, int iters) {
  v16i acc[NACC];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  v4i a, b;
  for (int i = 0; i < 4; ++i) { a[i] = src[(gid + i) & 65535]; b[i] = src[(gid + 77 + i) & 65535]; }
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[gid] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256, 2) k_i8_16(const int* __restrict__ src, int* out, int iters) {
  v4i acc[NACC];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  v4i a, b;
  for (int i = 0; i < 4; ++i) { a[i] = src[(gid + i) & 65535]; b[i] = src[(gid + 77 + i) & 65535]; }
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4i{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[gid] = s;
}

// (2) f64 -> S planes of 7-bit signed slices, scaled per row by a power of two (here: a given exponent per row; the pass
// that finds it -- a row maximum -- is not even counted).  One thread per element, planes written as int8, coalesced.
template <int S>
__global__ void __launch_bounds__(256) k_split(const double* __restrict__ z, const int* __restrict__ row_exp, signed char* __restrict__ planes,
                                               int n, long long plane_stride) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)n * n) return;
  const int r = (int)(idx % n);
  double x = ldexp(z[idx], -row_exp[r]);     // |x| < 1
#pragma unroll
  for (int p = 0; p < S; ++p) {
    x *= 128.0;
    const double q = trunc(x);               // 7 bits + sign, exact
    planes[(long long)p * plane_stride + idx] = (signed char)(int)q;
    x -= q;
  }
}

// (3) residues: the operand scaled to integers of ~53 bits (two 32-bit halves here), one int8 plane per modulus
template <int N>
__global__ void __launch_bounds__(256) k_residues(const double* __restrict__ z, const int* __restrict__ row_exp, signed char* __restrict__ planes,
                                                  int n, long long plane_stride) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)n * n) return;
  const int r = (int)(idx % n);
  const double x = ldexp(z[idx], 52 - row_exp[r]);
  const long long v = (long long)x;
  constexpr int mod[20] = {256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193, 191, 181, 179, 173};
#pragma unroll
  for (int p = 0; p < N; ++p) {
    long long m = v % mod[p];
    planes[(long long)p * plane_stride + idx] = (signed char)(m > 127 ? m - mod[p] : m);
  }
}

template <class F>
static double time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.gcnArchName, cus);
  int* src;
  int* out;
  CK(hipMalloc(&src, 65536 * 4 + 64));
  std::vector<int> h(65536 + 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (int)(i * 2654435761u);
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  const int iters = 20000;
  printf("\n(1) int8 MFMA issued from registers (an upper bound for any int8 GEMM):\n");
  for (int wg_per_cu : {1, 2}) {
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(k_i8_32<4>, dim3(cus * wg_per_cu), dim3(256), 0, 0, src, out, iters); }, 3);
      const double ops = 2.0 * 32 * 32 * 32 * 4.0 * iters * 4.0 * cus * wg_per_cu;
      printf("  v_mfma_i32_32x32x32_i8, %d wave(s) per SIMD, 4 accumulators: %8.3f ms  %7.1f TOP/s\n", wg_per_cu, ms, ops / ms / 1e9);
    }
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(k_i8_16<8>, dim3(cus * wg_per_cu), dim3(256), 0, 0, src, out, iters); }, 3);
      const double ops = 2.0 * 16 * 16 * 64 * 8.0 * iters * 4.0 * cus * wg_per_cu;
      printf("  v_mfma_i32_16x16x64_i8, %d wave(s) per SIMD, 8 accumulators: %8.3f ms  %7.1f TOP/s\n", wg_per_cu, ms, ops / ms / 1e9);
    }
  }
  // ---- (2), (3): the operand that changes between calls
  const int n = 6000;
  double* z;
  int* rexp;
  signed char* planes;
  const long long elems = (long long)n * n;
  CK(hipMalloc(&z, elems * 8));
  CK(hipMalloc(&rexp, n * 4));
  CK(hipMalloc(&planes, elems * 20));
  {
    std::vector<double> hz(elems);
    unsigned long long sd = 88172645463325252ull;
    for (auto& x : hz) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; x = (double)((long long)(sd % 2000001) - 1000000) / 1000001.0; }
    CK(hipMemcpy(z, hz.data(), elems * 8, hipMemcpyHostToDevice));
    CK(hipMemset(rexp, 0, n * 4));
  }
  const unsigned grid = (unsigned)((elems + 255) / 256);
  printf("\n(2) split of one %d x %d f64 operand into s planes of 7-bit slices (read 288 MB, write s x 36 MB):\n", n, n);
  double ms8 = 0, ms16 = 0;
  {
    const double ms = time_ms([&] { hipLaunchKernelGGL(k_split<6>, dim3(grid), dim3(256), 0, 0, z, rexp, planes, n, elems); }, 5);
    printf("  s = 6: %.3f ms (%.2f TB/s)\n", ms, (elems * 8.0 + elems * 6.0) / ms / 1e9);
  }
  {
    const double ms = time_ms([&] { hipLaunchKernelGGL(k_split<8>, dim3(grid), dim3(256), 0, 0, z, rexp, planes, n, elems); }, 5);
    printf("  s = 8: %.3f ms (%.2f TB/s)\n", ms, (elems * 8.0 + elems * 8.0) / ms / 1e9);
    ms8 = ms;
  }
  printf("(3) the same operand as residues modulo N pairwise coprime moduli <= 256 (read 288 MB, write N x 36 MB):\n");
  {
    const double ms = time_ms([&] { hipLaunchKernelGGL(k_residues<16>, dim3(grid), dim3(256), 0, 0, z, rexp, planes, n, elems); }, 5);
    printf("  N = 16: %.3f ms (%.2f TB/s)\n", ms, (elems * 8.0 + elems * 16.0) / ms / 1e9);
    ms16 = ms;
  }
  {
    const double ms = time_ms([&] { hipLaunchKernelGGL(k_residues<20>, dim3(grid), dim3(256), 0, 0, z, rexp, planes, n, elems); }, 5);
    printf("  N = 20: %.3f ms (%.2f TB/s)\n", ms, (elems * 8.0 + elems * 20.0) / ms / 1e9);
  }
  // ---- the model for W = V^T Z (256 x 6000 x 6000)
  const double flops = 2.0 * 256 * 6000.0 * 6000.0;
  const double t_native = flops / (0.82 * 78.6e12) * 1e3;
  printf("\nmodel, W = V^T Z (256 x 6000 by 6000 x 6000): k_gemm3 at 0.82 of 78.6 TFLOP/s = %.3f ms per call\n", t_native);
  printf("  slices, s = 8 (36 int8 GEMMs of the same shape): split of Z alone %.3f ms = %.2f x the native call, before a single\n"
         "    integer MFMA; the 36 products are 6.6e11 integer operations\n", ms8, ms8 / t_native);
  printf("  residues, N = 16 (16 int8 GEMMs): split of Z alone %.3f ms = %.2f x the native call; the 16 products are 2.9e11\n"
         "    integer operations\n", ms16, ms16 / t_native);
  // ---- and for the largest cubic product of the step: a top-level D&C merge, (3000 x 2250) (2250 x 4500) per half, k_gemm2 at 0.775
  {
    const double f2 = 2.0 * 3000.0 * 2250.0 * 4500.0;
    const double t2 = f2 / (0.775 * 78.6e12) * 1e3;
    const double best_i8 = 4.17e15;   // (1): the best register-only rate above, TOP/s -- no kernel reaches it
    printf("model, top-level D&C merge (3000 x 2250 by 2250 x 4500): k_gemm2 at 0.775 = %.3f ms\n", t2);
    printf("  slices, s = 9 (45 products, error <= 2^-50 by the emulation below): %.3f ms at the register-only bound, %.3f ms at 0.6 of it\n",
           45.0 * f2 / best_i8 * 1e3, 45.0 * f2 / (0.6 * best_i8) * 1e3);
    printf("  residues, N = 16: %.3f ms at 0.6 of the bound + conversion of both operands (135 MB: %.3f ms at this probe's rate) + CRT of the output\n",
           16.0 * f2 / (0.6 * best_i8) * 1e3, ms16 * 135.0 / 288.0);
  }
  printf("\nconclusion: on this chip the f64 matrix rate is 1 / 53 of the measured int8 rate (78.6 against <= 4170), an error-free slice\n"
         "scheme at f64 accuracy needs 36 - 45 int8 products, and the operand that changes between calls must be split every time.\n"
         "W = V^T Z: slower than k_gemm3 even at the unreachable register-only rate (split 0.12 ms + >= 0.16 ms against 0.29 ms).\n"
         "The cubic D&C merges could gain at most ~1.3 x with the residue scheme and a 2.5 POP/s int8 GEMM (which does not exist\n"
         "here), on 142 of the step's ~2150 ms.  Not adopted.\n");
  return 0;
}
