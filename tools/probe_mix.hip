// Can the f64 MFMA pipe and the HBM run at their rates at the same time?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mix.hip -o /tmp/probe_mix && /tmp/probe_mix
// One launch of 2 x CUs workgroups: the first CUs workgroups run the k_gemm2-like MFMA loop of probe_mfma2 ("gemm_bar":
// fragments from LDS, 64 MFMAs per k-step, a barrier per 4 k-steps; one wave per SIMD), the second CUs workgroups
// read-modify-write a large buffer (16 B per lane, 1 KB per wave instruction).  Run with only the MFMA half, only the
// memory half, and both: a GEMM whose C traffic is as long as its MFMA work needs the "both" line to hold both rates.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
struct Stamp { unsigned long long c0, r0, c1, r1, role; };
__device__ __forceinline__ void stamp(unsigned long long& c, unsigned long long& r) {
  c = __builtin_amdgcn_s_memtime();
  r = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
}

__global__ void __launch_bounds__(256, 2) k_mix(const double* __restrict__ src, double* out, Stamp* st, int iters, int n_mfma,
                                                d2* big, long long chunk_d2, int passes, int do_mfma, int do_mem, int split, int variant) {
  extern __shared__ double hog[];   // split runs: 100 KB of dynamic LDS force one workgroup per CU
  constexpr int LD = 144;
  __shared__ double sa[16 * LD], sb[16 * LD];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long c0 = 0, r0 = 0, c1 = 0, r1 = 0;
  const bool mfma_role = split == 1 ? (blockIdx.x & 1) == 0 : (int)blockIdx.x < n_mfma;
  if (split == 1 && hog == nullptr) return;
  if (split == 2 && mfma_role && (blockIdx.x & 1)) return;   // MFMA on every other CU only
  if (mfma_role) {
    if (!do_mfma) return;
    d4 acc[4][4];
    for (int i = threadIdx.x; i < 16 * LD; i += 256) { sa[i] = src[(gid + i) & 65535]; sb[i] = src[(gid + 3 * i) & 65535]; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
    const int fr = lane & 15, fk = lane >> 4;
    const double* a_s = sa + fk * LD + (wave & 1) * 64 + fr;
    const double* b_s = sb + fk * LD + (wave >> 1) * 64 + fr;
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
      __syncthreads();
    }
    stamp(c1, r1);
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[gid] = s;
  } else {
    if (!do_mem) return;
    const int w = split == 1 ? blockIdx.x >> 1 : blockIdx.x - n_mfma;
    d2* base = big + (long long)w * chunk_d2;
    stamp(c0, r0);
    if (variant == 0) {          // read, add, write
      for (int p = 0; p < passes; ++p)
        for (long long off = 0; off + 8 * 256 <= chunk_d2; off += 8 * 256) {
          d2 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = base[off + u * 256 + threadIdx.x];
#pragma unroll
          for (int u = 0; u < 8; ++u) base[off + u * 256 + threadIdx.x] = v[u] + 1.0;
        }
    } else if (variant == 1) {   // read, write back unchanged (no f64 VALU)
      for (int p = 0; p < passes; ++p)
        for (long long off = 0; off + 8 * 256 <= chunk_d2; off += 8 * 256) {
          d2 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = base[off + u * 256 + threadIdx.x];
#pragma unroll
          for (int u = 0; u < 8; ++u) base[off + u * 256 + threadIdx.x] = v[u];
        }
    } else if (variant == 2) {   // read only, into registers (results discarded through asm)
      for (int p = 0; p < 2 * passes; ++p)
        for (long long off = 0; off + 8 * 256 <= chunk_d2; off += 8 * 256) {
          d2 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = base[off + u * 256 + threadIdx.x];
#pragma unroll
          for (int u = 0; u < 8; ++u) asm volatile("" :: "v"(v[u]));
        }
    } else if (variant == 3) {   // write only
      const d2 one = d2{1.0, 1.0};
      for (int p = 0; p < 2 * passes; ++p)
        for (long long off = 0; off + 8 * 256 <= chunk_d2; off += 8 * 256) {
#pragma unroll
          for (int u = 0; u < 8; ++u) base[off + u * 256 + threadIdx.x] = one;
        }
    } else {                     // read only, global -> LDS without passing through registers (global_load_lds_dwordx4)
      const unsigned lds_base = __builtin_amdgcn_readfirstlane(
          (unsigned)(size_t)(__attribute__((address_space(3))) double*)sa + (threadIdx.x >> 6) * 1024u);   // 1 KB landing zone per wave
      for (int p = 0; p < 2 * passes; ++p)
        for (long long off = 0; off + 8 * 256 <= chunk_d2; off += 8 * 256) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const d2* g = base + off + u * 256 + threadIdx.x;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((const char*)g), "s"(lds_base) : "memory");
          }
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    stamp(c1, r1);
  }
  if ((threadIdx.x & 63) == 0) st[gid >> 6] = Stamp{c0, r0, c1, r1, mfma_role ? 1ull : (split == 2 ? 2ull + (blockIdx.x & 1) : 0ull)};
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double *d_rand, *d_out; Stamp* d_st; d2* big;
  std::vector<double> hr(65536);
  srand(1);
  for (auto& v : hr) v = 2.0 * rand() / RAND_MAX - 1.0;
  CK(hipMalloc(&d_rand, 65536 * 8)); CK(hipMemcpy(d_rand, hr.data(), 65536 * 8, hipMemcpyHostToDevice));
  const int blocks = 2 * cus;
  CK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
  CK(hipMalloc(&d_st, sizeof(Stamp) * blocks * 4));
  const long long chunk_d2 = (32ll << 20) / 16;   // 32 MiB per memory workgroup
  CK(hipMalloc(&big, (size_t)cus * chunk_d2 * 16)); CK(hipMemset(big, 0, (size_t)cus * chunk_d2 * 16));
  const int iters = 4000, passes = 2;
  printf("device: %s  CUs=%d; MFMA half: %d workgroups x 4 waves x %d k-steps x 64 MFMAs; memory half: %d workgroups x %d x 32 MiB read + written\n",
         p.name, cus, cus, iters, cus, passes);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mix), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  const char* vname[5] = {"read+add+write", "read+write", "read only", "write only", "read to LDS (DMA)"};
  for (int variant = 0; variant < 5; ++variant)
  for (int split = 0; split < (variant == 0 ? 3 : 1); ++split)
  for (int mode = (variant == 0 ? 0 : 1); mode < 3; ++mode) {
    const int do_mfma = mode != 1, do_mem = mode != 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(d_st, 0, sizeof(Stamp) * blocks * 4));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_mix, dim3(split == 1 ? cus : blocks), dim3(256), split == 1 ? 100 * 1024 : 0, 0, d_rand, d_out, d_st, iters, cus, big,
                         split == 1 ? 2 * chunk_d2 : chunk_d2, passes, do_mfma, do_mem, split, variant);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep < 2) continue;
      std::vector<Stamp> h(blocks * 4);
      CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
      std::vector<double> cyc, clk, mem_s, mem_even, mem_odd;
      for (int w = 0; w < blocks * 4; ++w) {
        const double dc = (double)(h[w].c1 - h[w].c0), dr = (double)(h[w].r1 - h[w].r0);
        if (dr <= 0) continue;
        if (h[w].role == 1) { cyc.push_back(dc / (iters * 64.0)); clk.push_back(dc / dr * 0.1); }
        else { mem_s.push_back(dr * 1e-8); if (h[w].role == 2) mem_even.push_back(dr * 1e-8); if (h[w].role == 3) mem_odd.push_back(dr * 1e-8); }
      }
      auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      const double t_mem = med(mem_s), mfma_cyc = med(cyc), g = med(clk);
      const int n_role = split ? cus / 2 : cus;
      const char* sname[3] = {"[both roles on every CU]           ", "[one role per CU, half the CUs each]", "[memory everywhere, MFMA on every 2nd CU]"};
      printf("%-18s %s %-12s kernel %8.3f ms", vname[variant], sname[split], mode == 0 ? "MFMA only" : (mode == 1 ? "memory only" : "both"), ms);
      if (do_mfma) printf("   MFMA: %.1f cycles per MFMA per SIMD at %.3f GHz = %.1f TFLOP/s while its waves run", mfma_cyc, g,
                          n_role * 4.0 * 2048.0 * g / mfma_cyc * 1e-3);
      if (do_mem) printf("   memory: median workgroup %.3f ms -> %.2f TB/s (read + write)", t_mem * 1e3,
                         (double)cus * chunk_d2 * 16.0 * passes * 2.0 / t_mem * 1e-12);   // split: half the workgroups, twice the chunk
      if (split == 2 && do_mem) printf("   memory workgroups next to an MFMA workgroup: median %.3f ms, the others: %.3f ms", med(mem_even) * 1e3, med(mem_odd) * 1e3);
      printf("\n");
    }
  }
  return 0;
}
