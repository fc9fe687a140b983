// Mode-subset consumers on device-resident eigenpairs.
//
// The reference evaluates these with NumPy on the (eig_values, eig_vectors) pair that nma.eigen returns, solving the
// eigenproblem again for every quantity (nma.py:108-184 msf, :233-359 dcc, :476-524 prs).  Here the eigenpairs stay in
// HBM (struct sc_modes, api.hip) and each quantity is one or two kernels plus, for the matrix-valued ones, one MFMA
// GEMM; only the (N) / (N, N) result crosses PCIe.
//
//   msf[a]    = sum_{k in S} sum_d V[k, dim a + d]^2 / w[k]
//   dcc[a, b] = sum_{k in S} sum_d V[k, dim a + d] V[k, dim b + d] / w[k]      (optionally / sqrt(dcc[a,a] dcc[b,b]))
//   prs[a, b] = sum_{d, e} C[3a + d, 3b + e]^2,  C = pinv(H)                      (optionally / prs[a, a])
//
// V is stored as the solver leaves it: (n, n) row-major, row k = mode k.
#include <algorithm>

#include "common.h"
#include "gemm_f64.h"

namespace {

// ---- msf: partial sums over chunks of 32 selected modes, then the chunk + dim reduction -----------------------
constexpr int kMsfChunk = 32;

__global__ __launch_bounds__(256) void k_msf_partial(const double* __restrict__ v, const double* __restrict__ w,
                                                     const int* __restrict__ sel, int nsel, int n,
                                                     double* __restrict__ part) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int k0 = blockIdx.y * kMsfChunk, k1 = min(k0 + kMsfChunk, nsel);
  double acc = 0.0;
  for (int kk = k0; kk < k1; ++kk) {
    const int k = sel[kk];
    const double x = v[(size_t)k * n + j];
    acc += x * x / w[k];
  }
  part[(size_t)blockIdx.y * n + j] = acc;
}

__global__ __launch_bounds__(256) void k_msf_reduce(const double* __restrict__ part, int nchunk, int n, int dim,
                                                    double* __restrict__ out) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= n / dim) return;
  double acc = 0.0;
  for (int c = 0; c < nchunk; ++c)
    for (int d = 0; d < dim; ++d) acc += part[(size_t)c * n + a * dim + d];
  out[a] = acc;
}

// ---- dcc: pack the selected modes component-major, so the contraction over (mode, component) is one GEMM ------
//   P[(d * nsel + kk) * N + a] = V[sel[kk], dim a + d]                S = same / w[sel[kk]]
__global__ __launch_bounds__(256) void k_dcc_pack(const double* __restrict__ v, const double* __restrict__ w,
                                                  const int* __restrict__ sel, int nsel, int n, int dim,
                                                  double* __restrict__ p, double* __restrict__ s) {
  const int kk = blockIdx.y;
  const int k = sel[kk];
  const double inv = 1.0 / w[k];
  const int N = n / dim;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
    const double x = v[(size_t)k * n + j];
    const int a = j / dim, d = j - a * dim;
    const size_t o = ((size_t)d * nsel + kk) * N + a;
    p[o] = x;
    s[o] = x * inv;
  }
}

// in place: c[a, b] /= sqrt(c[a, a] c[b, b]) needs the original diagonal -> copy it first
__global__ void k_copy_diag(const double* __restrict__ c, int N, double* __restrict__ diag) {
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a < N) diag[a] = c[(size_t)a * N + a];
}

__global__ __launch_bounds__(256) void k_dcc_norm(double* __restrict__ c, const double* __restrict__ diag, int N) {
  const int a = blockIdx.x * 256 + threadIdx.x;   // fast axis of the column-major (= row-major, symmetric) result
  const int b = blockIdx.y;
  if (a >= N) return;
  // the reference divides by outer(sqrt(diag), sqrt(diag)) (nma.py:352-354)
  c[(size_t)b * N + a] = c[(size_t)b * N + a] / (sqrt(diag[a]) * sqrt(diag[b]));
}

// ---- prs ------------------------------------------------------------------------------------------------------
// scale the eigenvector rows by 1/w for |w| > rcond * max|w|, else 0 (numpy.linalg.pinv(hermitian=True) rule)
__global__ __launch_bounds__(256) void k_pinv_rows(const double* __restrict__ v, const double* __restrict__ w, int n,
                                                   double rcond, double* __restrict__ vs) {
  __shared__ double red[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, fabs(w[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const int k = blockIdx.x;
  const double wk = w[k];
  const double sc = fabs(wk) > rcond * red[0] ? 1.0 / wk : 0.0;
  for (int j = threadIdx.x; j < n; j += 256) vs[(size_t)k * n + j] = v[(size_t)k * n + j] * sc;
}

// out[a, b] = sum of the squared entries of the 3x3 block (a, b) of the covariance; row-major (N, N)
__global__ __launch_bounds__(256) void k_prs_blocks(const double* __restrict__ cov, int N, double* __restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const int a = blockIdx.y;
  if (b >= N) return;
  const size_t n = (size_t)3 * N;
  double acc = 0.0;
  // same association as np.add.reduceat over rows, then over columns (nma.py:515-517)
  double colsum[3];
  for (int e = 0; e < 3; ++e) {
    double s = 0.0;
    for (int d = 0; d < 3; ++d) {
      const double x = cov[(size_t)(3 * a + d) * n + 3 * b + e];
      s += x * x;
    }
    colsum[e] = s;
  }
  acc = (colsum[0] + colsum[1]) + colsum[2];
  out[(size_t)a * N + b] = acc;
}

__global__ __launch_bounds__(256) void k_prs_norm(double* __restrict__ m, const double* __restrict__ diag, int N) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const int a = blockIdx.y;
  if (b >= N) return;
  m[(size_t)a * N + b] = m[(size_t)a * N + b] / diag[a];
}

int run_gemm(sc_ctx* ctx, const GemmDesc& D, GemmDesc* d_desc) {
  SC_HIP(ctx, hipMemcpyAsync(d_desc, &D, sizeof(D), hipMemcpyHostToDevice, ctx->stream));
  // both callers: A(i, k) with unit row stride, B(k, j) with unit column stride
  return launch_gemm_f64(ctx, d_desc, 1, D.m, D.n, kGemmTile, 1, false, false, kGemmAmBn);
}

}  // namespace

size_t modes_scratch_bytes(int64_t n, int dim, int64_t nsel, int what) {
  const size_t N = (size_t)(n / dim);
  switch (what) {
    case 0: return align_up(((size_t)(nsel + kMsfChunk - 1) / kMsfChunk) * n * 8, 256) + align_up(N * 8, 256) + 1024;
    case 1: return 2 * align_up((size_t)nsel * n * 8, 256) + align_up(N * N * 8, 256) + align_up(N * 8, 256) + 1024;
    default: return 2 * align_up((size_t)n * n * 8, 256) + align_up(N * N * 8, 256) + align_up(N * 8, 256) + 1024;
  }
}

// d_sel: (nsel) int32 mode indices on the device; d_out (N); d_part scratch of modes_scratch_bytes(.., 0)
int modes_msf_device(sc_ctx* ctx, const double* d_v, const double* d_w, int64_t n64, int dim, const int* d_sel,
                     int64_t nsel64, char* scratch, double* d_out) {
  const int n = (int)n64, nsel = (int)nsel64, N = n / dim;
  hipStream_t st = ctx->stream;
  if (nsel == 0) {
    SC_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(double) * N, st));
    return SC_OK;
  }
  const int nchunk = (nsel + kMsfChunk - 1) / kMsfChunk;
  double* d_part = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(k_msf_partial, dim3((n + 255) / 256, nchunk), dim3(256), 0, st, d_v, d_w, d_sel, nsel, n, d_part);
  hipLaunchKernelGGL(k_msf_reduce, dim3((N + 255) / 256), dim3(256), 0, st, d_part, nchunk, n, dim, d_out);
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}

int modes_dcc_device(sc_ctx* ctx, const double* d_v, const double* d_w, int64_t n64, int dim, const int* d_sel,
                     int64_t nsel64, int norm, char* scratch, double* d_out) {
  const int n = (int)n64, nsel = (int)nsel64, N = n / dim;
  hipStream_t st = ctx->stream;
  if (nsel == 0) {
    SC_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)N * N, st));
  } else {
    double* d_p = reinterpret_cast<double*>(scratch);
    double* d_s = reinterpret_cast<double*>(scratch + align_up((size_t)nsel * n * 8, 256));
    hipLaunchKernelGGL(k_dcc_pack, dim3(std::min((n + 255) / 256, 64), nsel), dim3(256), 0, st, d_v, d_w, d_sel, nsel,
                       n, dim, d_p, d_s);
    SC_HIP(ctx, hipGetLastError());
    GemmDesc D{};
    D.a = d_s; D.sa_i = 1; D.sa_k = N;
    D.b = d_p; D.sb_k = N; D.sb_j = 1;
    D.c = d_out; D.ldc = N; D.m = N; D.n = N; D.k = nsel * dim;
    D.alpha = 1.0; D.beta = 0.0;
    GemmDesc* d_desc = reinterpret_cast<GemmDesc*>(scratch + 2 * align_up((size_t)nsel * n * 8, 256) +
                                                   align_up((size_t)N * 8, 256));
    SC_TRY(run_gemm(ctx, D, d_desc));
  }
  if (norm) {
    double* d_diag = reinterpret_cast<double*>(scratch + 2 * align_up((size_t)nsel * n * 8, 256));
    hipLaunchKernelGGL(k_copy_diag, dim3((N + 255) / 256), dim3(256), 0, st, d_out, N, d_diag);
    hipLaunchKernelGGL(k_dcc_norm, dim3((N + 255) / 256, N), dim3(256), 0, st, d_out, d_diag, N);
    SC_HIP(ctx, hipGetLastError());
  }
  return SC_OK;
}

// ANM only (n = 3N).  scratch: VS (n^2) | COV (n^2) | diag (N) | desc
int modes_prs_device(sc_ctx* ctx, const double* d_v, const double* d_w, int64_t n64, double rcond, int norm,
                     char* scratch, double* d_out) {
  const int n = (int)n64, N = n / 3;
  hipStream_t st = ctx->stream;
  const size_t mat = align_up((size_t)n * n * 8, 256);
  double* d_vs = reinterpret_cast<double*>(scratch);
  double* d_cov = reinterpret_cast<double*>(scratch + mat);
  double* d_diag = reinterpret_cast<double*>(scratch + 2 * mat);
  GemmDesc* d_desc = reinterpret_cast<GemmDesc*>(scratch + 2 * mat + align_up((size_t)N * 8, 256));
  hipLaunchKernelGGL(k_pinv_rows, dim3(n), dim3(256), 0, st, d_v, d_w, n, rcond, d_vs);
  SC_HIP(ctx, hipGetLastError());
  // cov[i, j] = sum_k VS[k, i] V[k, j]
  GemmDesc D{};
  D.a = d_vs; D.sa_i = 1; D.sa_k = n;
  D.b = d_v; D.sb_k = n; D.sb_j = 1;
  D.c = d_cov; D.ldc = n; D.m = n; D.n = n; D.k = n;
  D.alpha = 1.0; D.beta = 0.0;
  SC_TRY(run_gemm(ctx, D, d_desc));
  hipLaunchKernelGGL(k_prs_blocks, dim3((N + 255) / 256, N), dim3(256), 0, st, d_cov, N, d_out);
  if (norm) {
    hipLaunchKernelGGL(k_copy_diag, dim3((N + 255) / 256), dim3(256), 0, st, d_out, N, d_diag);
    hipLaunchKernelGGL(k_prs_norm, dim3((N + 255) / 256, N), dim3(256), 0, st, d_out, d_diag, N);
  }
  SC_HIP(ctx, hipGetLastError());
  return SC_OK;
}
