// Tall-skinny dense matmul (TSMM, k >> m ~ n in the reference's README.md:39):
//     out[cell][n] = sum_k in[cell][k] * phi[k][n]
// the two cublasDgemm calls of demo/gpu_tsmm/main.cpp:49-52 (100000 x 125 times
// 125 x 125) and the "B" / "B^T" products of demo/gpu_operator/main.cpp:149-155,
// on v_mfma_f64_16x16x4_f64.  phi (K x N, <= 150 KB) is staged once per workgroup
// in LDS; one wave owns 16 cells and keeps all ceil(N/16) accumulator tiles in
// registers; a persistent grid of one 512-thread workgroup per CU walks the cells.
//
// layout 0 (cell-major, demo/gpu_operator): in[cell*K + k], out[cell*N + n];
//          A operand = in tile (16 cells x 4 k), B operand = phi.
// layout 1 (cell-minor = the column-major arrays of demo/gpu_tsmm, lda = ldc =
//          ncells): in[k*ncells + cell], out[n*ncells + cell]; the product is
//          formed transposed (A = phi^T, B = in) so that loads and stores stay
//          contiguous in the cell index.
#include <algorithm>

#include "common.h"

namespace wf {

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NT, int LAYOUT>
__global__ __launch_bounds__(512) void k_tsmm(int64_t ncells, int K, int N, int n0, const double* __restrict__ in,
                                              const double* __restrict__ phi, double* __restrict__ out)
{
  extern __shared__ __attribute__((aligned(16))) double sphi[];   // [KP4][16*NT] (zero padded)
  constexpr int NP = 16 * NT;
  const int KT = (K + 3) / 4, KP4 = 4 * KT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 15, lg = lane >> 4;
  for (int p = t; p < KP4 * NP; p += 512) {
    const int k = p / NP, n = p % NP;
    sphi[p] = (k < K && n0 + n < N) ? phi[(size_t)k * N + n0 + n] : 0.0;
  }
  __syncthreads();
  const int64_t ntiles = (ncells + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
    const int64_t c0 = tile * 16;
    double4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int64_t c = c0 + lc;
    // operand loads run a few k-steps ahead of the MFMAs that consume them
    constexpr int AHEAD = 8;
    double vbuf[AHEAD];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
      const int k = 4 * a + lg;
      vbuf[a] = (a < KT && k < K && c < ncells) ? (LAYOUT == 0 ? in[c * K + k] : in[(int64_t)k * ncells + c]) : 0.0;
    }
    for (int ks0 = 0; ks0 < KT; ks0 += AHEAD) {
#pragma unroll
     for (int a = 0; a < AHEAD; ++a) {
      const int ks = ks0 + a;
      if (ks >= KT) break;
      const int k = 4 * ks + lg;
      const double v = vbuf[a];
      {
        const int kn = 4 * (ks + AHEAD) + lg;
        vbuf[a] = (ks + AHEAD < KT && kn < K && c < ncells) ? (LAYOUT == 0 ? in[c * K + kn] : in[(int64_t)kn * ncells + c]) : 0.0;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const double p = sphi[k * NP + 16 * nt + lc];
        if (LAYOUT == 0)
          acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(v, p, acc[nt], 0, 0, 0);   // D[cell][n]
        else
          acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(p, v, acc[nt], 0, 0, 0);   // D[n][cell]
      }
     }
    }
    // D layout: row = lg + 4 r, col = lc
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (LAYOUT == 0) {
          const int64_t c = c0 + lg + 4 * r;
          const int n = n0 + 16 * nt + lc;
          if (c < ncells && n < N) out[c * N + n] = acc[nt][r];
        } else {
          const int n = n0 + 16 * nt + lg + 4 * r;
          const int64_t c = c0 + lc;
          if (c < ncells && n < N) out[(int64_t)n * ncells + c] = acc[nt][r];
        }
      }
  }
}

template <int NT>
static int launch_tsmm_t(int layout, int64_t ncells, int K, int N, int n0, const double* in, const double* phi,
                         double* out, hipStream_t s)
{
  const int KP4 = 4 * ((K + 3) / 4);
  const size_t lds = (size_t)KP4 * 16 * NT * sizeof(double);
  if (lds > 160 * 1024) {
    set_error("wf_tsmm: K too large for the LDS-staged table (K * 128 * 8 B must fit 160 KB)");
    return WF_ERR_UNSUPPORTED;
  }
  const int64_t ntiles = (ncells + 15) / 16;
  const unsigned nb = (unsigned)std::min<int64_t>((ntiles + 7) / 8, 256);
  if (layout == 0) {
    auto kern = k_tsmm<NT, 0>;
    if (lds > 64 * 1024) WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, ncells, K, N, n0, in, phi, out);
  } else {
    auto kern = k_tsmm<NT, 1>;
    if (lds > 64 * 1024) WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, ncells, K, N, n0, in, phi, out);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("wf_tsmm launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

}  // namespace wf

using namespace wf;

extern "C" int wf_tsmm(int layout, int64_t ncells, int K, int N, const double* d_in, const double* d_phi,
                       double* d_out, void* stream)
{
  WF_REQUIRE(layout == 0 || layout == 1, "wf_tsmm: layout must be 0 (cell-major) or 1 (cell-minor)");
  WF_REQUIRE(ncells >= 0 && K > 0 && N > 0 && d_in && d_phi && d_out, "wf_tsmm: bad arguments");
  if (ncells == 0) return WF_OK;
  hipStream_t s = (hipStream_t)stream;
  // columns in passes of at most 128 (8 accumulator tiles per wave)
  for (int n0 = 0; n0 < N; n0 += 128) {
    const int nn = std::min(128, N - n0);
    int rc;
    if (nn <= 16) rc = launch_tsmm_t<1>(layout, ncells, K, N, n0, d_in, d_phi, d_out, s);
    else if (nn <= 32) rc = launch_tsmm_t<2>(layout, ncells, K, N, n0, d_in, d_phi, d_out, s);
    else if (nn <= 64) rc = launch_tsmm_t<4>(layout, ncells, K, N, n0, d_in, d_phi, d_out, s);
    else rc = launch_tsmm_t<8>(layout, ncells, K, N, n0, d_in, d_phi, d_out, s);
    if (rc != WF_OK) return rc;
  }
  return WF_OK;
}
