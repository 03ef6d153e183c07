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
#include <cstdlib>

#include "common.h"

namespace wf {

typedef double double4_t __attribute__((ext_vector_type(4)));

// k-steps per chunk: the A operands of a chunk are loaded one chunk ahead (two register
// sets), the B operands (LDS) one k-step ahead, so that no MFMA waits on the load issued
// just before it.  The LDS table is zero-padded to a whole number of chunks.
constexpr int kTsmmChunk = 8;

// NT: accumulator tiles (16 columns each) per wave.  (Splitting a cell tile's columns over
// several waves for small problems was measured at the reference shape 100 000 x 125:
// no gain -- 0.225 / 0.219 / 0.249 ms per pair of products with 1 / 2 / 4 parts.)
template <int NT, int LAYOUT, bool ACC>
__global__ __launch_bounds__(512) void k_tsmm(int64_t ncells, int Kfull, int N, int n0, int k0, int K,
                                              const double* __restrict__ in, const double* __restrict__ phi,
                                              double* __restrict__ out)
{
  // This launch covers table rows [k0, k0 + K) and columns [n0, n0 + 16 NT); with ACC the
  // accumulators start from out (the earlier row ranges of a K > 128 product).  ACC is a template
  // parameter: as a run-time branch the conditional loads made the compiler's vmcnt bookkeeping
  // conservative in the main loop and cost 15 % on the plain K <= 128 product.
  extern __shared__ __attribute__((aligned(16))) double sphi[];   // [4 * CH * nch][NP] (zero padded)
  // Row stride of the table in LDS: a wave's B-operand read (ds_read_b64) is served in two 32-lane
  // halves, each holding two k rows of 16 consecutive doubles; the halves are conflict-free when the
  // rows fall on different halves of the 64 banks, i.e. NP = 16 (mod 32) doubles.
  constexpr int NW = 16 * NT;                         // table columns in use
  constexpr int NP = NW + ((NW & 31) == 16 ? 0 : 16);
  constexpr int CH = kTsmmChunk;
  const int KT = (K + 3) / 4, nch = (KT + CH - 1) / CH, rows = 4 * CH * nch;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 15, lg = lane >> 4;
  // table -> LDS: 16 lanes per row segment, all loads of a row in flight together, no integer
  // division (this prologue is a fixed cost of every launch: 18 432 entries at the reference shape;
  // as a load -> store chain per entry it took ~35 us of the 106 us kernel)
  for (int k = t >> 4; k < rows; k += 32) {
    double v[NP / 16];
#pragma unroll
    for (int q = 0; q < NP / 16; ++q) {
      const int n = (t & 15) + 16 * q;
      v[q] = (k < K && n < NW && n0 + n < N) ? phi[(size_t)(k0 + k) * N + n0 + n] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NP / 16; ++q) sphi[k * NP + (t & 15) + 16 * q] = v[q];
  }
  __syncthreads();
  const double* bp = sphi + lg * NP + lc;   // B operand of k-step ks, tile nt: bp[(4 ks) NP + 16 nt]
  const int64_t ntiles = (ncells + 15) / 16;
  const int64_t tstride = (int64_t)gridDim.x * 8;
  // A operands stream from HBM as a flat sequence of (cell tile, k chunk) pairs; the loads of the
  // pair two steps ahead are always in flight, ACROSS tile boundaries, so a wave never starts a
  // tile by waiting for its first operands.
  auto load_a = [&](double (&a)[CH], int64_t tile, int ch) {
    tile += (ch / nch) * tstride;
    ch %= nch;
    const int64_t c = tile * 16 + lc;
    const bool ok = tile < ntiles && c < ncells;
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int k = 4 * (ch * CH + q) + lg;
      a[q] = (ok && k < K) ? (LAYOUT == 0 ? in[c * Kfull + k0 + k] : in[(int64_t)(k0 + k) * ncells + c]) : 0.0;
    }
  };
  double a_cur[CH], a_nxt[CH], a_nn[CH], pb[NT], pn[NT];
  {
    const int64_t first = (int64_t)blockIdx.x * 8 + wave;
    load_a(a_cur, first, 0);
    load_a(a_nxt, first, 1);
  }
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += tstride) {
    const int64_t c0 = tile * 16;
    double4_t acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = double4_t{0.0, 0.0, 0.0, 0.0};
    if (ACC) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (LAYOUT == 0) {
            const int64_t cc = c0 + lg + 4 * r;
            const int n = n0 + 16 * nt + lc;
            if (cc < ncells && n < N) acc[nt][r] = out[cc * N + n];
          } else {
            const int n = n0 + 16 * nt + lg + 4 * r;
            const int64_t cc = c0 + lc;
            if (cc < ncells && n < N) acc[nt][r] = out[(int64_t)n * ncells + cc];
          }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) pb[nt] = bp[16 * nt];
    for (int ch = 0; ch < nch; ++ch) {
      load_a(a_nn, tile, ch + 2);
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        // next k-step's B operands (the row after the table's last one is never read: ks + 1 < rows / 4)
        const int ksn = ch * CH + q + 1;
        if (ksn < nch * CH) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) pn[nt] = bp[(size_t)(4 * ksn) * NP + 16 * nt];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (LAYOUT == 0)
            acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[q], pb[nt], acc[nt], 0, 0, 0);   // D[cell][n]
          else
            acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[nt], a_cur[q], acc[nt], 0, 0, 0);   // D[n][cell]
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pb[nt] = pn[nt];
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        a_cur[q] = a_nxt[q];
        a_nxt[q] = a_nn[q];
      }
    }
    // D layout: row = lg + 4 r, col = lc
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (LAYOUT == 0) {
          const int64_t cc = c0 + lg + 4 * r;
          const int n = n0 + 16 * nt + lc;
          if (cc < ncells && n < N) out[cc * N + n] = acc[nt][r];
        } else {
          const int n = n0 + 16 * nt + lg + 4 * r;
          const int64_t cc = c0 + lc;
          if (cc < ncells && n < N) out[(int64_t)n * ncells + cc] = acc[nt][r];
        }
      }
  }
}

template <int NT>
static int launch_tsmm_t(int layout, int64_t ncells, int Kfull, int N, int n0, int k0, int K, const double* in,
                         const double* phi, double* out, hipStream_t s)
{

  const int KT = (K + 3) / 4, rows = 4 * kTsmmChunk * ((KT + kTsmmChunk - 1) / kTsmmChunk);
  const int NW = 16 * NT, NP = NW + ((NW & 31) == 16 ? 0 : 16);
  const size_t lds = (size_t)rows * NP * sizeof(double);
  if (lds > 160 * 1024) {
    set_error("wf_tsmm: row range too large for the LDS-staged table");
    return WF_ERR_UNSUPPORTED;
  }
  const int64_t ntiles = (ncells + 15) / 16;
  const unsigned nb = (unsigned)std::min<int64_t>((ntiles + 7) / 8, 256);
  auto go = [&](auto kern, size_t& set) -> int {
    if (lds > 64 * 1024 && lds > set) {
      WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), lds, s, ncells, Kfull, N, n0, k0, K, in, phi, out);
    return WF_OK;
  };
  static size_t set[4] = {0, 0, 0, 0};
  int rc;
  if (layout == 0) rc = k0 > 0 ? go(k_tsmm<NT, 0, true>, set[1]) : go(k_tsmm<NT, 0, false>, set[0]);
  else rc = k0 > 0 ? go(k_tsmm<NT, 1, true>, set[3]) : go(k_tsmm<NT, 1, false>, set[2]);
  if (rc != WF_OK) return rc;
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
  // Columns in passes of at most 128 (8 accumulator tiles of 16 per wave), the tiles spread evenly
  // over the passes (N = 216: 7 + 7 tiles rather than 8 + 6 with two idle); table rows in ranges
  // of at most 128 (the LDS-resident block is 128 x 144 doubles = 147 KB), whole 32-row chunks
  // first, later ranges accumulating onto out.  The P5..P7 tables of demo/gpu_operator
  // (216^2, 343^2, 512^2) take 4 / 9 / 16 launches.
  const int tiles = (N + 15) / 16, npass = (tiles + 7) / 8, tpp = (tiles + npass - 1) / npass;
  const int nk = (K + 127) / 128, kc = 32 * (((K + nk - 1) / nk + 31) / 32);
  for (int n0 = 0; n0 < N; n0 += 16 * tpp) {
    const int nt = std::min(tpp, (N - n0 + 15) / 16);
    for (int k0 = 0; k0 < K; k0 += kc) {
      const int kk = std::min(kc, K - k0);
      int rc;
      switch (nt) {
      case 1: rc = launch_tsmm_t<1>(layout, ncells, K, N, n0, k0, kk, d_in, d_phi, d_out, s); break;
      case 2: rc = launch_tsmm_t<2>(layout, ncells, K, N, n0, k0, kk, d_in, d_phi, d_out, s); break;
      case 3:
      case 4: rc = launch_tsmm_t<4>(layout, ncells, K, N, n0, k0, kk, d_in, d_phi, d_out, s); break;
      case 5: rc = launch_tsmm_t<5>(layout, ncells, K, N, n0, k0, kk, d_in, d_phi, d_out, s); break;
      case 6: rc = launch_tsmm_t<6>(layout, ncells, K, N, n0, k0, kk, d_in, d_phi, d_out, s); break;
      case 7: rc = launch_tsmm_t<7>(layout, ncells, K, N, n0, k0, kk, d_in, d_phi, d_out, s); break;
      default: rc = launch_tsmm_t<8>(layout, ncells, K, N, n0, k0, kk, d_in, d_phi, d_out, s); break;
      }
      if (rc != WF_OK) return rc;
    }
  }
  return WF_OK;
}
