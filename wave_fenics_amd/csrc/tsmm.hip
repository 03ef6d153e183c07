// Tall-skinny dense matmul (TSMM, k >> m ~ n in the reference's README.md:39):
//     out[cell][n] = sum_k in[cell][k] * phi[k][n]
// the two cublasDgemm calls of demo/gpu_tsmm/main.cpp:49-52 (100000 x 125 times
// 125 x 125) and the "B" / "B^T" products of demo/gpu_operator/main.cpp:149-155,
// on v_mfma_f64_16x16x4_f64.  A block of phi (<= 128 rows x 128 columns, <= 150 KB)
// is staged once per workgroup in LDS; one wave owns 16 cells and keeps the block's
// accumulator tiles (<= 8) in registers; a persistent grid of one 512-thread
// workgroup per CU walks the cells.  Larger tables run as several launches: column
// passes, and row ranges that accumulate onto out (wf_tsmm below).
//
// layout 0 (cell-major, demo/gpu_operator): in[cell*K + k], out[cell*N + n];
//          A operand = in tile (16 cells x 4 k), B operand = phi.
// layout 1 (cell-minor = the column-major arrays of demo/gpu_tsmm, lda = ldc =
//          ncells): in[k*ncells + cell], out[n*ncells + cell]; the product is
//          formed transposed (A = phi^T, B = in) so that loads and stores stay
//          contiguous in the cell index.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace wf {

typedef double double4_t __attribute__((ext_vector_type(4)));

// k-steps per chunk: the A operands of a chunk are loaded two chunks ahead (three rotating
// register sets), the B operands (LDS) one k-step ahead, so that no MFMA waits on the load
// issued just before it.  The LDS table is zero-padded to a whole number of chunks.
constexpr int kTsmmChunk = 8;
#ifndef WF_TSMM_WAVES
#define WF_TSMM_WAVES 8
#endif
constexpr int kTsmmWaves = WF_TSMM_WAVES;   // waves per workgroup (one workgroup per CU)

// Diagnostic (tools/tsmm_trace.hip defines WF_TSMM_TRACE and includes this file): per-wave
// timestamps of the prologue and of each cell tile, 100 MHz constant clock.
#ifdef WF_TSMM_TRACE
__device__ unsigned long long g_tsmm_trace[256 * kTsmmWaves * 8];
__device__ unsigned long long g_tsmm_trace_clk[256 * kTsmmWaves * 8];   // shader-clock counter at the same points
#define WF_TR(i)                                                                   \
  if ((threadIdx.x & 63) == 0 && (i) < 8) {                                        \
    g_tsmm_trace[(blockIdx.x * kTsmmWaves + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); \
    g_tsmm_trace_clk[(blockIdx.x * kTsmmWaves + (threadIdx.x >> 6)) * 8 + (i)] = clock64();  \
  }
#else
#define WF_TR(i)
#endif
#ifndef WF_TSMM_PRIO
#define WF_TSMM_PRIO 1
#endif
#ifndef WF_TSMM_ABLATE   // diagnostic bit mask: 1 = no A loads in the loop, 2 = no B (LDS) reads, 4 = no stores
#define WF_TSMM_ABLATE 0
#endif

// Wave-uniform base pointer (SGPR pair) + 32-bit per-lane byte offset: the form the global_load /
// global_store "saddr" encoding takes, with no 64-bit vector address arithmetic.
template <typename T>
__device__ __forceinline__ T* lane_ptr(T* base, uint32_t byte_offset)
{
  using B = std::conditional_t<std::is_const_v<T>, const char, char>;
  return reinterpret_cast<T*>(reinterpret_cast<B*>(base) + byte_offset);
}

// NT: accumulator tiles (16 columns each) per wave.  Measured and rejected: splitting a cell
// tile's columns over several waves for small problems (no gain at 100 000 x 125), and 12 waves
// per CU instead of 8 (WF_TSMM_WAVES, the kernel fits 3 waves per SIMD: 1.31 vs 1.33 ms at 1 M
// cells, 0.149 vs 0.150 ms at 100 000).  tools/tsmm_trace.hip shows the shader clock at
// 1.92-2.15 GHz while this kernel runs on random operands (2.25 GHz with loads and stores
// ablated, faster again on all-zero data): the 78.6 TFLOP/s figure the fractions are quoted
// against assumes 2.4 GHz.
template <int NT, int LAYOUT, bool ACC>
__global__ __launch_bounds__(64 * kTsmmWaves) void k_tsmm(int64_t ncells, int Kfull, int N, int n0, int k0, int K,
                                              int64_t main_tiles, int tail_ct,
                                              const double* __restrict__ in, const double* __restrict__ phi,
                                              double* __restrict__ out)
{
  // This launch covers table rows [k0, k0 + K) and columns [n0, n0 + 16 NT); with ACC the
  // accumulators start from out (the earlier row ranges of a K > 128 product).  ACC is a template
  // parameter: as a run-time branch the conditional loads made the compiler's vmcnt bookkeeping
  // conservative in the main loop and cost 15 % on the plain K <= 128 product.
  extern __shared__ __attribute__((aligned(16))) double sphi[];   // [4 * CH * nch + 4][NP] (zero padded)
  // Row stride of the table in LDS: a wave's B-operand read (ds_read_b64) is served in two 32-lane
  // halves, each holding two k rows of 16 consecutive doubles; the halves are conflict-free when the
  // rows fall on different halves of the 64 banks, i.e. NP = 16 (mod 32) doubles.
  constexpr int NW = 16 * NT;                         // table columns in use
  constexpr int NP = NW + ((NW & 31) == 16 ? 0 : 16);
  constexpr int CH = kTsmmChunk;
  const int KT = (K + 3) / 4, nch = (KT + CH - 1) / CH, rows = 4 * CH * nch;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 15, lg = lane >> 4;
  WF_TR(0);
  const double* bp = sphi + lg * NP + lc;   // B operand of k-step ks, tile nt: bp[(4 ks) NP + 16 nt]
  const int64_t ntiles = (ncells + 15) / 16;
  const int64_t tstride = (int64_t)gridDim.x * kTsmmWaves;
  // The main loop is branch-free straight-line code per chunk of CH k-steps (measured with
  // tools/tsmm_trace.hip at 1 M x 125: bounds-check branches around every A load and B read, the
  // register copies of a rolling operand buffer -- which wait for the loads just issued -- and the
  // per-element store guards kept the MFMA pipe idle 40 % of the time, with one or two waves per SIMD
  // alike).  So: loads use clamped addresses instead of guards (rows past ncells are computed on the
  // last cell and not stored; k past K multiplies a zero table row by a finite value of the same
  // cell); the A operands sit in three register sets whose roles rotate by unrolling the flat
  // sequence of (cell tile, chunk) pairs three times -- the set two chunks ahead is always in
  // flight, across tile boundaries, and nothing is ever copied; the B operands ping-pong between
  // two register sets inside the unrolled chunk; stores take a guard-free path for whole tiles.
  // A-operand address = wave-uniform base (SGPRs) + 32-bit per-lane offset; the lane part carries the
  // clamps (cell index inside a partial last tile, k inside a partial last k-step).
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int64_t cmax = ncells - 1;
  const int ksmax = (K - 1) >> 2;
  // Cell tile of wave w of workgroup b in round r: r * (8 nb) + w * nb + b -- wave-major, so that the
  // tiles of a partial last round land on different CUs (and SIMDs) instead of filling the first
  // workgroups' waves: at 100 000 cells (6250 tiles, 3.05 rounds) no SIMD then runs more than 7 tiles.
  const int64_t first_tile = (int64_t)swave * gridDim.x + blockIdx.x;
  int64_t pf_tile = first_tile;   // prefetch cursor (cell tile, chunk)
  int pf_ch = 0;
  const double* pf_base;
  uint32_t pf_lc;
  auto pf_setup = [&]() {
    const int64_t te = std::min(pf_tile, ntiles - 1);
    const int lcp = std::min(lc, (int)(cmax - te * 16));
    if (LAYOUT == 0) {
      pf_base = in + te * 16 * Kfull + k0;
      pf_lc = (uint32_t)(lcp * Kfull);
    } else {
      pf_base = in + (int64_t)k0 * ncells + te * 16;
      pf_lc = (uint32_t)lcp;
    }
  };
  auto a_load = [&](int q) -> double {
    const int kse = std::min(pf_ch * CH + q, ksmax);
    const int lgp = std::min(lg, K - 1 - 4 * kse);
    if (LAYOUT == 0) return *lane_ptr(pf_base + 4 * kse, (pf_lc + (uint32_t)lgp) * 8u);
    return *lane_ptr(pf_base + (int64_t)(4 * kse) * ncells, ((uint32_t)lgp * (uint32_t)ncells + pf_lc) * 8u);
  };
  auto pf_advance = [&]() {
    if (++pf_ch == nch) {
      pf_ch = 0;
      pf_tile += tstride;
    }
  };
  const uint32_t st_lane = LAYOUT == 0 ? (uint32_t)(lg * N + lc) * 8u : ((uint32_t)lg * (uint32_t)ncells + (uint32_t)lc) * 8u;
  double a0[CH], a1[CH], a2[CH], pb[2][NT];
  // table -> LDS, loads first: all of a thread's table loads are issued in one burst BEFORE the first A-operand
  // prefetches (vmcnt retires in order: with the A loads in front, the LDS stores of the table waited for the
  // HBM latency of the operands as well -- 5.5 us of prologue at the reference shape).  rows + 4 <= 132, i.e.
  // at most 5 rows per thread; 4 spare rows because the B prefetch runs one k-step past the table.
  constexpr int RPT = (4 * 4 * CH + 4 + 4 * kTsmmWaves - 1) / (4 * kTsmmWaves);
  double tv[RPT][NP / 16];
#pragma unroll
  for (int it = 0; it < RPT; ++it) {
    const int k = (t >> 4) + it * 4 * kTsmmWaves;
#pragma unroll
    for (int q = 0; q < NP / 16; ++q) {
      const int n = (t & 15) + 16 * q;
      tv[it][q] = (k < K && n < NW && n0 + n < N) ? phi[(size_t)(k0 + k) * N + n0 + n] : 0.0;
    }
  }
  pf_setup();
#pragma unroll
  for (int q = 0; q < CH; ++q) a0[q] = a_load(q);
  pf_advance();
  pf_setup();
#pragma unroll
  for (int q = 0; q < CH; ++q) a1[q] = a_load(q);
  pf_advance();
  // table registers -> LDS (the A prefetches issued after the table loads stay in flight: loads retire in order)
#pragma unroll
  for (int it = 0; it < RPT; ++it) {
    const int k = (t >> 4) + it * 4 * kTsmmWaves;
    if (k < rows + 4) {
#pragma unroll
      for (int q = 0; q < NP / 16; ++q) sphi[k * NP + (t & 15) + 16 * q] = tv[it][q];
    }
  }
  __syncthreads();
  WF_TR(1);
  [[maybe_unused]] int trace_slot = 2;
  int64_t tile = first_tile;
  int ch = 0;
  double4_t acc[NT];
  // one chunk: MFMAs on `cur`, loads of the chunk two ahead into `nn`
  auto chunk = [&](const double (&cur)[CH], double (&nn)[CH]) {
    const int64_t c0 = tile * 16;
#if WF_TSMM_PRIO
    // the two waves of a SIMD take turns at the higher issue priority, chunk by chunk (the arbiter otherwise
    // favours the older wave: tools/tsmm_trace.hip shows wave 0 done with its tiles 15 us before wave 4)
    if ((ch ^ (swave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
    if (ch == 0) {
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
      for (int nt = 0; nt < NT; ++nt) pb[0][nt] = bp[16 * nt];
    }
    const double* bq = bp + (size_t)(4 * ch * CH) * NP;
    pf_setup();
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      if (!(WF_TSMM_ABLATE & 2)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pb[(q + 1) & 1][nt] = bq[(4 * (q + 1)) * NP + 16 * nt];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (LAYOUT == 0)
          acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[q], pb[q & 1][nt], acc[nt], 0, 0, 0);   // D[cell][n]
        else
          acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(pb[q & 1][nt], cur[q], acc[nt], 0, 0, 0);   // D[n][cell]
      }
      if (!(WF_TSMM_ABLATE & 1)) nn[q] = a_load(q);
    }
    pf_advance();
    if (++ch == nch) {
      // D layout: row = lg + 4 r, col = lc
      if (!((WF_TSMM_ABLATE & 4) && acc[0][0] != 1.2345)) {
        // Stores: one per-lane offset for the whole kernel on a wave-uniform base.  Layout 0 needs four
        // row pointers (the column tiles are immediate offsets); layout 1 walks one pointer down the
        // 4 NT output rows -- computing them independently costs 64 VGPRs.  Guards are uniform
        // branches wherever the condition is (N = 125 has a partial last column tile in every cell
        // tile); per-lane guards only remain on partial tiles.
        const bool cells_whole = c0 + 16 <= ncells;
        if (LAYOUT == 0) {
          double* row[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) row[r] = lane_ptr(out + (c0 + 4 * r) * N + n0, st_lane);
          const bool lane_row_ok[4] = {c0 + lg < ncells, c0 + lg + 4 < ncells, c0 + lg + 8 < ncells, c0 + lg + 12 < ncells};
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int nb = n0 + 16 * nt;
            if (cells_whole && nb + 16 <= N) {
#pragma unroll
              for (int r = 0; r < 4; ++r) row[r][16 * nt] = acc[nt][r];
            } else if (nb + lc < N) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (lane_row_ok[r]) row[r][16 * nt] = acc[nt][r];
            }
          }
        } else {
          double* pv = lane_ptr(out + (int64_t)n0 * ncells + c0, st_lane);
          const bool lane_cell_ok = c0 + lc < ncells;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int nb = n0 + 16 * nt + 4 * r;   // rows nb + lg
              if (cells_whole && nb + 4 <= N) {
                *pv = acc[nt][r];
              } else if (lane_cell_ok && nb + lg < N) {
                *pv = acc[nt][r];
              }
              pv += 4 * ncells;
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
      WF_TR(trace_slot);
      ++trace_slot;
      ch = 0;
      tile += tstride;
    }
  };
  while (tile < main_tiles) {
    chunk(a0, a2);
    if (tile >= main_tiles) break;
    chunk(a1, a0);
    if (tile >= main_tiles) break;
    chunk(a2, a1);
  }
  // Tail: the cell tiles [main_tiles, ntiles) of a partial last round, split along the columns into units of
  // CT column tiles so that every SIMD gets at most one unit.  The kernel is MFMA-bound per SIMD (a tile is
  // 32 k-steps x 8 column tiles x 64.6 cycles = 7.9 us at 2.1 GHz), and at the reference's 100 000 cells
  // (6250 tiles = 6 per SIMD + 106) the SIMDs that ran a seventh whole tile set the kernel's duration:
  // tools/tsmm_trace.hip showed 69 us with 55 us of MFMA work on those SIMDs.  As 848 eighth-tiles the
  // remainder costs every SIMD 1 us.  A unit loads the tile's A operands in one burst (<= 32 per lane).
  if (tail_ct > 0) {
    auto tail = [&](auto ct_tag) {
      constexpr int CT = decltype(ct_tag)::value;
      constexpr int UP = (NT + CT - 1) / CT;   // units per cell tile
      const int64_t unit = first_tile;
      if (unit >= (ntiles - main_tiles) * UP) return;
      const int64_t tl = main_tiles + unit / UP;
      const int ntb = __builtin_amdgcn_readfirstlane((int)(unit % UP) * CT);
      const int64_t c0 = tl * 16;
      const int lcp = std::min(lc, (int)(cmax - c0));
      const double* base = LAYOUT == 0 ? in + c0 * Kfull + k0 : in + (int64_t)k0 * ncells + c0;
      const uint32_t off = LAYOUT == 0 ? (uint32_t)(lcp * Kfull) : (uint32_t)lcp;
      double a[4 * CH];
#pragma unroll
      for (int q = 0; q < 4 * CH; ++q) {
        const int kse = std::min(q, ksmax);
        const int lgp = std::min(lg, K - 1 - 4 * kse);
        a[q] = LAYOUT == 0 ? *lane_ptr(base + 4 * kse, (off + (uint32_t)lgp) * 8u)
                           : *lane_ptr(base + (int64_t)(4 * kse) * ncells, ((uint32_t)lgp * (uint32_t)ncells + off) * 8u);
      }
      double4_t tacc[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        tacc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
        if (ACC && ntb + c < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (LAYOUT == 0) {
              const int64_t cc = c0 + lg + 4 * r;
              const int n = n0 + 16 * (ntb + c) + lc;
              if (cc < ncells && n < N) tacc[c][r] = out[cc * N + n];
            } else {
              const int n = n0 + 16 * (ntb + c) + lg + 4 * r;
              const int64_t cc = c0 + lc;
              if (cc < ncells && n < N) tacc[c][r] = out[(int64_t)n * ncells + cc];
            }
          }
        }
      }
      const double* bt = bp + 16 * ntb;
#pragma unroll
      for (int q = 0; q < 4 * CH; ++q) {
        if (q < nch * CH) {   // the LDS table has 4 CH nch (+ 4) rows
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            if (ntb + c < NT) {
              const double b = bt[(4 * q) * NP + 16 * c];
              if (LAYOUT == 0) tacc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b, tacc[c], 0, 0, 0);
              else tacc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a[q], tacc[c], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (ntb + c < NT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (LAYOUT == 0) {
              const int64_t cc = c0 + lg + 4 * r;
              const int n = n0 + 16 * (ntb + c) + lc;
              if (cc < ncells && n < N) out[cc * N + n] = tacc[c][r];
            } else {
              const int n = n0 + 16 * (ntb + c) + lg + 4 * r;
              const int64_t cc = c0 + lc;
              if (cc < ncells && n < N) out[(int64_t)n * ncells + cc] = tacc[c][r];
            }
          }
        }
      }
    };
    if (tail_ct == 1) tail(std::integral_constant<int, 1>{});
    else if (tail_ct == 2) tail(std::integral_constant<int, 2>{});
    else tail(std::integral_constant<int, 4>{});
    WF_TR(trace_slot);
  }
}

template <int NT>
static int launch_tsmm_t(int layout, int64_t ncells, int Kfull, int N, int n0, int k0, int K, const double* in,
                         const double* phi, double* out, hipStream_t s)
{

  const int KT = (K + 3) / 4, rows = 4 * kTsmmChunk * ((KT + kTsmmChunk - 1) / kTsmmChunk);
  const int NW = 16 * NT, NP = NW + ((NW & 31) == 16 ? 0 : 16);
  const size_t lds = (size_t)(rows + 4) * NP * sizeof(double);
  if (lds > 160 * 1024) {
    set_error("wf_tsmm: row range too large for the LDS-staged table");
    return WF_ERR_UNSUPPORTED;
  }
  const int64_t ntiles = (ncells + 15) / 16;
  const unsigned nb = (unsigned)std::min<int64_t>(ntiles, 256);   // one workgroup per CU; small problems spread over CUs first
  // partial last round: split its tiles along the columns when every wave then gets at most one unit
  const int64_t nwaves = (int64_t)nb * kTsmmWaves, rem = ntiles % nwaves;
  int64_t main_tiles = ntiles;
  int tail_ct = 0;
  if (rem > 0 && KT <= 4 * kTsmmChunk)
    for (int ct : {1, 2, 4})
      if (ct < NT && rem * ((NT + ct - 1) / ct) <= nwaves) {
        tail_ct = ct;
        main_tiles = ntiles - rem;
        break;
      }
  // (the attribute is set on every launch that needs more than 64 KB: a per-process cache of "already set" is wrong for a
  // second device in the process and racy between host threads, and the call costs well under a microsecond)
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024)
      WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * kTsmmWaves), lds, s, ncells, Kfull, N, n0, k0, K, main_tiles, tail_ct, in, phi, out);
    return WF_OK;
  };
  int rc;
  if (layout == 0) rc = k0 > 0 ? go(k_tsmm<NT, 0, true>) : go(k_tsmm<NT, 0, false>);
  else rc = k0 > 0 ? go(k_tsmm<NT, 1, true>) : go(k_tsmm<NT, 1, false>);
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
  WF_REQUIRE(ncells < (int64_t(1) << 27), "wf_tsmm: at most 2^27 cells per call (32-bit lane offsets)");
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
