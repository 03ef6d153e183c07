// Dense (non-tensor-product) stiffness operator for affine simplex cells:
// the reference's skernel (common/operators.hpp:113-133) with arbitrary dense
// tables dphi[3][nq][nd], i.e. the "non-tensor-product operator path" of
// BASELINE.json configs[4] (P4 tetrahedra, nd = 35, nq = 64).
//
//   W[3 nq][cells] = T[3 nq][nd] . U[nd][cells]            (MFMA, f64 16x16x4)
//   F[dir][q][c]   = -c0^2 * sum_e clamp(w_q C_c)[dir][e] W[e][q][c]   (lane-local)
//   Y[nd][cells]   = T^T . F                                (MFMA, f64 16x16x4)
//
// This is the one place of the engine where the contraction is GEMM-shaped with a
// k-dimension worth a matrix core (k = nd = 35 and k = 3 nq = 192), so it runs on
// v_mfma_f64_16x16x4_f64.  One wave owns 16 cells (the N dimension); W stays in
// registers between the two products: the f64 accumulator layout
// (row = (lane>>4) + 4*reg, col = lane&15) is exactly the B-operand layout of
// k-step `reg`, so F feeds the second product with no LDS round trip.
// The table T lives in LDS; geometry is 6 doubles per affine cell.
// Gather/scatter ("permute/scatter stress"): per batch of 64 cells the host
// computes the list of unique dofs; x is read once per unique dof into LDS, the
// cell results are summed per unique dof in LDS (ds_add_f64) and leave with one
// global atomic per unique dof.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "common.h"

namespace wf {

typedef double double4_t __attribute__((ext_vector_type(4)));

// Diagnostic build (tools/dense_trace.sh): per-wave timestamps of the phases of the first batches
// of every workgroup, 100 MHz constant clock.
#ifdef WF_DENSE_TRACE
constexpr int kDenseTraceIters = 12, kDenseTraceSlots = 6;
__device__ unsigned long long g_dense_trace[512 * 4 * kDenseTraceIters * kDenseTraceSlots + 512 * 4];   // + HW_ID | XCC_ID << 32 per wave
#define WF_DTR(slot)                                                                                         \
  if ((threadIdx.x & 63) == 0 && trace_it < kDenseTraceIters && blockIdx.x < 512)                             \
  g_dense_trace[((blockIdx.x * 4 + (threadIdx.x >> 6)) * kDenseTraceIters + trace_it) * kDenseTraceSlots + (slot)] = wall_clock64()
#else
#define WF_DTR(slot)
#endif

// xt::isclose clamp to -1/0/1 (precomputation.hpp:105-107); the common case
// (no clamp zone hit) costs two compares.
__device__ __forceinline__ double clamp101d(double v)
{
  const double a = fabs(v);
  if (a <= 1e-8) return 0.0;
  if (fabs(a - 1.0) <= 1e-8 + 1e-5) return v < 0.0 ? -1.0 : 1.0;
  return v;
}

// QT = ceil(nq/16), KT = ceil(nd/4), DT = ceil(nd/16); KP = row pitch of T in LDS.
// LDS bank layout (64 banks x 4 B; a ds_read_b64 is served 32 lanes at a time): the first product
// reads T[row(lc)][4 ks + lg] (16 rows x 2 columns per half wave), the second T[row(lg)][16 dt + lc]
// (2 rows x 16 columns).  With KP = 4 KT + 2 (twice an odd number) and the 16 quadrature points of
// a slab assigned to MFMA rows in the order pi(j) = (j >> 1) + 8 (j & 1), both patterns touch 32
// distinct 8-byte banks: rows pi(.) KP cover all even bank pairs, and the two rows of a half wave
// in the second product are 8 KP = 16 (mod 32) doubles apart.  (KP = 4 KT + 1 with rows in natural
// order had 2-way conflicts on a third of the lanes in both products.)
__host__ __device__ constexpr int dense_pitch(int KT) { return 4 * KT + 2; }
__device__ __forceinline__ int dense_row_perm(int j) { return (j >> 1) + 8 * (j & 1); }
// NU: unique dofs of a batch per thread (numax <= NU * 64 NW), a compile-time bound for the
// register-staged gather of the NEXT batch.
// XR: output rows past the last whole 16-row tile that are NOT given an MFMA tile of their own
// (nd = 16 (DT - 1) + XR, XR <= 4; 0 = pad nd to 16 DT as before).  P4 tetrahedra have nd = 35:
// a third tile would spend 16 rows of matrix-core time on 3 rows of result, a fifth of the
// kernel's MFMA cycles.  Those rows are instead accumulated lane-locally with VALU FMAs in the
// shadow of the MFMAs (each lane owns the quadrature points q = lg mod 4 of its cell) and summed
// over the four lane groups once per batch.
template <int QT, int KT, int DT, int NW, int NU, int XR>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_stiffness_dense(int nd, int nq, int nbatch, int numax,
                                                         const double* __restrict__ Tg,      // [3*16*QT][KP] padded table
                                                         const double* __restrict__ wq,      // [16*QT] weights (0 beyond nq)
                                                         const double* __restrict__ Cg,      // [nbatch*16*NW][6]
                                                         const uint32_t* __restrict__ locP,  // [nbatch][ceil(KT/2)][4][16*NW]: local index of dof 4(2j)+lg | that of dof 4(2j+1)+lg << 16
                                                         const int32_t* __restrict__ uoff,   // [nbatch+1]
                                                         const int32_t* __restrict__ uniq,   // unique dofs of all batches
                                                         const uint8_t* __restrict__ clampb, // [nbatch] 1: some w_q C_c of the batch lies in a clamp window
                                                         double coeff, int do_clamp, const double* __restrict__ x,
                                                         double* __restrict__ y, int ablate_arg, int stagger)
{
  [[maybe_unused]] const int ablate = WF_ABLATE_FLAGS(ablate_arg);
  constexpr int NQP = 16 * QT, KP = dense_pitch(KT), NT = 64 * NW, NCB = 16 * NW;
  constexpr int DTM = XR > 0 ? DT - 1 : DT, XN = XR > 0 ? XR : 1, KT2 = (KT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* T = smem;                 // [3*NQP][KP]
  double* sw = T + 3 * NQP * KP;    // [NQP]
  double* Xu = sw + NQP;            // [numax]  x at the unique dofs of the batch
  double* Yu = Xu + numax;          // [numax]  sum of the cell results per unique dof

  (void)nq;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 15, lg = lane >> 4;
  // Two workgroups share a CU (one wave of each per SIMD; workgroups b and b + 256, per HW_ID in
  // tools/dense_trace.py).  Experiment knob WF_DENSE_STAGGER (default 0 = off): start the second
  // resident workgroup half a batch late so that its gather / scatter phases fall under the other's
  // MFMA section.  It gains nothing (0.600 / 0.607 / 0.614 ms at 0 / 1 / 2 sleeps before the other
  // changes of the round): the phase offset persists, but a lone wave keeps the MFMA pipe only 57 %
  // busy -- barely more than half of the saturated pipe two overlapping MFMA sections share.
  if (stagger > 0 && ((blockIdx.x / 256) & 1)) {
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);   // 127 * 64 cycles each
  }
  for (int p = t; p < 3 * NQP * KP; p += NT) T[p] = Tg[p];
  for (int p = t; p < NQP; p += NT) sw[p] = wq[p];

  // Software pipeline over the batches of this (persistent) workgroup: while batch b is in its
  // MFMA section, the gather of batch b + G is in flight in registers (x values xr, local
  // indices and geometry) and the unique-dof list of batch b + 2G is being fetched (uq), so the
  // only memory latency left on the critical path is the scatter's.  Inside the MFMA section no
  // global load is consumed (loads retire in order).
  const int G = gridDim.x;
  // Per-batch scalars (unique-dof range, clamp flag) are fetched with VECTOR loads one batch before
  // they are needed: as scalar loads their (cold, ~1 us) latency sat in front of the MFMA section,
  // and a pending s_load also turns every LDS wait into lgkmcnt(0).
  auto vgpr_zero = [&]() {   // an opaque per-lane 0: pointer + this stays a global pointer but lives in VGPRs
    int z = 0;
    asm volatile("" : "+v"(z));
    return z;
  };
  auto vload_i32 = [&](const int32_t* p) { return p[vgpr_zero()]; };
  struct Range {
    int32_t lo, hi;   // uoff[b], uoff[b + 1]  (b clamped to a valid batch; `live` says whether b < nbatch)
    bool live;
  };
  auto load_range = [&](int b) {
    const int bb = b < nbatch ? b : nbatch - 1;
    return Range{vload_i32(uoff + bb), vload_i32(uoff + bb + 1), b < nbatch};
  };
  // All prefetch loads are unconditional, on clamped indices (a thread past the end of a batch's
  // unique-dof list re-reads the last entry, a batch past the end re-reads the last batch): guards
  // around loads become branches, and at every join the compiler's vmcnt bookkeeping falls back to
  // waiting for loads it has just issued.  Whether an entry is live is decided where it is used
  // (nu_*: number of unique dofs of the batch).
  auto load_uq = [&](int32_t (&uq)[NU], const Range& rg) {
#pragma unroll
    for (int m = 0; m < NU; ++m) uq[m] = uniq[min(rg.lo + t + NT * m, rg.hi - 1)];
  };
  auto count_of = [&](const Range& rg) { return rg.live ? rg.hi - rg.lo : 0; };
  auto load_clamp = [&](int b) -> int {
    return clampb[(b < nbatch ? b : nbatch - 1) + vgpr_zero()];
  };
  auto load_x = [&](double (&xr)[NU], const int32_t (&uq)[NU]) {
#pragma unroll
    for (int m = 0; m < NU; ++m) xr[m] = (ablate & 4) ? 1.0 + m : x[uq[m]];
  };
  // Local indices travel as packed pairs: loaded as 16-bit values the compiler packs them two to a
  // register right after the load, i.e. waits for the prefetch it has just issued.
  auto load_cell = [&](uint32_t (&loc)[KT2], double (&C)[6], int b) {
    const int bb = b < nbatch ? b : nbatch - 1;
#pragma unroll
    for (int j = 0; j < KT2; ++j) loc[j] = locP[(((size_t)bb * KT2 + j) * 4 + lg) * NCB + wave * 16 + lc];
    const double* cp = Cg + ((size_t)bb * NCB + wave * 16 + lc) * 6;
#pragma unroll
    for (int e = 0; e < 6; ++e) C[e] = cp[e];
  };
  int32_t uq_cur[NU], uq_nxt[NU];     // unique dofs (this thread's share) of the batch being computed / of the next one
  double xr[NU];
  uint32_t loc[KT2], locn[KT2];
  auto loc_of = [&](int ks) -> uint32_t { return (loc[ks >> 1] >> (16 * (ks & 1))) & 0xffffu; };
  double C[6], Cn[6];
  // prologue: first batch straight into LDS, indices of the second
  int nu_cur, nu_nxt;   // live entries of uq_cur / uq_nxt
  {
    const Range r0 = load_range(blockIdx.x), r1 = load_range(blockIdx.x + G);
    load_uq(uq_cur, r0);
    nu_cur = count_of(r0);
    load_x(xr, uq_cur);
    load_cell(loc, C, blockIdx.x);
    load_uq(uq_nxt, r1);
    nu_nxt = count_of(r1);
  }
  Range rg2 = load_range(blockIdx.x + 2 * G);   // range of the batch whose unique-dof list is fetched next
  int clamp_cur = load_clamp(blockIdx.x);
#pragma unroll
  for (int m = 0; m < NU; ++m) {
    const int u = t + NT * m;
    if (u < numax) {
      Xu[u] = xr[m];
      Yu[u] = 0.0;
    }
  }

  [[maybe_unused]] int trace_it = 0;
#ifdef WF_DENSE_TRACE
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 512)
    g_dense_trace[512 * 4 * kDenseTraceIters * kDenseTraceSlots + blockIdx.x * 4 + (threadIdx.x >> 6)] =
        (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
#endif
  for (int batch = blockIdx.x; batch < nbatch; batch += G) {
    // the -1/0/1 clamp of G (precomputation.hpp:105-107) is the identity unless a product w_q C_c
    // falls into one of its windows; the host marks the batches where that happens (same
    // double-precision products), every other batch skips ~200 VALU instructions per slab
    const bool clamp_here = do_clamp && clamp_cur;
    __syncthreads();   // Xu holds this batch's x values, Yu is zero
    WF_DTR(0);

    // B operands of the first product: this lane's dof values, one per k-step
    double ub[KT];
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      const double v = (ablate & 32) ? 1.0 + ks : Xu[loc_of(ks)];   // padded k: local index 0, a valid entry
      ub[ks] = (4 * ks + lg) < nd ? v : 0.0;
    }
    WF_DTR(1);
    // gather of the next batch (registers) and index list of the one after it
    int32_t uq_nn[NU];
    load_cell(locn, Cn, batch + G);
    load_uq(uq_nn, rg2);
    const int nu_nn = count_of(rg2);
    const Range rg3 = load_range(batch + 3 * G);
    const int clamp_nxt = load_clamp(batch + G);

    double4_t Y[DTM];
#pragma unroll
    for (int dt = 0; dt < DTM; ++dt) Y[dt] = double4_t{0.0, 0.0, 0.0, 0.0};
    double yx[XN];
#pragma unroll
    for (int i = 0; i < XN; ++i) yx[i] = 0.0;

    // one 16-point slab of quadrature points at a time keeps the register
    // footprint small (3 accumulator tiles instead of 3*QT) -> more waves per SIMD
#pragma unroll 1
    for (int qt = 0; qt < ((ablate & 8) ? 0 : QT); ++qt) {
      // ---- W = T . U for the rows (dir, 16 qt .. 16 qt + 15) ---------------------
      // The A operands come from LDS.  Written as "read, then MFMA" the compiler waits for each
      // read right before the MFMA that uses it (ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma), so
      // every k-step pays the LDS latency; explicit register double buffers keep the reads of
      // step s + 1 in flight behind the MFMAs of step s.
      double4_t W[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) W[e] = double4_t{0.0, 0.0, 0.0, 0.0};
      const double* Ta = T + (16 * qt + dense_row_perm(lc)) * KP + lg;   // + e NQP KP + 4 ks
      double a3[3][3];   // rotating operand sets, reads two k-steps ahead (no copies: a copy waits for the read just issued)
#pragma unroll
      for (int e = 0; e < 3; ++e) a3[0][e] = Ta[e * NQP * KP];
      if (KT > 1) {
#pragma unroll
        for (int e = 0; e < 3; ++e) a3[1][e] = Ta[e * NQP * KP + 4];
      }
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        if (ks + 2 < KT) {
#pragma unroll
          for (int e = 0; e < 3; ++e) a3[(ks + 2) % 3][e] = Ta[e * NQP * KP + 4 * (ks + 2)];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
        for (int e = 0; e < 3; ++e) W[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3[ks % 3][e], ub[ks], W[e], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the x gather of the next batch (NU scattered 8-byte loads per thread, up to 64 cache lines per
      // wave instruction) goes out behind the first product of the first slab, not in front of the
      // MFMA section: a wave blocks in the issue of such a burst (the CU's outstanding-request
      // capacity), and here its 27 queued MFMAs cover that
      if (qt == 0) load_x(xr, uq_nxt);
      // first A operands of the second product: in flight during the lane-local geometry product
      // (row = e NQP + 16 qt + pi(4 r + lg), column d = 16 dt + lc; columns >= 4 KT are never stored: pad = 0)
      // accumulator register r of lane group lg is quadrature point pi(4 r + lg) = 2 r + pi(lg) of the slab
      const double* Tb = T + (16 * qt + dense_row_perm(lg)) * KP + lc;   // + (e NQP + 2 r) KP + 16 dt
      const double* Tx = T + (16 * qt + dense_row_perm(lg)) * KP + 16 * DTM;   // extra rows: + (e NQP + 2 r) KP + i (no lc: broadcast reads)
      double b3[3][DTM], x3[3][XN];   // operands of steps st, st + 1, st + 2 of the second product
      auto load_b = [&](int st) {
        const int e = st / 4, r = st % 4;
#pragma unroll
        for (int dt = 0; dt < DTM; ++dt)
          b3[st % 3][dt] = (XR > 0 || (16 * dt + lc) < 4 * KT) ? Tb[(e * NQP + 2 * r) * KP + 16 * dt] : 0.0;
        if (XR > 0) {
#pragma unroll
          for (int i = 0; i < XN; ++i) x3[st % 3][i] = Tx[(e * NQP + 2 * r) * KP + i];
        }
      };
      load_b(0);
      load_b(1);
      // ---- F = coeff * G W (lane-local: the three directions of one (q, cell) share lane and register)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 2 * r + dense_row_perm(lg);
        const double w = sw[q];
        double g00 = w * C[0], g01 = w * C[1], g02 = w * C[2], g11 = w * C[3], g12 = w * C[4], g22 = w * C[5];
        if (clamp_here) {   // precomputation.hpp:105-107
          g00 = clamp101d(g00); g01 = clamp101d(g01); g02 = clamp101d(g02);
          g11 = clamp101d(g11); g12 = clamp101d(g12); g22 = clamp101d(g22);
        }
        const double w0 = W[0][r], w1 = W[1][r], w2 = W[2][r];
        W[0][r] = coeff * (g00 * w0 + g01 * w1 + g02 * w2);
        W[1][r] = coeff * (g01 * w0 + g11 * w1 + g12 * w2);
        W[2][r] = coeff * (g02 * w0 + g12 * w1 + g22 * w2);
      }
      // ---- Y += T^T . F : accumulator register r of a tile is the B operand of k-step r
#pragma unroll
      for (int st = 0; st < 12; ++st) {
        const int e = st / 4, r = st % 4;
        if (st + 2 < 12) load_b(st + 2);
        const double b = W[e][r];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < DTM; ++dt) Y[dt] = __builtin_amdgcn_mfma_f64_16x16x4f64(b3[st % 3][dt], b, Y[dt], 0, 0, 0);
        if (XR > 0) {
#pragma unroll
          for (int i = 0; i < XN; ++i) yx[i] = fma(x3[st % 3][i], b, yx[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    WF_DTR(2);
    // ---- per-batch accumulation over unique dofs, then one atomic per unique dof
#pragma unroll
    for (int dt = 0; dt < DTM; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ks = 4 * dt + r;   // d = 16 dt + lg + 4 r = 4 ks + lg
        if (!(ablate & 2) && ks < KT && 4 * ks + lg < nd) atomicAdd(&Yu[loc_of(ks < KT ? ks : 0)], Y[dt][r]);
      }
    if (XR > 0) {
      // row 16 DTM + i of cell lc: sum of the four lane groups' partial sums; lane group i scatters it
      // (its loc[4 DTM] is the local index of dof 16 DTM + lg)
      double mine = 0.0;
#pragma unroll
      for (int i = 0; i < XN; ++i) {
        double v = yx[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lg == i) mine = v;
      }
      if (!(ablate & 2) && lg < XR) atomicAdd(&Yu[loc_of(4 * DTM < KT ? 4 * DTM : 0)], mine);
    }
    WF_DTR(3);
    __syncthreads();   // Yu complete; every wave has taken its operands out of Xu
    WF_DTR(4);
    // Order matters: every consumer of a prefetched register (xr, uq_nn, locn, Cn, the next ranges)
    // comes BEFORE the first global atomic.  On gfx9 loads and atomics share vmcnt and the compiler
    // waits for vmcnt(0) once both kinds are pending, so an atomic issued earlier put its whole
    // round trip (~0.5 us, five times per batch) in front of the next consumer.
    double yv[NU];
    int32_t uq_old[NU];
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      const int u = t + NT * m;
      yv[m] = u < numax ? Yu[u] : 0.0;
      uq_old[m] = uq_cur[m];
    }
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      const int u = t + NT * m;
      if (u < numax) {   // same thread, same entries: next batch's x values in, sums back to zero
        Xu[u] = xr[m];
        Yu[u] = 0.0;
      }
      uq_cur[m] = uq_nxt[m];
      uq_nxt[m] = uq_nn[m];
    }
#pragma unroll
    for (int j = 0; j < KT2; ++j) loc[j] = locn[j];
#pragma unroll
    for (int e = 0; e < 6; ++e) C[e] = Cn[e];
    rg2 = rg3;
    clamp_cur = clamp_nxt;
    const int nu_old = nu_cur;
    nu_cur = nu_nxt;
    nu_nxt = nu_nn;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      if (t + NT * m < nu_old) {
        if (ablate & 1) {
          if (yv[m] == 1.2345e300) y[uq_old[m]] = yv[m];
        } else {
          unsafeAtomicAdd(&y[uq_old[m]], yv[m]);
        }
      }
    }
    WF_DTR(5);
    ++trace_it;
  }
}

struct DenseOpData {
  int nd = 0, nq = 0, QT = 0, KT = 0, DT = 0, nbatch = 0, numax = 0, nw = 4;
  double* d_T = nullptr;
  double* d_w = nullptr;
  double* d_C = nullptr;
  uint32_t* d_locP = nullptr;
  int32_t* d_uoff = nullptr;
  int32_t* d_uniq = nullptr;
  uint8_t* d_clampb = nullptr;
  size_t bytes = 0;
};

void dense_free(DenseOpData* d)
{
  if (!d) return;
  (void)hipFree(d->d_T);
  (void)hipFree(d->d_w);
  (void)hipFree(d->d_C);
  (void)hipFree(d->d_locP);
  (void)hipFree(d->d_uoff);
  (void)hipFree(d->d_uniq);
  (void)hipFree(d->d_clampb);
  delete d;
}

template <typename Tp>
static int up(Tp** p, const std::vector<Tp>& h, size_t* total)
{
  *p = nullptr;
  if (h.empty()) return WF_OK;
  WF_HIP_CHECK(hipMalloc((void**)p, h.size() * sizeof(Tp)));
  WF_HIP_CHECK(hipMemcpy(*p, h.data(), h.size() * sizeof(Tp), hipMemcpyHostToDevice));
  *total += h.size() * sizeof(Tp);
  return WF_OK;
}

// Host setup: padded table, per-cell affine geometry C = |det J| K K^T, per-batch
// unique-dof lists and local indices.
int dense_setup(int nd, int nq, int ncells, int ndofs, const int32_t* dofmap, const double* dphi,
                const double* weights, const double* xverts, const int32_t* geom_dofmap, DenseOpData** out)
{
  std::unique_ptr<DenseOpData, void (*)(DenseOpData*)> d(new DenseOpData, dense_free);
  d->nd = nd;
  d->nq = nq;
  d->QT = (nq + 15) / 16;
  d->KT = (nd + 3) / 4;
  d->DT = (nd + 15) / 16;
  d->nw = 4;

  const int NCB = 16 * d->nw;
  const int NQP = 16 * d->QT, KP = dense_pitch(d->KT);
  std::vector<double> T((size_t)3 * NQP * KP, 0.0), w(NQP, 0.0);
  for (int dir = 0; dir < 3; ++dir)
    for (int q = 0; q < nq; ++q)
      for (int k = 0; k < nd; ++k) T[((size_t)dir * NQP + q) * KP + k] = dphi[((size_t)dir * nq + q) * nd + k];
  for (int q = 0; q < nq; ++q) w[q] = weights[q];
  const int nbatch = (ncells + NCB - 1) / NCB;
  d->nbatch = nbatch;
  std::vector<double> C((size_t)nbatch * NCB * 6, 0.0);
  for (int c = 0; c < ncells; ++c) {
    const int32_t* v = geom_dofmap + (size_t)c * 4;
    double J[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) J[i * 3 + j] = xverts[(size_t)v[j + 1] * 3 + i] - xverts[(size_t)v[0] * 3 + i];
    const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6])
                       + J[2] * (J[3] * J[7] - J[4] * J[6]);
    const double id = 1.0 / det;
    double K[9];
    K[0] = (J[4] * J[8] - J[5] * J[7]) * id;
    K[1] = (J[2] * J[7] - J[1] * J[8]) * id;
    K[2] = (J[1] * J[5] - J[2] * J[4]) * id;
    K[3] = (J[5] * J[6] - J[3] * J[8]) * id;
    K[4] = (J[0] * J[8] - J[2] * J[6]) * id;
    K[5] = (J[2] * J[3] - J[0] * J[5]) * id;
    K[6] = (J[3] * J[7] - J[4] * J[6]) * id;
    K[7] = (J[1] * J[6] - J[0] * J[7]) * id;
    K[8] = (J[0] * J[4] - J[1] * J[3]) * id;
    const double ad = std::fabs(det);
    auto kk = [&](int a, int b) {
      double s = 0.0;
      for (int m = 0; m < 3; ++m) s += (K[a * 3 + m] * ad) * K[b * 3 + m];
      return s;
    };
    double* cc = &C[(size_t)c * 6];
    cc[0] = kk(0, 0); cc[1] = kk(0, 1); cc[2] = kk(0, 2); cc[3] = kk(1, 1); cc[4] = kk(1, 2); cc[5] = kk(2, 2);
  }
  const int KT2 = (d->KT + 1) / 2;
  std::vector<uint32_t> locP((size_t)nbatch * KT2 * 4 * NCB, 0);
  std::vector<int32_t> uoff(nbatch + 1, 0), uniq;
  uniq.reserve((size_t)ncells * nd / 2);
  std::vector<int32_t> tmp;
  int numax = 0;
  for (int b = 0; b < nbatch; ++b) {
    const int c0 = b * NCB, nc = std::min(NCB, ncells - c0);
    tmp.assign(dofmap + (size_t)c0 * nd, dofmap + (size_t)(c0 + nc) * nd);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    const int nu = (int)tmp.size();
    if (nu > 65535) {
      set_error("dense_setup: more than 65535 unique dofs in a batch");
      return WF_ERR_UNSUPPORTED;
    }
    numax = std::max(numax, nu);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < nd; ++k) {
        const int32_t g = dofmap[(size_t)(c0 + c) * nd + k];
        const int u = (int)(std::lower_bound(tmp.begin(), tmp.end(), g) - tmp.begin());
        const int ks = k / 4, lg = k % 4;
        locP[(((size_t)b * KT2 + ks / 2) * 4 + lg) * NCB + c] |= (uint32_t)u << (16 * (ks & 1));
      }
    uniq.insert(uniq.end(), tmp.begin(), tmp.end());
    uoff[b + 1] = (int32_t)uniq.size();
  }
  d->numax = std::max(numax, 1);
  (void)ndofs;
  // batches in which the clamp is not the identity
  std::vector<uint8_t> clampb(nbatch, 0);
  for (int c = 0; c < ncells; ++c) {
    bool hit = false;
    for (int q = 0; q < nq && !hit; ++q)
      for (int e = 0; e < 6 && !hit; ++e) {
        const double a = std::fabs(weights[q] * C[(size_t)c * 6 + e]);
        hit = (a > 0.0 && a <= 1e-8) || (a != 1.0 && std::fabs(a - 1.0) <= 1e-8 + 1e-5);
      }
    if (hit) clampb[c / NCB] = 1;
  }
  int rc;
  if ((rc = up(&d->d_T, T, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_w, w, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_C, C, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_locP, locP, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_uoff, uoff, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_uniq, uniq, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_clampb, clampb, &d->bytes)) != WF_OK) return rc;
  *out = d.release();
  return WF_OK;
}

size_t dense_bytes(const DenseOpData* d) { return d ? d->bytes : 0; }

template <int QT, int KT, int DT, int NW, int NU, int XR>
static int launch_dense_t(const DenseOpData* d, double coeff, int do_clamp, const double* d_x, double* d_y,
                          hipStream_t s)
{
  constexpr int NQP = 16 * QT, KP = dense_pitch(KT);
  const size_t lds = ((size_t)3 * NQP * KP + NQP + 2 * d->numax) * sizeof(double);
  if (lds > 160 * 1024) {
    set_error("stiffness_dense: tables do not fit LDS");
    return WF_ERR_UNSUPPORTED;
  }
  auto kern = k_stiffness_dense<QT, KT, DT, NW, NU, XR>;
  if (lds > 64 * 1024)
    WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
#ifdef WF_DIAG   // diagnostic overrides, compiled in by tools/diag_build.sh only
  const int wgs_per_cu = std::getenv("WF_DENSE_WGS") ? std::atoi(std::getenv("WF_DENSE_WGS")) : 2;
  const int ablate = std::getenv("WF_ABLATE") ? std::atoi(std::getenv("WF_ABLATE")) : 0;
  const int stagger = std::getenv("WF_DENSE_STAGGER") ? std::atoi(std::getenv("WF_DENSE_STAGGER")) : 0;
#else
  const int wgs_per_cu = 2, ablate = 0, stagger = 0;
#endif
  const unsigned nb = (unsigned)std::min(d->nbatch, 256 * wgs_per_cu);   // persistent: the table is staged into LDS once per workgroup
  hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * NW), lds, s, d->nd, d->nq, d->nbatch, d->numax, d->d_T, d->d_w, d->d_C,
                     d->d_locP, d->d_uoff, d->d_uniq, d->d_clampb, coeff, do_clamp, d_x, d_y,
                     ablate, stagger);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("stiffness_dense launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

// NU = 5 covers the unique dofs of 64 well-numbered P4 cells (1154 on the Kuhn box); 9 is the worst case 64 * 35
#define WF_DENSE_CASE(Q, K, D, X)                                                                                \
  if (d->QT == Q && d->KT == K && d->DT == D && (X == 0 || d->nd == 16 * (D - 1) + X))                            \
    return d->numax <= 5 * 256 ? launch_dense_t<Q, K, D, 4, 5, X>(d, coeff, do_clamp, d_x, d_y, s)                \
                               : launch_dense_t<Q, K, D, 4, 9, X>(d, coeff, do_clamp, d_x, d_y, s);

// Compiled shapes: Lagrange P1..P4 on the tetrahedron with the m = p Gauss-Jacobi
// rule (nd, nq) = (4,1) (10,8) (20,27) (35,64), plus P4 with the m = 3 rule.
int launch_stiffness_dense(const DenseOpData* d, double coeff, int do_clamp, const double* d_x, double* d_y,
                           hipStream_t s)
{
  if (d->nbatch == 0) return WF_OK;
  WF_DENSE_CASE(1, 1, 1, 0)
  WF_DENSE_CASE(1, 3, 1, 0)
  WF_DENSE_CASE(2, 5, 2, 4)   // P3: nd = 20 = 16 + 4
  WF_DENSE_CASE(2, 5, 2, 0)
  WF_DENSE_CASE(4, 9, 3, 3)   // P4: nd = 35 = 32 + 3
  WF_DENSE_CASE(4, 9, 3, 0)
  WF_DENSE_CASE(2, 9, 3, 3)
  WF_DENSE_CASE(2, 9, 3, 0)
  set_error("stiffness_dense: (nd, nq) shape not compiled (supported: tetrahedron P1..P4)");
  return WF_ERR_UNSUPPORTED;
}

}  // namespace wf

#ifdef WF_DENSE_TRACE
extern "C" int wf_debug_dense_trace(unsigned long long* host, size_t n)
{
  void* sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(wf::g_dense_trace)) != hipSuccess) return -1;
  if (n > sizeof(wf::g_dense_trace) / 8) n = sizeof(wf::g_dense_trace) / 8;
  return hipMemcpy(host, sym, n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
