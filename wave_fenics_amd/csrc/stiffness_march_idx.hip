// k_stiffness_march_idx<P, BX, BY>: the marching stiffness kernel (stiffness_march.hip) for
// an ARBITRARY dofmap -- StiffnessOperator::operator(), common/operators.hpp:183-200, on any
// conforming hexahedral mesh whose cells link up like a lattice (generic_plan.cpp), whatever
// the cell order and dof numbering.
//
// Same structure as the box kernel: a 256-thread workgroup owns a column of BX x BY cells and
// marches through <= lz layers; the next layer's geometry (48 B per point) and x planes are in
// flight while the current layer is computed; the z-shared plane is carried in a register; the
// finished planes go to y with one fp64 atomic per tile dof.  The only difference is where
// ADDRESSES come from: the column's dof tile [P lz + 1][P BY + 1][P BX + 1] is a table of dof
// offsets (PATTERN, shared by all work items with the same relative numbering) that is staged
// in LDS once per work item -- inside the layer loop the only global loads are then the
// prefetches (vmcnt retires loads in order: an index load consumed inside a layer would have to
// wait for the geometry prefetch issued before it).
// HBM-bound; algorithmic bytes ncells (48 nq + 4 nd) + 16 ndofs; the index
// table costs 4 P (P BX + 1)(P BY + 1) / (BX BY) bytes per cell when it is not L2-resident.
//
// Compiled for the stiffness operator at P <= 4 (P >= 5: stiffness_march_ks.hip; the dense mass
// operator on the same columns: mass_march.hip).
#include <type_traits>

#include "stiffness_core.h"

namespace wf {

constexpr int OP_STIFFNESS = 0, OP_MASS = 1;

// Diagnostic build (tools/march_trace.sh -DWF_IDX_TRACE): per-wave timestamps of the phases of the
// first layers of the first 512 workgroups, 100 MHz constant clock.
#ifdef WF_IDX_TRACE
constexpr int kIdxTraceIters = 12, kIdxTraceSlots = 6;
__device__ unsigned long long g_idx_trace[512 * 4 * kIdxTraceIters * kIdxTraceSlots];
#define WF_ITR(slot)                                                                                       \
  if ((threadIdx.x & 63) == 0 && trace_it < kIdxTraceIters && blockIdx.x < 512)                             \
  g_idx_trace[((blockIdx.x * 4 + (threadIdx.x >> 6)) * kIdxTraceIters + trace_it) * kIdxTraceSlots + (slot)] = wall_clock64()
#else
#define WF_ITR(slot)
#endif

template <int OP, int P, int BX, int BY>
__global__ __launch_bounds__(256, 2) void k_march_idx(int lz, int tile_size, const int32_t* __restrict__ item_base,
                                                                const int32_t* __restrict__ item_pattern,
                                                                const int32_t* __restrict__ item_layers,
                                                                const int32_t* __restrict__ pat_off,
                                                                const void* __restrict__ geom,
                                                                const double* __restrict__ dD, DMat dm, double coeff,
                                                                const double* __restrict__ x, double* __restrict__ y,
                                                                const int32_t* __restrict__ items)
{
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  constexpr int CB = BX * BY, NT = CB * n2;
  constexpr int TX = P * BX + 1, TY = P * BY + 1, TP = TX * TY;
  constexpr int NPOS = (P * TP + 255) / 256;        // flush / x-prefetch positions per thread
  constexpr int NPOS0 = ((P + 1) * TP + 255) / 256; // prologue x positions per thread
  constexpr int NCP = (TP + 255) / 256;             // positions of one plane per thread
  static_assert(NT <= 256, "column does not fit a 256-thread workgroup");

  __shared__ __attribute__((aligned(16))) double Ux[(P + 1) * TP];   // x planes of the layer
  // results of the layer's cells, summed where they share a face (ds_add_f64): planes 0..P-1 of the tile
  // (see stiffness_march.hip); reused for the carried plane in the epilogue
  __shared__ __attribute__((aligned(16))) double O[P * TP > CB * n2 ? P * TP : CB * n2];
  __shared__ __attribute__((aligned(16))) double Fr[CB * nd];
  __shared__ __attribute__((aligned(16))) double Fs[CB * nd];
  __shared__ __attribute__((aligned(16))) double sD[n * n];
  extern __shared__ __attribute__((aligned(16))) int32_t sIdx[];     // [(P lz + 1)][TP] dof offsets of the column, -1 = none

  const int t = threadIdx.x;
  // work item; an optional item list selects a subset (interior / interface split of the ghost exchange)
  const size_t item = items ? (size_t)items[blockIdx.x] : (size_t)blockIdx.x;
  const int nl = item_layers[item];
  const size_t gbase = (size_t)item_base[item];
  const int32_t* __restrict__ pat = pat_off + (size_t)item_pattern[item] * tile_size;
  const bool active = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;
  const int lx = cl % BX, ly = cl / BX;

  // geometry registers: stiffness 3 x double2 per point (G upper triangle), mass 1 double (detJ w)
  constexpr int GW = OP == OP_STIFFNESS ? 3 : 1;
  using GT = typename std::conditional<OP == OP_STIFFNESS, double2, double>::type;
  // two register sets that swap roles from layer to layer (the layer loop is unrolled by two): a
  // copy gcur = gnext ends up at the loop's back edge, behind the flush, and waits for the atomics
  GT gA[n][GW], gB[n][GW];
  auto load_g = [&](GT (&g)[n][GW], int l, int k0 = 0, int k1 = P + 1) {
    if constexpr (OP == OP_STIFFNESS) {
      const double2* gp = static_cast<const double2*>(geom) + ((item * lz + l) * n * 3) * (size_t)NT + (t < NT ? t : NT - 1);
#pragma unroll
      for (int k = 0; k < n; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (k >= k0 && k < k1) g[k][p] = load_stream(gp + (size_t)(k * 3 + p) * NT);
    } else {
      const double* gp = static_cast<const double*>(geom) + ((item * lz + l) * n) * (size_t)NT + (t < NT ? t : NT - 1);
#pragma unroll
      for (int k = 0; k < n; ++k)
        if (k >= k0 && k < k1) g[k][0] = __builtin_nontemporal_load(gp + (size_t)k * NT);
    }
  };
  // stiffness at P >= 4: the next layer's geometry is requested in three instalments over the layer
  // (see stiffness_march.hip)
  constexpr bool kSpread = OP == OP_STIFFNESS && P >= 4;
  constexpr int G1 = kSpread ? (n + 1) / 3 : n, G2 = kSpread ? (2 * n + 1) / 3 : n;
  // index table first (L2-resident for regular numberings), then the first layer's geometry and
  // x planes together: one HBM latency in the prologue, not two (loads retire in order)
  if (t < n * n) sD[t] = dD[t];
  for (int e = t; e < P * TP; e += 256) O[e] = 0.0;
  for (int e = t; e < (P * nl + 1) * TP; e += 256) sIdx[e] = pat[e];
  __syncthreads();
  load_g(gA, 0);
  // ---- prologue: x planes 0..P of the first layer -> LDS ------------------------
#pragma unroll
  for (int m = 0; m < NPOS0; ++m) {
    const int pos = t + 256 * m;
    if (pos < (P + 1) * TP) {
      const int32_t off = sIdx[pos];
      Ux[pos] = off >= 0 ? x[gbase + off] : 0.0;
    }
  }
  __syncthreads();

  double carry = 0.0;
  const double* Uc = Ux + (P * ly) * TX + P * lx;

  [[maybe_unused]] int trace_it = 0;
  // (idle threads load the last thread's geometry instead of branching around the loads; `has_next` is a compile-time
  // constant of the layer body at P4, a run-time flag below: stiffness_march.hip)
  auto layer = [&](auto hn_tag, GT (&gcur)[n][GW], GT (&gnext)[n][GW], int l) {
    const bool has_next = hn_tag;
    WF_ITR(0);
    // (a) next layer's x planes and geometry: in flight during this layer's arithmetic
    // (unconditional loads on clamped addresses -- dead entries and positions past the tile read the item's first
    // dof, the last layer its own planes again: a guard around a load is a branch, and behind it the compiler waits
    // for every memory operation still pending; what is live is decided where the registers are consumed)
    double xn[NPOS];
    const int ln = has_next ? l + 1 : l;
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      const int pos = t + 256 * m;
      const int32_t off = pos < P * TP ? sIdx[(P * ln + 1) * TP + pos] : -1;
      xn[m] = x[gbase + (off >= 0 ? off : 0)];
    }
    if (has_next) load_g(gnext, l + 1, 0, G1);

    WF_ITR(1);
    // (b) element kernels of the layer
    double out[n];
    if constexpr (OP == OP_STIFFNESS) {
      double ft[n];
      stiffness_phase1<P>(Uc, TP, TX, Fr + cl * nd, Fs + cl * nd, sD, dm, gcur, coeff, i, j, active, ft);
      __syncthreads();
      if (kSpread && has_next) load_g(gnext, l + 1, G1, G2);
      stiffness_phase2<P>(Fr + cl * nd, Fs + cl * nd, sD, dm, ft, i, j, active, out);
    }
    WF_ITR(2);
    double xcp[NCP];
#pragma unroll
    for (int m = 0; m < NCP; ++m) {
      const int pos = t + 256 * m;
      xcp[m] = pos < TP ? Ux[P * TP + pos] : 0.0;
    }
    if (active) {
      out[0] += carry;        // z-shared plane: partial sum of the layer below
      carry = out[P];
      double* To = O + (P * ly + j) * TX + P * lx + i;
#pragma unroll
      for (int k = 0; k < P; ++k) __hip_atomic_fetch_add(To + k * TP, out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    WF_ITR(3);

    // (c) rotate the x planes, (d) combine the cells of the layer (fixed
    // order) and add the finished planes to y.  For the stiffness operator the rotation -- the
    // consumers of the prefetched registers -- comes before the flush, so that its wait does not
    // include this layer's atomics (loads and atomics share vmcnt; see stiffness_march.hip).
    auto rotate = [&]() {
    if (has_next) {
#pragma unroll
      for (int m = 0; m < NCP; ++m) {
        const int pos = t + 256 * m;
        if (pos < TP) Ux[pos] = xcp[m];
      }
#pragma unroll
      for (int m = 0; m < NPOS; ++m) {
        const int pos = t + 256 * m;
        if (pos < P * TP) Ux[TP + pos] = sIdx[(P * ln + 1) * TP + pos] >= 0 ? xn[m] : 0.0;
      }
    }
    };
    auto flush = [&]() {
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      const int pos = t + 256 * m;
      if (pos >= P * TP) continue;
      const int32_t off = sIdx[(P * l) * TP + pos];
      const double v = O[pos];
      O[pos] = 0.0;
      if (off < 0) continue;
      unsafeAtomicAdd(y + gbase + off, v);
    }
    };
    // Measured both ways for both operators (cfg-size meshes): stiffness P4 0.245 -> 0.239 ms with the
    // rotation first; the dense mass (a third of the geometry bytes, shorter layers) is faster with
    // the flush first, P2 / P4 0.166 / 0.157 -> 0.158 / 0.147 ms.
    constexpr bool flush_first = OP == OP_MASS;
    if constexpr (flush_first) {
      flush();
      WF_ITR(4);
      rotate();
    } else {
      rotate();
      __builtin_amdgcn_sched_barrier(0);
      WF_ITR(4);
      if (kSpread && has_next) load_g(gnext, l + 1, G2, n);
      flush();
    }
    WF_ITR(5);
    ++trace_it;

    __syncthreads();
  };
  using HasNext = std::integral_constant<bool, true>;
  using IsLast = std::integral_constant<bool, false>;
  for (int l = 0; l < nl; l += 2) {
    if constexpr (P < 4) {
      layer(l + 1 < nl, gA, gB, l);
      if (l + 1 < nl) layer(l + 2 < nl, gB, gA, l + 1);
    } else if (l + 1 < nl) {
      layer(HasNext{}, gA, gB, l);
      if (l + 2 < nl)
        layer(HasNext{}, gB, gA, l + 1);
      else
        layer(IsLast{}, gB, gA, l + 1);
    } else {
      layer(IsLast{}, gA, gB, l);
    }
  }

  // ---- epilogue: the last (carried) plane ------------------------------------
  if (active) O[cl * n2 + ji] = carry;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NCP; ++m) {
    const int pos = t + 256 * m;
    if (pos >= TP) continue;
    const int32_t off = sIdx[(P * nl) * TP + pos];
    if (off < 0) continue;
    const int J = pos / TX, I = pos % TX;
    const int ca = I / P, ia = I % P, cb = J / P, jb = J % P;
    double v = 0.0;
    if (cb < BY) {
      if (ca < BX) v += O[(cb * BX + ca) * n2 + jb * n + ia];
      if (ia == 0 && ca > 0) v += O[(cb * BX + ca - 1) * n2 + jb * n + P];
    }
    if (jb == 0 && cb > 0) {
      if (ca < BX) v += O[((cb - 1) * BX + ca) * n2 + P * n + ia];
      if (ia == 0 && ca > 0) v += O[((cb - 1) * BX + ca - 1) * n2 + P * n + P];
    }
    unsafeAtomicAdd(y + gbase + off, v);
  }
}

template <int OP, int P, int BX, int BY>
static int launch_t(const MarchPlanDev& pd, const double* d_G6blk, const double* d_D, const DMat& dm, double coeff,
                    const double* d_x, double* d_y, const int32_t* d_items, int nitems, hipStream_t s)
{
  const int nwg = d_items ? nitems : pd.nitems;
  if (nwg == 0) return WF_OK;
  const size_t dyn = (size_t)pd.tile_size * sizeof(int32_t);
  // static + dynamic LDS may exceed the 64 KB default limit.  The attribute is per device; it is set on
  // every launch (a cheap host call) instead of being cached in a process-wide static, which a second
  // device or a concurrent first launch from another host thread would not see.
  WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_march_idx<OP, P, BX, BY>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  hipLaunchKernelGGL((k_march_idx<OP, P, BX, BY>), dim3((unsigned)nwg), dim3(256), dyn, s, pd.lz, pd.tile_size,
                     pd.d_item_base, pd.d_item_pattern, pd.d_item_layers, pd.d_pat_off,
                     static_cast<const void*>(d_G6blk), d_D, dm, coeff, d_x, d_y, d_items);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("stiffness_march_idx launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

// the stiffness operator runs the k-split kernel (stiffness_march_ks.hip) at P >= 5
// P >= 5 runs the k-split kernel.  P4, any dofmap, cfg2: k_march_idx<4,5,2> 0.204 ms, k_march_ks<4,5,1,true> 0.208 ms
// (-DWF_IDX_KS_MINP=4 selects the latter)
#ifndef WF_IDX_KS_MINP
#define WF_IDX_KS_MINP 5
#endif
static bool march_idx_uses_ks(int P) { return P >= WF_IDX_KS_MINP; }

// the cross-sections with BX * BY == floor(256 / n^2) cells (geometry batch layout)
void march_idx_shape(int kind, int P, int* bx, int* by)
{
  static const int kBX[8] = {0, 8, 7, 4, 5, 7, 5, 2}, kBY[8] = {0, 8, 4, 4, 2, 1, 1, 2};
  if (kind == OP_KIND_MASS) {
    mass_march_shape(P, bx, by);
    return;
  }
  if (march_idx_uses_ks(P)) {
    march_ks_shape(P, bx, by);   // keeps a compiled (*bx, *by), else the degree's default
    return;
  }
  *bx = kBX[P];
  *by = kBY[P];
}

// LDS of one workgroup: the kernel's static arrays + the index tile
size_t march_idx_lds_bytes(int kind, int P, int BX, int BY, int lz)
{
  if (kind == OP_KIND_MASS) return mass_march_lds_bytes(P, BX, BY, lz);
  if (march_idx_uses_ks(P)) return march_ks_lds_bytes(P, BX, BY, lz, true);
  const int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY, TP = (P * BX + 1) * (P * BY + 1);
  return (size_t)((P + 1) * TP + CB * P * n2 + 2 * CB * nd + n * n) * sizeof(double) + (size_t)(P * lz + 1) * TP * sizeof(int32_t);
}

// LDS a workgroup may use so that as many fit a CU as the kernel's registers allow: the dense-mass kernel runs two
// 256-thread workgroups per CU; the k-split stiffness kernel one 512-thread, two (P >= 5) or three 256-thread ones
size_t march_idx_lds_budget(int kind, int P, int BX, int BY)
{
  if (kind != OP_KIND_STIFFNESS || !march_idx_uses_ks(P)) return (size_t)80 * 1024;
  const int n = P + 1, NTc = BX * BY * n * n, WG = 2 * (((NTc + 63) / 64) * 64);
  const int per_cu = WG >= 512 ? (P <= 3 ? 2 : 1) : (P <= 4 ? 768 / WG : 512 / WG);
  return (size_t)158 * 1024 / per_cu;
}

int launch_stiffness_march_idx(int P, const MarchPlanDev& pd, const double* d_G6blk, const double* d_D,
                               const DMat& dm, double coeff, const double* d_x, double* d_y, const int32_t* d_items,
                               int nitems, hipStream_t s)
{
  if (march_idx_uses_ks(P))
    return launch_stiffness_march_ks_idx(P, pd.bx, pd.by, pd, d_G6blk, d_D, dm, coeff, d_x, d_y, d_items, nitems, s);
  switch (P) {
    case 1: return launch_t<OP_STIFFNESS, 1, 8, 8>(pd, d_G6blk, d_D, dm, coeff, d_x, d_y, d_items, nitems, s);
    case 2: return launch_t<OP_STIFFNESS, 2, 7, 4>(pd, d_G6blk, d_D, dm, coeff, d_x, d_y, d_items, nitems, s);
    case 3: return launch_t<OP_STIFFNESS, 3, 4, 4>(pd, d_G6blk, d_D, dm, coeff, d_x, d_y, d_items, nitems, s);
    case 4: return launch_t<OP_STIFFNESS, 4, 5, 2>(pd, d_G6blk, d_D, dm, coeff, d_x, d_y, d_items, nitems, s);
  }
  set_error("stiffness_march_idx: degree must be 1..7");
  return WF_ERR_UNSUPPORTED;
}

}  // namespace wf

#ifdef WF_IDX_TRACE
extern "C" int wf_debug_idx_trace(unsigned long long* host, size_t n)
{
  void* sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(wf::g_idx_trace)) != hipSuccess) return -1;
  if (n > sizeof(wf::g_idx_trace) / 8) n = sizeof(wf::g_idx_trace) / 8;
  return hipMemcpy(host, sym, n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
