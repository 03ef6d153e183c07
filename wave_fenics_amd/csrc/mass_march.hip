// k_mass_march<P, BX, BY>: dense (sum-factorised) mass operator y += Phi^T diag(det J w) Phi x with a square
// 1-D table Phi = phi1 (x) phi1 (x) phi1 on the lattice columns of ANY dofmap -- MassOperator::apply,
// common/cuda/mass.hpp:76-95 + mass_kernel.cu:5-37, the DGEMM pair of demo/gpu_operator/main.cpp:144-160
// (k >> m ~ n), marching through the columns like the stiffness kernels (plan of generic_plan.cpp).
//
// Structure:
//  * a cell lives inside ONE wave (n^2 = (P+1)^2 lanes; floor(64 / n^2) cells per wave), so the five passes of
//    the element kernel exchange their data through wave-private LDS scratch with no workgroup barrier --
//    LDS operations of one wave execute in order;
//  * every pass is a "pencil" contraction in registers, in place: a lane reads the n values of one line of the
//    cell, multiplies by the 1-D table held in VGPRs (n^2 doubles, the same in every lane: as scalar operands
//    they would need 2 n^2 SGPRs) and writes the n results over them -- 2 n LDS accesses per lane and pass:
//        X  : lane (j, k)   A[k][j][qi]   = sum_i  phi[qi][i] U[k][j][i]
//        Y  : lane (qi, k)  A[k][qj][qi]  = sum_j  phi[qj][j] A[k][j][qi]
//        Z  : lane (qi, qj) w[qk] = detJ[qk] sum_k phi[qk][k] A[k][qj][qi];  A[k][qj][qi] = sum_qk phi[qk][k] w[qk]
//        Y^T: lane (qi, k)  A[k][j][qi]   = sum_qj phi[qj][j] A[k][qj][qi]
//        X^T: lane (j, k)   T[k][..][..] += sum_qi phi[qi][i] A[k][j][qi]      (ds_add_f64 into the layer's tile)
//  * the cells of a layer sum their results in ONE LDS tile (ds_add_f64), so the flush of a position is one LDS
//    read, one index read and one global atomic; the flush of layer l - 1 is issued between the first passes of
//    layer l (a burst of atomics at the layer's end stalled the loads issued behind it);
//  * the x planes, the result tile and the carried z-shared plane are double-buffered in LDS: ONE workgroup
//    barrier per layer;
//  * the item's index table streams through a ring of 4 P + 1 planes in LDS (global -> register with the x
//    prefetch, -> ring at the layer's end), so the LDS footprint does not depend on the segment length.
// tools/mass_trace.sh/.py: per-wave phase timeline.  P6 before these four changes (private per-cell results gathered
// by the flush, whole table staged in LDS: lz = 4): layer 5.0 us = prefetch issue 1.0 + passes 1.4 + x -> LDS 0.5 +
// flush 1.8; now about 3.2 us.  HBM-bound by its bytes (8 B of det J w per point + x + y), latency-bound in practice.
#include <type_traits>

#include "stiffness_core.h"

namespace wf {

// Diagnostic build (tools/mass_trace.sh): per-wave timestamps of the phases of the first layers of the first 512
// workgroups, 100 MHz constant clock.
#ifdef WF_MASS_TRACE
constexpr int kMassTraceIters = 12, kMassTraceSlots = 10;
__device__ unsigned long long g_mass_trace[512 * 4 * kMassTraceIters * kMassTraceSlots];
#define WF_MSTR(slot)                                                                 \
  if ((threadIdx.x & 63) == 0 && l < kMassTraceIters && blockIdx.x < 512 && (threadIdx.x >> 6) < 4) \
  g_mass_trace[((blockIdx.x * 4 + (threadIdx.x >> 6)) * kMassTraceIters + l) * kMassTraceSlots + (slot)] = wall_clock64()
#else
#define WF_MSTR(slot)
#endif

template <int P, int BX, int BY>
struct MassLayout {
  static constexpr int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY;
  static constexpr int CW = 64 / n2;                      // cells per wave
  static constexpr int NWV = (CB + CW - 1) / CW;          // waves per workgroup
  static constexpr int WG = 64 * NWV;
  static constexpr int TX = P * BX + 1, TY = P * BY + 1, TP = TX * TY;
  static constexpr int oUx = 0;                           // [2][(P + 1) TP]
  static constexpr int oO = oUx + 2 * (P + 1) * TP;       // [2][P TP] result tile of a layer, cells combined
  static constexpr int oCy = oO + 2 * P * TP;             // [2][CB n2]
  static constexpr int oA = oCy + 2 * CB * n2;            // [CB nd] wave-private scratch of a cell (the passes work in place)
  static constexpr int oDump = oA + CB * nd;               // [WG] where a thread's out-of-tile last position is stored
  static constexpr int ndoubles = ((oDump + 64 * ((CB + CW - 1) / CW) + 1) / 2) * 2;
  // ring of index-table planes behind the doubles: at the time layer l stores the planes of layer l + 2
  // (<= P l + 3 P) the flush of layer l - 1 may still read plane P (l - 1): 4 P + 1 planes are live
  static constexpr int RP = 4 * P + 1, RPT = RP * TP;
  static_assert(CW >= 1, "a cell does not fit a wave");
};

size_t mass_march_lds_bytes(int P, int BX, int BY, int lz)
{
  const int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY, TP = (P * BX + 1) * (P * BY + 1);
  const size_t d = (size_t)2 * (P + 1) * TP + (size_t)2 * P * TP + (size_t)2 * CB * n2 + (size_t)CB * nd + 64 * (size_t)((CB + (64 / n2) - 1) / (64 / n2)) + 2;
  (void)lz;   // the index table streams through a ring of 4 P + 1 planes: the footprint does not depend on the segment length
  return d * sizeof(double) + ((size_t)(4 * P + 1) * TP + 64 * (size_t)((CB + (64 / n2) - 1) / (64 / n2))) * sizeof(int32_t);
}

struct MassArgs {
  int lz, tile_size;
  const int32_t* item_base;
  const int32_t* item_pattern;
  const int32_t* item_layers;
  const int32_t* pat_off;
  const double* detJ;     // [item lz + layer][k][CB n2]
  const double* phi1;     // [n][n] row-major: phi1[q][a]
  const double* x;
  double* y;
};

// Diagnostic build only (tools/variant_lib.sh ... -DWF_MASS_ABL=<mask>; results are wrong): 1 = plain stores instead
// of the y atomics, 2 = no y memory operation, 4 = no x loads, 8 = no det J loads, 16 = no passes, 32 = no table loads
#ifndef WF_MASS_ABL
#define WF_MASS_ABL 0
#endif
#ifndef WF_MASS_FLUSH_SLOTS
#define WF_MASS_FLUSH_SLOTS 2
#endif

// LDS accumulate without return (ds_add_f64)
__device__ __forceinline__ void lds_add(double* p, double v)
{
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// compiler-level ordering of the wave-private LDS exchange (the hardware executes one wave's LDS operations in order)
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int P, int BX, int BY>
__global__ __launch_bounds__((MassLayout<P, BX, BY>::WG), 2) void k_mass_march(MassArgs a)
{
  using L = MassLayout<P, BX, BY>;
  constexpr int n = L::n, n2 = L::n2, nd = L::nd, CB = L::CB, CW = L::CW, WG = L::WG;
  constexpr int TX = L::TX, TP = L::TP;
  constexpr int NPOS = (P * TP + WG - 1) / WG;          // flush / x-prefetch positions per thread
  constexpr int NPOS0 = ((P + 1) * TP + WG - 1) / WG;   // prologue x positions per thread
  constexpr int NCP = (TP + WG - 1) / WG;               // positions of one plane per thread
  extern __shared__ __attribute__((aligned(16))) double ms_smem[];
  double* Ux = ms_smem + L::oUx;
  double* O = ms_smem + L::oO;
  double* Cy = ms_smem + L::oCy;
  int32_t* sIdx = reinterpret_cast<int32_t*>(ms_smem + L::ndoubles);   // ring [RP][TP] of table planes: dof offsets, -1 = none
  constexpr int RPT = L::RPT;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cw = lane / n2, pq = lane % n2, p0 = pq % n, p1 = pq / n;
  const int cl = wave * CW + cw;                       // cell of the layer
  const bool active = cw < CW && cl < CB;
  const int lx = cl % BX, ly = cl / BX;
  double* A = ms_smem + L::oA + (active ? cl : 0) * nd;
  const size_t item = blockIdx.x;
  const int nl = a.item_layers[item];
  const size_t gbase = (size_t)a.item_base[item];

  // the 1-D table in VECTOR registers (the same in every lane): the opaque zero makes the address per-lane, so
  // that the compiler does not keep the n^2 wave-uniform values in SGPRs (it has ~100) and spill them
  int lane_zero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
  double ph[n][n];
#pragma unroll
  for (int q = 0; q < n; ++q)
#pragma unroll
    for (int c = 0; c < n; ++c) ph[q][c] = a.phi1[q * n + c + lane_zero];

  // det J w of this lane's quadrature column (qi, qj) = (p0, p1), all levels; two sets swapping roles
  double dA[n], dB[n];
  auto load_d = [&](double (&d)[n], int l) {
    const double* dp = a.detJ + ((item * (size_t)a.lz + l) * n) * (size_t)(CB * n2) + (active ? cl * n2 + pq : 0);
#pragma unroll
    for (int k = 0; k < n; ++k) d[k] = (WF_MASS_ABL & 8) ? 1.0 + k : __builtin_nontemporal_load(dp + (size_t)k * (CB * n2));
  };

  // ---- prologue ---------------------------------------------------------------------------
  // Index table of the item (plane-major [(P nl + 1)][TP], linear entry E): planes 0 .. 2 P here, then the P planes
  // of layer l + 2 ride the prefetch of layer l (global -> register -> ring at the layer's end).  Entry E lives at
  // ring position E mod RPT, kept as a wave-uniform base + the thread's position with one conditional wrap.
  const int32_t* __restrict__ pat = a.pat_off + (size_t)a.item_pattern[item] * a.tile_size;
  const int Emax = (P * nl + 1) * TP - 1;
  for (int e = t; e < (2 * P + 1) * TP; e += WG) sIdx[e] = pat[e <= Emax ? e : Emax];
  __syncthreads();
  load_d(dA, 0);
#pragma unroll
  for (int m = 0; m < NPOS0; ++m) {
    const int pos = t + WG * m;
    if (pos < (P + 1) * TP) {
      const int32_t off = sIdx[pos];
      Ux[pos] = off >= 0 ? a.x[gbase + off] : 0.0;
    }
  }
  if (active) Cy[CB * n2 + cl * n2 + pq] = 0.0;   // carry into the first layer (buffer of "layer -1")
  for (int e = t; e < 2 * P * TP; e += WG) O[e] = 0.0;
  __syncthreads();

  const int ucell = (P * ly) * TX + P * lx;

  // flush of a layer's result tile: the cells' contributions were summed in LDS by the X^T pass (ds_add_f64), so a
  // position is one LDS read (+ the zero for the tile's next use), one index read and one global atomic -- no
  // position decoding, no neighbour cases (as a gather over up to four cells' private results the flush took 1.8 us
  // of a 5.0 us layer at P6, 0.9 us of it without the atomics: tools/mass_trace.py)
  // ring position of entry E = base + pos (base < RPT wave-uniform, pos < RPT)
  auto ring = [&](int base, int pos) {
    const int e = base + pos;
    return e >= RPT ? e - RPT : e;
  };
  auto ring_advance = [&](int& base, int by) {
    base += by;
    if (base >= RPT) base -= RPT;
  };
  // bases of: the planes being flushed (layer l - 1: entry P (l - 1) TP), the planes of the x prefetch (layer l + 1:
  // entry (P (l + 1) + 1) TP) and the planes arriving (layer l + 2)
  int rb_flush = 0, rb_x = (P + 1) * TP, rb_in = (2 * P + 1) * TP;
  int e_in = (2 * P + 1) * TP;   // linear entry of the arriving planes (global side)
  static_assert((2 * P + 1) * L::TP < L::RPT, "ring too small");
  auto flush = [&](double* Tb, int l, int m0, int m1) {
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      if (m < m0 || m >= m1) continue;
      const int pos = t + WG * m;
      if (pos >= P * TP) continue;
      const int32_t off = sIdx[ring(rb_flush, pos)];
      const double v = Tb[pos];
      Tb[pos] = 0.0;
      if (off < 0) continue;
#if WF_MASS_ABL & 1
      a.y[gbase + off] = v;
#elif WF_MASS_ABL & 2
      asm volatile("" ::"v"(v));
#else
      // (measured: a plain y[g] += v for the positions strictly inside the tile, which belong to this work item
      // alone, is 3.5x SLOWER -- P4 0.117 -> 0.408 ms: the load -> add -> store chain waits in the flush, the
      // atomic is fire-and-forget)
      unsafeAtomicAdd(a.y + gbase + off, v);
#endif
    }
  };
  // the flush of layer l - 1 is spread over the five passes of layer l (slot s = 0..4): a burst of NPOS atomics
  // per thread stalled the issue of the memory instructions behind it (tools/mass_trace.py: +0.5 us in the flush
  // and +0.5 us in the next prefetch issue at P6)
  auto flush_slot = [&](double* Tprev, int l, int slot) {
    constexpr int NS = WF_MASS_FLUSH_SLOTS;   // passes that carry a share of the flush (the first NS)
    if (l > 0 && slot < NS) flush(Tprev, l - 1, slot * NPOS / NS, (slot + 1) * NPOS / NS);
  };

  // pencil contraction out[q] = sum_c phi[q][c] in[c] (forward) or out[c] = sum_q phi[q][c] in[q] (transposed)
  auto fwd = [&](const double (&in)[n], double (&out)[n]) {
#pragma unroll
    for (int q = 0; q < n; ++q) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < n; ++c) s += ph[q][c] * in[c];
      out[q] = s;
    }
  };
  auto bwd = [&](const double (&in)[n], double (&out)[n]) {
#pragma unroll
    for (int c = 0; c < n; ++c) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < n; ++q) s += ph[q][c] * in[q];
      out[c] = s;
    }
  };

  // `has_next` (l + 1 < nl) and `has_next2` (l + 2 < nl) are compile-time constants of the layer body (three copies
  // chosen by uniform branches below): a uniform `if (has_next)` around the det J prefetch makes the compiler merge,
  // at the join, the wait-count state of the path without those loads -- in which the x / table loads are the
  // youngest -- and (c) then waited with vmcnt(0) for the det J stream of the NEXT layer (ISA before this change).
  // The stores of (c) sit in straight-line code: a thread's out-of-tile last position goes to a dump slot.
  auto layer = [&](auto hn_tag, auto hn2_tag, double (&dcur)[n], double (&dnext)[n], int l, int b) {
    constexpr bool has_next = decltype(hn_tag)::value, has_next2 = decltype(hn2_tag)::value;
    const double* Ub = Ux + b * (P + 1) * TP;
    double* Un = Ux + (b ^ 1) * (P + 1) * TP;
    double* Tb = O + b * (P * TP);
    double* Tprev = O + (b ^ 1) * (P * TP);
    const int ln = has_next ? l + 1 : l;
    WF_MSTR(0);
    // (a) next layer's x planes and det J: in flight during the passes (unconditional loads on clamped addresses)
    double xn[NPOS];
    int32_t tn[NPOS];
    const int rbx = has_next ? rb_x : ring(rb_x, RPT - P * TP);   // last layer: its own planes again (never stored)
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      const int pos = t + WG * m;
      const int32_t off = pos < P * TP ? sIdx[ring(rbx, pos)] : -1;
      xn[m] = (WF_MASS_ABL & 4) ? 1.0 + m : a.x[gbase + (off >= 0 ? off : 0)];
      const int e = e_in + pos;
      tn[m] = (WF_MASS_ABL & 32) ? e : pat[e <= Emax ? e : Emax];
    }
    if (has_next) load_d(dnext, ln);
    WF_MSTR(1);

    // (b) the five passes of the element kernel, wave-private, with the previous layer's flush between them
    double in[n], out[n];
    flush_slot(Tprev, l, 0);
    const bool run_passes = active && !(WF_MASS_ABL & 16);
    if (run_passes) {
      // X: lane (j, k) = (p0, p1)
#pragma unroll
      for (int c = 0; c < n; ++c) in[c] = Ub[ucell + p1 * TP + p0 * TX + c];
      fwd(in, out);
#pragma unroll
      for (int q = 0; q < n; ++q) A[(p1 * n + p0) * n + q] = out[q];
      wave_sync();
    }
    WF_MSTR(2);
    flush_slot(Tprev, l, 1);
    if (run_passes) {
      // Y: lane (qi, k) = (p0, p1)
#pragma unroll
      for (int c = 0; c < n; ++c) in[c] = A[(p1 * n + c) * n + p0];
      fwd(in, out);
#pragma unroll
      for (int q = 0; q < n; ++q) A[(p1 * n + q) * n + p0] = out[q];
      wave_sync();
    }
    WF_MSTR(3);
    flush_slot(Tprev, l, 2);
    if (run_passes) {
      // Z: lane (qi, qj) = (p0, p1): forward, times det J w, transposed
#pragma unroll
      for (int c = 0; c < n; ++c) in[c] = A[(c * n + p1) * n + p0];
      fwd(in, out);
#pragma unroll
      for (int q = 0; q < n; ++q) out[q] *= dcur[q];
      bwd(out, in);
#pragma unroll
      for (int c = 0; c < n; ++c) A[(c * n + p1) * n + p0] = in[c];
      wave_sync();
    }
    WF_MSTR(4);
    flush_slot(Tprev, l, 3);
    if (run_passes) {
      // Y^T: lane (qi, k) = (p0, p1)
#pragma unroll
      for (int q = 0; q < n; ++q) in[q] = A[(p1 * n + q) * n + p0];
      bwd(in, out);
#pragma unroll
      for (int c = 0; c < n; ++c) A[(p1 * n + c) * n + p0] = out[c];
      wave_sync();
    }
    WF_MSTR(5);
    flush_slot(Tprev, l, 4);
    if (run_passes) {
      // X^T: lane (j, k) = (p0, p1); planes 0..P-1 -> O, plane P -> carry, plane 0 picks up the previous carry
#pragma unroll
      for (int q = 0; q < n; ++q) in[q] = A[(p1 * n + p0) * n + q];
      bwd(in, out);
      if (p1 == 0) {
#pragma unroll
        for (int c = 0; c < n; ++c) out[c] += Cy[(b ^ 1) * (CB * n2) + cl * n2 + p0 * n + c];
      }
      if (p1 < P) {
        double* dst = Tb + ucell + p1 * TP + p0 * TX;
#pragma unroll
        for (int c = 0; c < n; ++c) lds_add(dst + c, out[c]);
      } else {
        double* dst = Cy + b * (CB * n2) + cl * n2 + p0 * n;
#pragma unroll
        for (int c = 0; c < n; ++c) dst[c] = out[c];
      }
    }
    WF_MSTR(6);
    // (c) x planes of the next layer -> the other buffer (the consumer of xn)
    if (has_next) {
      double* dumpd = ms_smem + L::oDump + t;
      int32_t* dumpi = sIdx + RPT + t;
#pragma unroll
      for (int m = 0; m < NCP; ++m) {
        const int pos = t + WG * m;
        const bool in = WG * (m + 1) <= TP || pos < TP;
        const double v = Ub[P * TP + (in ? pos : 0)];
        *(in ? Un + pos : dumpd) = v;
      }
#pragma unroll
      for (int m = 0; m < NPOS; ++m) {
        const int pos = t + WG * m;
        const bool in = WG * (m + 1) <= P * TP || pos < P * TP;
        const int32_t off = sIdx[ring(rb_x, in ? pos : 0)];
        *(in ? Un + TP + pos : dumpd) = off >= 0 ? xn[m] : 0.0;
      }
      if (has_next2) {
#pragma unroll
        for (int m = 0; m < NPOS; ++m) {
          const int pos = t + WG * m;
          const bool in = WG * (m + 1) <= P * TP || pos < P * TP;
          *(in ? sIdx + ring(rb_in, pos) : dumpi) = tn[m];
        }
      }
    }
    if (l > 0) ring_advance(rb_flush, P * TP);
    ring_advance(rb_x, P * TP);
    ring_advance(rb_in, P * TP);
    e_in += P * TP;
    WF_MSTR(7);
    __syncthreads();   // the one workgroup barrier of the layer
    WF_MSTR(8);
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  auto layer_any = [&](double (&dcur)[n], double (&dnext)[n], int l, int b) {
    if (l + 2 < nl)
      layer(Yes{}, Yes{}, dcur, dnext, l, b);
    else if (l + 1 < nl)
      layer(Yes{}, No{}, dcur, dnext, l, b);
    else
      layer(No{}, No{}, dcur, dnext, l, b);
  };
  for (int l = 0; l < nl; l += 2) {
    layer_any(dA, dB, l, 0);
    if (l + 1 < nl) layer_any(dB, dA, l + 1, 1);
  }

  // ---- epilogue: the last layer's tile and the last (carried) plane ---------------------------------------------------
  flush(O + ((nl - 1) & 1) * (P * TP), nl - 1, 0, NPOS);
  ring_advance(rb_flush, P * TP);                 // -> plane P nl
  {
    const double* Cb = Cy + ((nl - 1) & 1) * (CB * n2);
#pragma unroll
    for (int m = 0; m < NCP; ++m) {
      const int pos = t + WG * m;
      if (pos >= TP) continue;
      const int32_t off = sIdx[ring(rb_flush, pos)];
      if (off < 0) continue;
      const int J = pos / TX, I = pos % TX;
      const int ca = I / P, ia = I % P, cb = J / P, jb = J % P;
      double v = 0.0;
      if (cb < BY) {
        if (ca < BX) v += Cb[(cb * BX + ca) * n2 + jb * n + ia];
        if (ia == 0 && ca > 0) v += Cb[(cb * BX + ca - 1) * n2 + jb * n + P];
      }
      if (jb == 0 && cb > 0) {
        if (ca < BX) v += Cb[((cb - 1) * BX + ca) * n2 + P * n + ia];
        if (ia == 0 && ca > 0) v += Cb[((cb - 1) * BX + ca - 1) * n2 + P * n + P];
      }
      unsafeAtomicAdd(a.y + gbase + off, v);
    }
  }
}

template <int P, int BX, int BY>
static int launch_mass_t(const MassArgs& a, int nitems, size_t lds, hipStream_t s)
{
  if (nitems == 0) return WF_OK;
  WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mass_march<P, BX, BY>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_mass_march<P, BX, BY>), dim3((unsigned)nitems), dim3(MassLayout<P, BX, BY>::WG), lds, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("mass_march launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

// column cross-sections (the first of a degree is its default): whole cells per wave (floor(64 / n^2))
// Smaller cross-sections (more, smaller workgroups per CU) measured slower at ~10 M dofs: P4 4x2 0.116 ms, 2x2
// 0.155, 2x1 0.166; P6 2x2 0.126, 2x1 0.138, 1x1 0.182; P2 7x4 0.126, 7x2 0.140 (tools/bench_mass_lz.py).
#define WF_MASS_SHAPES(X) \
  X(1, 8, 8) X(2, 7, 4) X(3, 4, 4) X(4, 4, 2) X(4, 2, 2) X(5, 2, 2) X(6, 2, 2) X(6, 2, 1) X(7, 2, 1)

// keeps (*bx, *by) if that cross-section is compiled for the degree, else the degree's default
void mass_march_shape(int P, int* bx, int* by)
{
#define X(PP, BXX, BYY) \
  if (P == PP && *bx == BXX && *by == BYY) return;
  WF_MASS_SHAPES(X)
#undef X
#define X(PP, BXX, BYY) \
  if (P == PP) {        \
    *bx = BXX;          \
    *by = BYY;          \
    return;             \
  }
  WF_MASS_SHAPES(X)
#undef X
}

int launch_mass_march(int P, const MarchPlanDev& pd, const double* d_detJblk, const double* d_phi1, const double* d_x,
                      double* d_y, hipStream_t s)
{
  MassArgs a{};
  a.lz = pd.lz;
  a.tile_size = pd.tile_size;
  a.item_base = pd.d_item_base;
  a.item_pattern = pd.d_item_pattern;
  a.item_layers = pd.d_item_layers;
  a.pat_off = pd.d_pat_off;
  a.detJ = d_detJblk;
  a.phi1 = d_phi1;
  a.x = d_x;
  a.y = d_y;
  const size_t lds = mass_march_lds_bytes(P, pd.bx, pd.by, pd.lz);
#define X(PP, BXX, BYY) \
  if (P == PP && pd.bx == BXX && pd.by == BYY) return launch_mass_t<PP, BXX, BYY>(a, pd.nitems, lds, s);
  WF_MASS_SHAPES(X)
#undef X
  set_error("mass_march: cross-section not compiled");
  return WF_ERR_UNSUPPORTED;
}

}  // namespace wf

#ifdef WF_MASS_TRACE
extern "C" int wf_debug_mass_trace(unsigned long long* host, size_t n)
{
  void* sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(wf::g_mass_trace)) != hipSuccess) return -1;
  if (n > sizeof(wf::g_mass_trace) / 8) n = sizeof(wf::g_mass_trace) / 8;
  return hipMemcpy(host, sym, n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
