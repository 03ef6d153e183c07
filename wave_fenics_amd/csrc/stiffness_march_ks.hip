// k_march_ks<P, BX, BY, IDX>: the marching stiffness kernel in "k-split" form -- the production
// kernel for y += -c0^2 K x at degrees the one-thread-per-column kernels (stiffness_march.hip,
// stiffness_march_idx.hip) cannot hold in registers, P = 5, 6, 7 (common/operators.hpp:113-133,
// 183-200; degree range of the reference: common/cuda/spectral_mass.hpp:43-48).
//
// Same marching structure: a workgroup owns a column of BX x BY cells and walks through its
// layers; the next layer's geometry (48 B per point, the dominant HBM stream) and x planes are in
// flight while the current layer is computed; finished dof planes go to y with one fp64 atomic per
// tile dof.  What differs:
//  * TWO threads per cell column (i, j): the workgroup has two halves of TH threads (whole waves),
//    half h handles the quadrature levels k in [h KH, min(n, (h + 1) KH)), KH = ceil(n / 2).  A
//    thread then holds 2 x 6 KH doubles of geometry (P6: 96 VGPRs for both register sets instead
//    of 168), the kernel fits 256 VGPRs and runs 8 waves per CU.  The half is wave-uniform, so each
//    half runs its own specialisation with compile-time k (the z-direction entries of D stay scalar
//    operands);
//  * all three flux components go through LDS between the two phases of the element kernel (the
//    z contraction needs the levels of the other half), and the z-shared plane is handed from the
//    upper half of one layer to the lower half of the next through LDS;
//  * the x planes, the per-cell results and that carried plane are double-buffered in LDS, which
//    leaves TWO workgroup barriers per layer (after phase 1, after phase 2) instead of three: the
//    flush of layer l runs beside phase 1 of layer l + 1.
// IDX = false: box mesh, implicit lattice addresses (k_stiffness_march's interface);
// IDX = true : any dofmap, addresses from the work item's index table staged in LDS
//              (k_march_idx's interface, plan of generic_plan.cpp).
// HBM-bound; algorithmic bytes ncells (48 nq + 4 nd) + 16 ndofs.
#include <type_traits>

#include "stiffness_core.h"

namespace wf {

template <int P>
struct KSplit {
  static constexpr int n = P + 1, KH = (n + 1) / 2;
};

// phase 1 of the element kernel for the levels of half H: reference gradient at the points
// (i, j, k), times G -> Fr, Fs, Ft (LDS, compact cell layout [k][j][i])
// The z-direction entries of D are wave-uniform (compile-time k).  Up to P4 they are kernel-argument
// constants that live in SGPRs (DMat by value).  From P5 on the 2 n^2 SGPRs they need exceed what a wave
// has: the compiler spilled them to VGPR lanes and every use cost two v_readlane (P6: ~1000 v_readlane per
// kernel).  There they are read from the LDS copy of D instead (every lane the same address: a broadcast
// read; sD holds D row-major followed by its transpose).  Scalar loads from global memory at the start of
// each phase were measured and rejected (P4 0.2015 -> 0.266 ms, P6 0.232 -> 0.314 ms): a pending s_load
// turns every LDS wait of the phase into lgkmcnt(0).
template <int P>
constexpr bool ks_dz_from_lds() { return P >= 5; }

template <int P, int H>
__device__ __forceinline__ void ks_phase1(const double* __restrict__ U, int sk, int sj, double* __restrict__ Fr,
                                          double* __restrict__ Fs, double* __restrict__ Ft,
                                          const double* __restrict__ sD, const DMat& dm,
                                          const double2 (&g)[KSplit<P>::KH][3], double coeff, int i, int j)
{
  constexpr int n = P + 1, n2 = n * n, KH = KSplit<P>::KH, K0 = H * KH, K1 = H == 0 ? KH : n;
  double ru[n], di[n], dj[n];
#pragma unroll
  for (int a = 0; a < n; ++a) {
    ru[a] = U[a * sk + j * sj + i];
    di[a] = sD[i * n + a];
    dj[a] = sD[j * n + a];
  }
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
    for (int a = 0; a < n; ++a) {
      ur += di[a] * U[k * sk + j * sj + a];
      us += dj[a] * U[k * sk + a * sj + i];
      ut += (ks_dz_from_lds<P>() ? sD[k * n + a] : dm.v[k * n + a]) * ru[a];
    }
    const double2 *gk = g[k - K0];
    const double g00 = gk[0].x, g01 = gk[0].y, g02 = gk[1].x, g11 = gk[1].y, g12 = gk[2].x, g22 = gk[2].y;
    // operators.hpp:126-128: fw = coeff * (G row . w)
    Fr[k * n2 + j * n + i] = coeff * (g00 * ur + g01 * us + g02 * ut);
    Fs[k * n2 + j * n + i] = coeff * (g01 * ur + g11 * us + g12 * ut);
    Ft[k * n2 + j * n + i] = coeff * (g02 * ur + g12 * us + g22 * ut);
  }
}

// phase 2: out[k - K0] = (K_cell u)[i, j, k] for the levels of half H
template <int P, int H>
__device__ __forceinline__ void ks_phase2(const double* __restrict__ Fr, const double* __restrict__ Fs,
                                          const double* __restrict__ Ft, const double* __restrict__ sD,
                                          const DMat& dm, int i, int j, double (&out)[KSplit<P>::KH])
{
  constexpr int n = P + 1, n2 = n * n, KH = KSplit<P>::KH, K0 = H * KH, K1 = H == 0 ? KH : n;
  double fz[n], dti[n], dtj[n];
#pragma unroll
  for (int a = 0; a < n; ++a) {
    fz[a] = Ft[a * n2 + j * n + i];
    dti[a] = sD[a * n + i];
    dtj[a] = sD[a * n + j];
  }
#pragma unroll
  for (int k = K0; k < K1; ++k) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < n; ++a) {
      s += dti[a] * Fr[k * n2 + j * n + a];
      s += dtj[a] * Fs[k * n2 + a * n + i];
      s += (ks_dz_from_lds<P>() ? sD[n * n + k * n + a] : dm.v[a * n + k]) * fz[a];   // sD + n^2: transpose of D
    }
    out[k - K0] = s;
  }
}

// shared-memory carve-up (doubles unless noted); dynamic LDS so that the index table can follow
template <int P, int BX, int BY>
struct KSLayout {
  static constexpr int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY, NTc = CB * n2;
  static constexpr int TH = ((NTc + 63) / 64) * 64, WG = 2 * TH;
  static constexpr int TX = P * BX + 1, TY = P * BY + 1, TP = TX * TY;
  static constexpr int oUx = 0;                          // [2][(P + 1) TP]
  static constexpr int oO = oUx + 2 * (P + 1) * TP;      // [2][P TP] result tile of a layer, the cells' contributions summed (ds_add_f64)
  static constexpr int oCy = oO + 2 * P * TP;            // [2][NTc]
  static constexpr int oFr = oCy + 2 * NTc;              // [CB nd] x 3
  static constexpr int oD = oFr + 3 * CB * nd;           // [2][n n]: D and its transpose
  static constexpr int ndoubles = ((oD + 2 * n * n + 1) / 2) * 2;
  static_assert(NTc <= 256, "column does not fit two halves of 256 threads");
};

size_t march_ks_lds_bytes(int P, int BX, int BY, int lz, bool idx)
{
  const int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY, TP = (P * BX + 1) * (P * BY + 1);
  size_t d = (size_t)2 * (P + 1) * TP + (size_t)2 * P * TP + (size_t)2 * CB * n2 + (size_t)3 * CB * nd + 2 * n * n + 2;
  return d * sizeof(double) + (idx ? (size_t)(P * lz + 1) * TP * sizeof(int32_t) : 0);
}

struct KSArgs {
  // box
  int nx, ny, nz, lz, lz0;
  // idx
  int tile_size;
  const int32_t* item_base;
  const int32_t* item_pattern;
  const int32_t* item_layers;
  const int32_t* pat_off;
  // both
  const int32_t* items;   // optional work-item list (interior / interface split)
  const double2* G6blk;
  const double* dD;       // device: D[q][a] (n x n, row-major) followed by its transpose
  double coeff;
  const double* x;
  double* y;
};

#ifndef WF_KS_FLUSH_SPLIT
#define WF_KS_FLUSH_SPLIT 0
#endif

template <int P, int BX, int BY, bool IDX, int H>
__device__ __forceinline__ void ks_march(const KSArgs& a, const DMat& dm, double* smem)
{
  using L = KSLayout<P, BX, BY>;
  constexpr int n = L::n, n2 = L::n2, nd = L::nd, CB = L::CB, NTc = L::NTc, TH = L::TH, WG = L::WG;
  constexpr int TX = L::TX, TY = L::TY, TP = L::TP, KH = KSplit<P>::KH, K0 = H * KH, K1 = H == 0 ? KH : n, NK = K1 - K0;
  constexpr int NPOS = (P * TP + WG - 1) / WG;          // flush / x-prefetch positions per thread
  constexpr int NPOS0 = ((P + 1) * TP + WG - 1) / WG;   // prologue x positions per thread
  constexpr int NCP = (TP + WG - 1) / WG;               // positions of one plane per thread

  double* Ux = smem + L::oUx;
  double* O = smem + L::oO;
  double* Cy = smem + L::oCy;
  double* Fr = smem + L::oFr;
  double* Fs = Fr + CB * nd;
  double* Ft = Fs + CB * nd;
  double* sD = smem + L::oD;
  int32_t* sIdx = reinterpret_cast<int32_t*>(smem + L::ndoubles);   // IDX: [(P nl + 1)][TP] dof offsets, -1 = none

  const int t = threadIdx.x;
  const int tc = t - H * TH;   // column thread inside the half
  const bool active = tc < NTc;
  const int cl = tc / n2, ji = tc % n2, j = ji / n, i = ji % n;
  const int lx = cl % BX, ly = cl / BX;
  const size_t item = a.items ? (size_t)a.items[blockIdx.x] : (size_t)blockIdx.x;

  // ---- where the column sits -----------------------------------------------------------
  int nl;                 // layers of this work item
  size_t gblock0;         // geometry block of its first layer, block stride gstride
  size_t gstride;
  size_t gbase = 0;       // IDX: smallest dof of the item; box: lattice offset of the tile origin in the first plane
  [[maybe_unused]] int32_t poff[NPOS];   // box: lattice offset of (I0 + I, J0 + J, pl) relative to the layer's first plane, -1 = outside
  [[maybe_unused]] size_t plane = 0;
  [[maybe_unused]] int EX = TX, EY = TY;
  [[maybe_unused]] int z0 = 0;
  if constexpr (IDX) {
    nl = a.item_layers[item];
    gbase = (size_t)a.item_base[item];
    gblock0 = item * (size_t)a.lz;
    gstride = 1;
  } else {
    const int nbx = (a.nx + BX - 1) / BX, nby = (a.ny + BY - 1) / BY, ncols = nbx * nby;
    const int col = (int)(item % ncols), seg = (int)(item / ncols);
    const int Bx = col % nbx, By = col / nbx;
    z0 = seg == 0 ? 0 : a.lz0 + (seg - 1) * a.lz;
    const int z1 = min(a.nz, seg == 0 ? a.lz0 : z0 + a.lz);
    nl = z1 - z0;
    const int NX = P * a.nx + 1, NY = P * a.ny + 1;
    plane = (size_t)NX * NY;
    const int I0 = P * Bx * BX, J0 = P * By * BY;
    EX = min(TX, NX - I0);
    EY = min(TY, NY - J0);
    gbase = (size_t)I0 + (size_t)NX * J0;
    gblock0 = (size_t)Bx + (size_t)nbx * (By + (size_t)nby * z0);
    gstride = (size_t)nbx * nby;
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      const int pos = t + WG * m;
      const int pl = pos / TP, r = pos % TP, J = r / TX, I = r % TX;
      poff[m] = -1;
      if (pos < P * TP && I < EX && J < EY) poff[m] = (int32_t)((size_t)I + (size_t)NX * J + plane * pl);
    }
  }

  // geometry registers of this thread: levels K0 .. K1-1 of column (i, j); two sets that swap roles
  // from layer to layer (a copy gcur = gnext would be moved to the loop's back edge by the compiler)
  double2 gA[KH][3], gB[KH][3];
  auto load_g = [&](double2 (&g)[KH][3], int l, int k0, int k1) {
    const double2* gp = a.G6blk + ((gblock0 + gstride * (size_t)l) * n * 3) * (size_t)NTc + (active ? tc : NTc - 1);
#pragma unroll
    for (int k = K0; k < K1; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (k - K0 >= k0 && k - K0 < k1) g[k - K0][p] = load_stream(gp + (size_t)(k * 3 + p) * NTc);
  };
  // the next layer's geometry is requested in two instalments (before phase 1, after barrier A)
  constexpr int G1 = (NK + 1) / 2;

  // ---- prologue ---------------------------------------------------------------------------
  for (int e = t; e < 2 * n * n; e += WG) sD[e] = a.dD[e];
  if constexpr (IDX) {
    const int32_t* __restrict__ pat = a.pat_off + (size_t)a.item_pattern[item] * a.tile_size;
    for (int e = t; e < (P * nl + 1) * TP; e += WG) sIdx[e] = pat[e];
    __syncthreads();
  }
  load_g(gA, 0, 0, NK);
#pragma unroll
  for (int m = 0; m < NPOS0; ++m) {
    const int pos = t + WG * m;
    if (pos < (P + 1) * TP) {
      double v = 0.0;
      if constexpr (IDX) {
        const int32_t off = sIdx[pos];
        if (off >= 0) v = a.x[gbase + off];
      } else {
        const int pl = pos / TP, r = pos % TP, J = r / TX, I = r % TX;
        if (I < EX && J < EY) v = a.x[plane * (size_t)(P * z0 + pl) + gbase + (size_t)I + (size_t)(P * a.nx + 1) * J];
      }
      Ux[pos] = v;
    }
  }
  if (H == 1 && active) Cy[NTc + tc] = 0.0;   // carry into the first layer (buffer of "layer -1")
  for (int e = t; e < 2 * P * TP; e += WG) O[e] = 0.0;
  __syncthreads();

  const int ucell = (P * ly) * TX + P * lx;

  // flush of the planes a layer has finished: the cells' contributions were summed in the LDS tile by phase 2
  // (ds_add_f64), so a position is one LDS read (+ the zero for the tile's next use) and one global atomic (as a
  // gather over the up to four cells' private results the flush decoded its position and took four guarded reads)
  auto flush = [&](double* Tb, int l, int m0, int m1) {
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      if (m < m0 || m >= m1) continue;
      const int pos = t + WG * m;
      if (pos >= P * TP) continue;
      const double v = Tb[pos];
      Tb[pos] = 0.0;
      size_t dst;
      if constexpr (IDX) {
        const int32_t off = sIdx[(P * l) * TP + pos];
        if (off < 0) continue;
        dst = gbase + off;
      } else {
        if (poff[m] < 0) continue;
        dst = plane * (size_t)(P * (z0 + l)) + gbase + poff[m];
      }
      unsafeAtomicAdd(a.y + dst, v);
    }
  };

  // x planes 1..P of layer `lt` -> registers (positions outside the tile / dead entries read a valid
  // dummy address: a guard around a load is a branch, and at its join the compiler's wait-count
  // bookkeeping turns conservative; what is live is decided where the registers are consumed)
  double xn[NPOS];
  auto load_x = [&](int lt) {
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      const int pos = t + WG * m;
      if constexpr (IDX) {
        const int32_t off = pos < P * TP ? sIdx[(P * lt + 1) * TP + pos] : -1;
        xn[m] = a.x[gbase + (off >= 0 ? off : 0)];
      } else {
        xn[m] = a.x[plane * (size_t)(P * (z0 + lt) + 1) + gbase + (poff[m] >= 0 ? poff[m] : 0)];
      }
    }
  };
  // Where the request for the next layer goes (measured per cross-section, cfg-size meshes):
  //  EARLY: the request for layer l + 2 (x planes, first half of the geometry) right after barrier B of
  //         layer l, in front of the flush's atomics, into the registers layer l has just finished with --
  //         loads, stores and atomics retire in issue order (vmcnt), so a load issued BEHIND an atomic
  //         cannot be waited for without the atomic's round trip, and hipcc makes even the address
  //         arithmetic of such a load wait for it.  Faster for the 512-thread workgroups (P4 5x2:
  //         0.241 -> 0.221 ms box, 0.264 -> 0.236 ms indexed);
  //  LATE : the request for layer l + 1 at the top of layer l.  Shorter register live ranges: faster for the
  //         small workgroups that run three per CU (P4 5x1: 0.201 vs 0.233 ms, where EARLY spills).
  constexpr bool EARLY = WG >= 512;
  if constexpr (EARLY) {   // prefetch for layer 1 (the prologue has loaded layer 0)
    const int l1 = nl > 1 ? 1 : 0;
    load_x(l1);
    if (nl > 1) load_g(gB, 1, 0, G1);
  }

  // One layer.  b = l & 1 selects the LDS buffers; `gcur` holds the layer's geometry, `gnext`
  // receives the next layer's (second instalment after barrier A).
  // `has_next` / `has_next2` are compile-time properties of the layer body (three copies, chosen by uniform
  // branches in the loop below): with `if (has_next)` around each prefetch instalment the compiler merges, at every
  // join, the wait-count state of the path that issued the geometry loads with the one that did not -- where the x
  // loads are the youngest pending loads -- and the consumers of the x registers in (c) then wait for the geometry
  // issued behind them too (stiffness_march.hip: vmcnt(2)/(1)/(0) instead of (11)/(10)/(9), P4 0.2017 -> 0.1954 ms).
  auto layer = [&](auto hn_tag, auto hn2_tag, double2 (&gcur)[KH][3], double2 (&gnext)[KH][3], int l, int b) {
    const bool has_next = hn_tag, has_next2 = hn2_tag;   // compile-time constants in the specialised copies
    const double* Ub = Ux + b * (P + 1) * TP;
    double* Un = Ux + (b ^ 1) * (P + 1) * TP;
    double* Tb = O + b * (P * TP);
    const int ln = has_next ? l + 1 : l;
    if constexpr (!EARLY) {
      load_x(ln);
      if (has_next) load_g(gnext, ln, 0, G1);
    }
    // Diagnostic switch, measured and not adopted: the flush of layer l - 1 (the other tile buffer) issued inside
    // layer l -- FS = 1 all of it in front of phase 1, FS = 2 half there and half behind barrier A -- instead of at
    // the end of its own layer (FS = 0).  Within +-1 % for the box operators, 3-6 % slower for the indexed P5 ones
    // (tools/variant_lib.sh + bench_shapes.py); it is what made the dense-mass kernel 20 % faster (mass_march.hip).
    constexpr int FS = WF_KS_FLUSH_SPLIT;
    constexpr int NPA = FS == 2 ? (NPOS + 1) / 2 : NPOS;
    if (FS > 0 && l > 0) flush(O + (b ^ 1) * (P * TP), l - 1, 0, NPA);

    // (a) phase 1
    if (active) ks_phase1<P, H>(Ub + ucell, TP, TX, Fr + cl * nd, Fs + cl * nd, Ft + cl * nd, sD, dm, gcur, a.coeff, i, j);
    __syncthreads();   // barrier A
    if (has_next) load_g(gnext, ln, G1, NK);
    if (FS == 2 && l > 0) flush(O + (b ^ 1) * (P * TP), l - 1, NPA, NPOS);

    // (b) phase 2; results of planes 0..P-1 -> O, plane P -> carry; the z-shared plane picks up the
    // carry the upper half left in the previous layer
    if (active) {
      double out[KH];
      ks_phase2<P, H>(Fr + cl * nd, Fs + cl * nd, Ft + cl * nd, sD, dm, i, j, out);
      if (H == 0) out[0] += Cy[(b ^ 1) * NTc + tc];
#pragma unroll
      for (int k = K0; k < K1; ++k) {
        if (k < P)
          __hip_atomic_fetch_add(Tb + k * TP + ucell + j * TX + i, out[k - K0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
          Cy[b * NTc + tc] = out[k - K0];
      }
    }
    // (c) x planes of the next layer -> the other buffer: plane P of this layer becomes plane 0,
    // planes 1..P come from the prefetch (this is the consumer of xn)
    if (has_next) {
#pragma unroll
      for (int m = 0; m < NCP; ++m) {
        const int pos = t + WG * m;
        if (pos < TP) Un[pos] = Ub[P * TP + pos];
      }
#pragma unroll
      for (int m = 0; m < NPOS; ++m) {
        const int pos = t + WG * m;
        if (pos < P * TP) {
          bool live;
          if constexpr (IDX)
            live = sIdx[(P * ln + 1) * TP + pos] >= 0;
          else
            live = poff[m] >= 0;
          Un[TP + pos] = live ? xn[m] : 0.0;
        }
      }
    }
    __syncthreads();   // barrier B
    // (d) EARLY: request layer l + 2, into the registers this layer is done with
    if constexpr (EARLY) {
      const int l2 = has_next2 ? l + 2 : ln;
      load_x(l2);
      if (has_next2) load_g(gcur, l2, 0, G1);
    }
    // (e) flush: runs beside the next layer's phase 1 (no barrier in between)
    if (FS == 0) flush(Tb, l, 0, NPOS);
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  // Measured per cross-section (tools/bench_shapes.py, two libraries side by side): the copies help the workgroups
  // that request LATE (P5 3x1 0.205 -> 0.195 ms, P6 2x1 0.216 -> 0.194, P6 2x1 indexed 0.230 -> 0.205) and hurt the
  // 512-thread ones that request EARLY (P4 5x2 0.205 -> 0.24, P5 7x1 0.219 -> 0.233, P6 5x1 0.213 -> 0.238, P7 2x2
  // 0.171 -> 0.184), which keep one body with run-time flags.
  auto layer_any = [&](double2 (&gcur)[KH][3], double2 (&gnext)[KH][3], int l, int b) {
    if constexpr (EARLY || (P == 7 && BX * BY > 1))   // (P7 2x1: 0.165 ms with one body, 0.167 with the copies)
      layer(l + 1 < nl, l + 2 < nl, gcur, gnext, l, b);   // (plain bools: a wrapper struct with operator bool compiled to slower code)
    else if (l + 2 < nl)
      layer(Yes{}, Yes{}, gcur, gnext, l, b);
    else if (l + 1 < nl)
      layer(Yes{}, No{}, gcur, gnext, l, b);
    else
      layer(No{}, No{}, gcur, gnext, l, b);
  };
  for (int l = 0; l < nl; l += 2) {
    layer_any(gA, gB, l, 0);
    if (l + 1 < nl) layer_any(gB, gA, l + 1, 1);
  }

  // ---- epilogue: the last layer's tile (unless flushed already) and the last (carried) plane --------
  if (WF_KS_FLUSH_SPLIT > 0) flush(O + ((nl - 1) & 1) * (P * TP), nl - 1, 0, NPOS);
  {
    const double* Cb = Cy + ((nl - 1) & 1) * NTc;   // written before barrier B of the last layer
#pragma unroll
    for (int m = 0; m < NCP; ++m) {
      const int pos = t + WG * m;
      if (pos >= TP) continue;
      const int J = pos / TX, I = pos % TX;
      size_t dst;
      if constexpr (IDX) {
        const int32_t off = sIdx[(P * nl) * TP + pos];
        if (off < 0) continue;
        dst = gbase + off;
      } else {
        if (I >= EX || J >= EY) continue;
        dst = plane * (size_t)(P * (z0 + nl)) + gbase + (size_t)I + (size_t)(P * a.nx + 1) * J;
      }
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
      unsafeAtomicAdd(a.y + dst, v);
    }
  }
}

// waves per SIMD the register allocation aims at: 512-thread workgroups (two waves per SIMD each) run one
// per CU; 256-thread workgroups (small columns) three per CU
template <int P, int BX, int BY>
constexpr int ks_min_waves() { return KSLayout<P, BX, BY>::WG >= 512 ? 2 : (P <= 4 ? 3 : 2); }

template <int P, int BX, int BY, bool IDX>
__global__ __launch_bounds__((KSLayout<P, BX, BY>::WG), (ks_min_waves<P, BX, BY>())) void k_march_ks(KSArgs a, DMat dm)
{
  extern __shared__ __attribute__((aligned(16))) double ks_smem[];
  // the half is wave-uniform (TH is a multiple of 64): a scalar branch picks the specialisation
  const int h = __builtin_amdgcn_readfirstlane((int)threadIdx.x / KSLayout<P, BX, BY>::TH);
  if (h == 0)
    ks_march<P, BX, BY, IDX, 0>(a, dm, ks_smem);
  else
    ks_march<P, BX, BY, IDX, 1>(a, dm, ks_smem);
}

template <int P, int BX, int BY, bool IDX>
static int launch_ks_t(const KSArgs& a, const DMat& dm, int nwg, size_t lds, hipStream_t s)
{
  if (nwg == 0) return WF_OK;
  // static + dynamic LDS above the 64 KB default needs the attribute; it is per device, so set it on every launch
  // (a cheap host call) rather than cache it in a process-wide static
  WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_march_ks<P, BX, BY, IDX>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_march_ks<P, BX, BY, IDX>), dim3((unsigned)nwg), dim3(KSLayout<P, BX, BY>::WG), lds, s, a, dm);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("march_ks launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

// Compiled column cross-sections (BX * BY * n^2 <= 256 threads per half); the first entry of a degree is
// the default.  Small columns give 256-thread workgroups, of which two or three share a CU.
// (P <= 4 runs the one-thread-per-column kernels: measured equal at P3 / P4 -- cfg2 0.2216 vs 0.2242 ms inside
// bench.py, 0.2015 vs 0.2003 ms with x and y resident in the Infinity Cache -- and 5 % faster at P2, where
// the two halves of a k-split column get 2 and 1 levels.  The P4 5x1 / 5x2 cross-sections stay compiled as a
// tuning option for such comparisons.)
#define WF_KS_SHAPES(X)                                                                   \
  X(4, 5, 1) X(4, 5, 2)                                                                   \
  X(5, 3, 1) X(5, 7, 1) X(5, 2, 1)                                                        \
  X(6, 2, 1) X(6, 1, 1) X(6, 5, 1)                                                        \
  X(7, 2, 1) X(7, 2, 2) X(7, 1, 1)

bool march_ks_shape(int P, int* bx, int* by)
{
  // keep *bx, *by when they name a compiled cross-section of this degree, else the default
  int dbx = 0, dby = 0;
  bool found = false;
#define X(PP, BXX, BYY)                                  \
  if (P == PP) {                                         \
    if (!dbx) { dbx = BXX; dby = BYY; }                  \
    if (*bx == BXX && *by == BYY) found = true;          \
  }
  WF_KS_SHAPES(X)
#undef X
  if (!dbx) return false;
  if (!found) {
    *bx = dbx;
    *by = dby;
  }
  return true;
}

// workgroups of the kernel that fit a CU (LDS and register file): the round size of the z segmentation
int march_ks_resident(int P, int bx, int by)
{
  const int n = P + 1, NTc = bx * by * n * n, TH = ((NTc + 63) / 64) * 64, WG = 2 * TH;
  const int per_cu_regs = WG >= 512 ? (P <= 3 ? 2 : 1) : (P <= 4 ? 768 / WG : 512 / WG);
  const size_t lds = march_ks_lds_bytes(P, bx, by, 0, false);
  const int per_cu_lds = (int)std::max<size_t>(1, (size_t)160 * 1024 / lds);
  return 256 * std::max(1, std::min(per_cu_regs, per_cu_lds));
}

int launch_stiffness_march_ks_box(int P, int bx, int by, int nx, int ny, int nz, int lz, int lz0, const double* d_G6blk,
                                  const double* d_D, const DMat& dm, double coeff, const double* d_x, double* d_y,
                                  const int32_t* d_items, int nitems, hipStream_t s)
{
  if ((size_t)nx * ny * nz == 0) return WF_OK;
  KSArgs a{};
  a.nx = nx; a.ny = ny; a.nz = nz; a.lz = lz; a.lz0 = lz0;
  a.items = d_items;
  a.G6blk = reinterpret_cast<const double2*>(d_G6blk);
  a.dD = d_D; a.coeff = coeff; a.x = d_x; a.y = d_y;
  const int ncols = ((nx + bx - 1) / bx) * ((ny + by - 1) / by);
  const int nseg = 1 + (std::max(nz - lz0, 0) + lz - 1) / lz;
  const int nwg = d_items ? nitems : ncols * nseg;
  const size_t lds = march_ks_lds_bytes(P, bx, by, 0, false);
#define X(PP, BXX, BYY) \
  if (P == PP && bx == BXX && by == BYY) return launch_ks_t<PP, BXX, BYY, false>(a, dm, nwg, lds, s);
  WF_KS_SHAPES(X)
#undef X
  set_error("march_ks: cross-section not compiled");
  return WF_ERR_UNSUPPORTED;
}

int launch_stiffness_march_ks_idx(int P, int bx, int by, const MarchPlanDev& pd, const double* d_G6blk, const double* d_D,
                                  const DMat& dm, double coeff, const double* d_x, double* d_y, const int32_t* d_items,
                                  int nitems, hipStream_t s)
{
  KSArgs a{};
  a.lz = pd.lz;
  a.tile_size = pd.tile_size;
  a.item_base = pd.d_item_base;
  a.item_pattern = pd.d_item_pattern;
  a.item_layers = pd.d_item_layers;
  a.pat_off = pd.d_pat_off;
  a.items = d_items;
  a.G6blk = reinterpret_cast<const double2*>(d_G6blk);
  a.dD = d_D; a.coeff = coeff; a.x = d_x; a.y = d_y;
  const int nwg = d_items ? nitems : pd.nitems;
  const size_t lds = march_ks_lds_bytes(P, bx, by, pd.lz, true);
#define X(PP, BXX, BYY) \
  if (P == PP && bx == BXX && by == BYY) return launch_ks_t<PP, BXX, BYY, true>(a, dm, nwg, lds, s);
  WF_KS_SHAPES(X)
#undef X
  set_error("march_ks: cross-section not compiled");
  return WF_ERR_UNSUPPORTED;
}

}  // namespace wf
