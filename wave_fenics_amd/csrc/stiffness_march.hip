// Structured-box stiffness operator, "marching" form: the production kernel for
// y += -c0^2 K x on lexicographically numbered box meshes.
//
// One 256-thread workgroup owns a column of BX x BY cells and marches through
// `lz` cell layers in z.  Per layer it runs the column-thread sum-factorised
// core (stiffness_core.h) on BX*BY cells at once and
//   * streams the layer's symmetric geometry (the dominant HBM stream, 48 B per
//     point) straight into registers with 16-byte lane-contiguous loads, issued
//     one full layer AHEAD of its use (register double buffer), together with the
//     next layer's x planes and the y values this layer will update, so HBM
//     latency never sits on the per-layer critical path;
//   * keeps the z-direction partial sums of the shared dof plane in a register
//     (the same thread owns the same (i, j) column in every layer), combines the
//     cells of the layer through an LDS tile without LDS atomics (fixed order),
//     and adds the finished planes to y with hardware fp64 atomics
//     (global_atomic_add_f64, no return) in whole lattice rows: one atomic per
//     tile dof and layer instead of one per element-local dof (1.2 instead of
//     1.95 atomics per dof at P4), issued as contiguous runs of P*BX+1 doubles.
//     Measured on MI355X: mixing plain read-modify-write of column-interior dofs
//     with atomics on the column faces (which share 128-byte lines) is ~2x
//     slower than making every update an atomic, so every update is an atomic.
// Reference semantics: StiffnessOperator::operator() (common/operators.hpp:183-200),
// y accumulated, coefficient -c0^2 (operators.hpp:114-115).
#include <cstdlib>

#include <type_traits>

#include "stiffness_core.h"

namespace wf {

// Measured and rejected on cfg2 (P4, 54^3 cells; this kernel 0.226 ms):
//  * one geometry register set (168 VGPRs, three workgroups per CU, flush deferred past the next
//    layer's phase 1, two barriers per layer), reloaded either after phase 1 of the element kernel
//    or triple by triple inside it (prefetch distance a whole layer): 0.246 / 0.249 ms with three
//    workgroups per CU, 0.234 / 0.230 ms with two -- more resident workgroups make the kernel
//    slower, not faster (a later spill-free build of the second variant, 166 VGPRs, flush values
//    parked in LDS: 0.228 ms with three per CU against 0.221 ms for this kernel);
//  * giving each XCD a contiguous range of columns (workgroup b -> item (b % 8) * n / 8 + b / 8)
//    instead of the round-robin order: 0.247 ms.
// Diagnostic build (tools/march_trace.sh): per-wave timestamps of the phases of the first layers of
// the first 512 workgroups, 100 MHz constant clock.
#ifndef WF_MARCH_BRANCHLESS_ROTATE
#define WF_MARCH_BRANCHLESS_ROTATE 1
#endif
#ifndef WF_MARCH_TILE_ADD
#define WF_MARCH_TILE_ADD 1
#endif
#ifdef WF_MARCH_TRACE
constexpr int kMarchTraceIters = 12, kMarchTraceSlots = 6;
__device__ unsigned long long g_march_trace[512 * 4 * kMarchTraceIters * kMarchTraceSlots];
#define WF_MTR(slot)                                                                                          \
  if ((threadIdx.x & 63) == 0 && trace_it < kMarchTraceIters && blockIdx.x < 512)                              \
  g_march_trace[((blockIdx.x * 4 + (threadIdx.x >> 6)) * kMarchTraceIters + trace_it) * kMarchTraceSlots + (slot)] = wall_clock64()
#else
#define WF_MTR(slot)
#endif

template <int P, int BX, int BY>
__global__ __launch_bounds__(256, 2) void k_stiffness_march(int nx, int ny, int nz, int lz, int lz0,
                                                            const double2* __restrict__ G6blk,
                                                            const double* __restrict__ dD, DMat dm,
                                                            double coeff, const double* __restrict__ x,
                                                            double* __restrict__ y,
                                                            const int32_t* __restrict__ items, int ablate_arg)
{
  [[maybe_unused]] const int ablate = WF_ABLATE_FLAGS(ablate_arg);
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  constexpr int CB = BX * BY, NT = CB * n2;
  constexpr int TX = P * BX + 1, TY = P * BY + 1, TP = TX * TY;
  constexpr int NPOS = (P * TP + 255) / 256;        // flush / x-prefetch positions per thread
  constexpr int NPOS0 = ((P + 1) * TP + 255) / 256; // prologue x positions per thread
  constexpr int NCP = (TP + 255) / 256;             // positions of one plane per thread
  static_assert(NT <= 256, "column does not fit a 256-thread workgroup");

  // x planes of the layer + a dump row: threads whose last position lies past the tile store there, so that the
  // stores that consume the prefetched x registers sit in straight-line code (see (c))
  __shared__ __attribute__((aligned(16))) double Ux[(P + 1) * TP + (WF_MARCH_BRANCHLESS_ROTATE ? 256 : 0)];
#if WF_MARCH_TILE_ADD
  // results of the layer's cells, summed where they share a face (ds_add_f64): planes 0..P-1 of the tile.  The
  // flush of a position is then one LDS read (+ the zero for the next layer) and one global atomic; as a gather
  // over the (up to four) cells' private results it decoded its position and took four guarded LDS reads.
  __shared__ __attribute__((aligned(16))) double O[P * TP > CB * n2 ? P * TP : CB * n2];
#else
  __shared__ __attribute__((aligned(16))) double O[CB * P * n2];     // per-cell results, planes 0..P-1
#endif
  __shared__ __attribute__((aligned(16))) double Fr[CB * nd];
  __shared__ __attribute__((aligned(16))) double Fs[CB * nd];
  __shared__ __attribute__((aligned(16))) double sD[n * n];

  const int t = threadIdx.x;
  const int nbx = (nx + BX - 1) / BX, nby = (ny + BY - 1) / BY;
  const int ncols = nbx * nby;
  // work item = (column, z segment); an optional item list selects a subset (the
  // interior / interface split used to overlap the ghost exchange)
  const int item = items ? items[blockIdx.x] : (int)blockIdx.x;
  const int col = item % ncols, seg = item / ncols;
  const int Bx = col % nbx, By = col / nbx;
  // z segments: [0, lz0), then pieces of lz layers (lz0 = lz unless the operator is split for the
  // ghost exchange: a short first segment keeps the work that reads the z ghost plane small)
  const int z0 = seg == 0 ? 0 : lz0 + (seg - 1) * lz, z1 = min(nz, seg == 0 ? lz0 : z0 + lz);
  const bool active = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;
  const int lx = cl % BX, ly = cl / BX;

  const int NX = P * nx + 1, NY = P * ny + 1;
  const size_t plane = (size_t)NX * NY;
  const int I0 = P * Bx * BX, J0 = P * By * BY;
  const int EX = min(TX, NX - I0), EY = min(TY, NY - J0);

  // ---- per-thread tile positions (identical in every layer) -----------------
  // position m: (I, J, pl) with pl in [0, P); flush: lattice plane P*kz + pl;
  // x prefetch of the next layer: LDS plane pl + 1, lattice plane P*(kz+1) + pl + 1.
  int32_t poff[NPOS];   // lattice offset of (I0+I, J0+J, pl) relative to the layer's first plane; -1 = outside
#pragma unroll
  for (int m = 0; m < NPOS; ++m) {
    const int pos = t + 256 * m;
    const int pl = pos / TP, r = pos % TP, J = r / TX, I = r % TX;
    poff[m] = -1;
    if (pos < P * TP && I < EX && J < EY)
      poff[m] = (int32_t)((size_t)(I0 + I) + (size_t)NX * (J0 + J) + plane * pl);
  }

  const int32_t poff0 = (int32_t)((size_t)I0 + (size_t)NX * J0);   // always inside the mesh

  // ---- prologue: geometry of layer z0 -> registers, x planes 0..P -> LDS ------
  auto load_g = [&](double2 (&g)[n][3], int kz, int k0 = 0, int k1 = P + 1) {
    size_t blk = (size_t)Bx + (size_t)nbx * (By + (size_t)nby * kz);
    if (ablate & 2) blk = 0;   // diagnostic: geometry served from L2
    const double2* gp = G6blk + (blk * n * 3) * (size_t)NT + (t < NT ? t : NT - 1);
    if (ablate & 16) {   // diagnostic: ordinary (temporal) geometry loads
#pragma unroll
      for (int k = 0; k < n; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (k >= k0 && k < k1) g[k][p] = gp[(size_t)(k * 3 + p) * NT];
    } else {
#pragma unroll
      for (int k = 0; k < n; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (k >= k0 && k < k1) g[k][p] = load_stream(gp + (size_t)(k * 3 + p) * NT);
    }
  };
  // the next layer's geometry is requested in three instalments spread over the layer (k planes
  // [0, G1), [G1, G2), [G2, n)) instead of one burst of 3 n loads at the top
  // (P4: 0.2209 -> 0.2182 ms; P2 has a third of the geometry per layer and was 5 % slower with it)
  constexpr int G1 = P >= 4 ? (n + 1) / 3 : n, G2 = P >= 4 ? (2 * n + 1) / 3 : n;
  double2 gA[n][3], gB[n][3];
  load_g(gA, z0);
  if (t < n * n) sD[t] = dD[t];
#if WF_MARCH_TILE_ADD
  for (int e = t; e < P * TP; e += 256) O[e] = 0.0;
#endif
  {
    const size_t base = plane * (size_t)(P * z0);
#pragma unroll
    for (int m = 0; m < NPOS0; ++m) {
      const int pos = t + 256 * m;
      if (pos < (P + 1) * TP) {
        const int pl = pos / TP, r = pos % TP, J = r / TX, I = r % TX;
        double v = 0.0;
        if (I < EX && J < EY) {
          const size_t gi = base + (size_t)(I0 + I) + (size_t)NX * (J0 + J) + plane * pl;
          v = (ablate & 4) ? 1.0 + pos : x[gi];
        }
        Ux[pos] = v;
      }
    }
  }
  __syncthreads();

  double carry = 0.0;
  const double* Uc = Ux + (P * ly) * TX + P * lx;

  [[maybe_unused]] int trace_it = 0;
  // One layer; `gcur` holds its geometry, `gnext` receives the next layer's.  The two register sets
  // swap roles from layer to layer (the loop below is unrolled by two): a copy gcur = gnext is
  // placed by the compiler at the loop's back edge, behind the flush, and then waits for the
  // prefetch AND the atomics in front of it.
  // `has_next` is a compile-time property of the layer body (two copies, chosen by one uniform branch per layer):
  // with `if (has_next)` around each prefetch instalment the compiler merges, at every join, the wait-count state
  // of the path that issued the geometry loads with the one that did not -- in which the x loads are the YOUNGEST
  // pending loads -- and the rotate then waited with vmcnt(2)/(1)/(0), i.e. for the geometry instalments issued
  // after the x loads too, instead of vmcnt(11)/(10)/(9).
  auto layer = [&](auto hn_tag, double2 (&gcur)[n][3], double2 (&gnext)[n][3], int kz) {
    const bool has_next = hn_tag;   // a compile-time constant for the specialised copies (P >= 4)
    const size_t base = plane * (size_t)(P * kz);   // first lattice plane of this layer
    WF_MTR(0);

    // (a) next layer's x planes and geometry: in flight during this layer's arithmetic.  The loads
    // are unconditional, on clamped addresses (positions outside the mesh read the tile's first entry,
    // idle threads the last thread's geometry, the last layer re-reads its own x planes): a guard around a
    // load is a branch, and at the join the compiler waits for every memory operation still
    // pending -- here the previous layer's atomics -- before it issues the prefetch.  What is live
    // is decided where the registers are consumed, in (c).
    const int kzn = has_next ? kz + 1 : kz;
    double xn[NPOS];
    {
      const double* xb = x + plane * (size_t)(P * kzn) + plane;   // first prefetched plane: P * kzn + 1
#pragma unroll
      for (int m = 0; m < NPOS; ++m) xn[m] = (ablate & 4) ? 1.0 + m : xb[poff[m] >= 0 ? poff[m] : poff0];
    }
    if (has_next) load_g(gnext, kzn, 0, G1);   // uniform branch; a self-prefetch in the last layer would re-read 1/lz of the geometry

    WF_MTR(1);
    // (b) element kernels of the layer
    double out[n];
    if (ablate & 8) {
      stiffness_column<P>(Uc, TP, TX, Fr + cl * nd, Fs + cl * nd, sD, dm, gcur, coeff, i, j, active, out, ablate);
      if (has_next) load_g(gnext, kzn, G1, G2);
    } else {
      double ft[n];
      stiffness_phase1<P>(Uc, TP, TX, Fr + cl * nd, Fs + cl * nd, sD, dm, gcur, coeff, i, j, active, ft);
      __syncthreads();
      if (has_next) load_g(gnext, kzn, G1, G2);
      stiffness_phase2<P>(Fr + cl * nd, Fs + cl * nd, sD, dm, ft, i, j, active, out);
    }
    WF_MTR(2);
    double xcp[NCP];
#pragma unroll
    for (int m = 0; m < NCP; ++m) {
      const int pos = t + 256 * m;
      xcp[m] = pos < TP ? Ux[P * TP + pos] : 0.0;
    }
    if (active) {
      out[0] += carry;        // z-shared plane: partial sum of the layer below
      carry = out[P];
#if WF_MARCH_TILE_ADD
      double* To = O + (P * ly + j) * TX + P * lx + i;
#pragma unroll
      for (int k = 0; k < P; ++k) __hip_atomic_fetch_add(To + k * TP, out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
#pragma unroll
      for (int k = 0; k < P; ++k) O[(cl * P + k) * n2 + ji] = out[k];
#endif
    }
    __syncthreads();
    WF_MTR(3);

    // (c) rotate the x planes.  This consumes the prefetched x registers and so carries the wait for
    // their loads; it comes BEFORE the flush because loads and
    // atomics share vmcnt on gfx9 and the compiler waits for vmcnt(0) once both kinds are pending:
    // placed after the flush, every layer waited for the round trip of its own atomics.
    if (has_next) {
#if WF_MARCH_BRANCHLESS_ROTATE
      // No per-lane branches between the prefetch and its consumers: behind an exec-masked branch the compiler's
      // wait-count bookkeeping fell back to vmcnt(0), i.e. the rotate waited for the geometry instalments issued
      // AFTER the x loads as well (ISA: s_waitcnt vmcnt(1) / (2) / (0) in front of the three stores).  Only a
      // thread's last position can lie outside the tile; it is stored to the dump row instead.
#pragma unroll
      for (int m = 0; m < NCP; ++m) {
        const int pos = t + 256 * m;
        Ux[(256 * (m + 1) <= TP || pos < TP) ? pos : (P + 1) * TP + t] = xcp[m];
      }
#pragma unroll
      for (int m = 0; m < NPOS; ++m) {
        const int pos = t + 256 * m;
        Ux[(256 * (m + 1) <= P * TP || pos < P * TP) ? TP + pos : (P + 1) * TP + t] = poff[m] >= 0 ? xn[m] : 0.0;
      }
#else
#pragma unroll
      for (int m = 0; m < NCP; ++m) {
        const int pos = t + 256 * m;
        if (pos < TP) Ux[pos] = xcp[m];
      }
#pragma unroll
      for (int m = 0; m < NPOS; ++m) {
        const int pos = t + 256 * m;
        if (pos < P * TP) Ux[TP + pos] = poff[m] >= 0 ? xn[m] : 0.0;
      }
#endif
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the LDS writes (and the wait for xn) above the atomics
    if (has_next) load_g(gnext, kzn, G2, n);
    WF_MTR(4);
    // (d) combine the cells of the layer (fixed order) and add the finished planes to y
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
#if WF_MARCH_TILE_ADD
      const int pos = t + 256 * m;
      double v = 0.0;
      if (pos < P * TP) {
        v = O[pos];
        O[pos] = 0.0;
      }
      if (poff[m] < 0) continue;
#else
      if (poff[m] < 0) continue;
      const int pos = t + 256 * m;
      const int pl = pos / TP, r = pos % TP, J = r / TX, I = r % TX;
      const int ca = I / P, ia = I % P, cb = J / P, jb = J % P;
      double v = 0.0;
      if (cb < BY) {
        if (ca < BX) v += O[((cb * BX + ca) * P + pl) * n2 + jb * n + ia];
        if (ia == 0 && ca > 0) v += O[((cb * BX + ca - 1) * P + pl) * n2 + jb * n + P];
      }
      if (jb == 0 && cb > 0) {
        if (ca < BX) v += O[(((cb - 1) * BX + ca) * P + pl) * n2 + P * n + ia];
        if (ia == 0 && ca > 0) v += O[(((cb - 1) * BX + ca - 1) * P + pl) * n2 + P * n + P];
      }
#endif
      double* dst = y + base + poff[m];
      if (ablate & 1) {
        if (v == 1.2345e300) *dst = v;
      } else {
        unsafeAtomicAdd(dst, v);
      }
    }
    WF_MTR(5);
    ++trace_it;

    __syncthreads();
  };
  using HasNext = std::integral_constant<bool, true>;
  using IsLast = std::integral_constant<bool, false>;
  // P <= 3 issues the whole prefetch in one burst at the top of the layer and measured 6 % SLOWER with the two
  // copies (P2 0.3125 -> 0.3325 ms; P3 unchanged; P4 0.2017 -> 0.1954 ms): one body with a run-time flag there.
  constexpr bool kTwoCopies = P >= 4;
  for (int kz = z0; kz < z1; kz += 2) {
    if constexpr (!kTwoCopies) {
      layer(kz + 1 < z1, gA, gB, kz);   // (plain bools: a wrapper struct with operator bool compiled to 6 % slower code)
      if (kz + 1 < z1) layer(kz + 2 < z1, gB, gA, kz + 1);
    } else if (kz + 1 < z1) {
      layer(HasNext{}, gA, gB, kz);
      if (kz + 2 < z1)
        layer(HasNext{}, gB, gA, kz + 1);
      else
        layer(IsLast{}, gB, gA, kz + 1);
    } else {
      layer(IsLast{}, gA, gB, kz);
    }
  }

  // ---- epilogue: the last (carried) plane ------------------------------------
  if (active) O[cl * n2 + ji] = carry;
  __syncthreads();
  {
    const size_t base = plane * (size_t)(P * z1);
#pragma unroll
    for (int m = 0; m < NCP; ++m) {
      const int pos = t + 256 * m;
      if (pos >= TP) continue;
      const int J = pos / TX, I = pos % TX;
      if (I >= EX || J >= EY) continue;
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
      double* dst = y + base + (size_t)(I0 + I) + (size_t)NX * (J0 + J);
      if (ablate & 1) {
        if (v == 1.2345e300) *dst = v;
      } else {
        unsafeAtomicAdd(dst, v);
      }
    }
  }
}

static int march_ablate()
{
#ifdef WF_DIAG
  const char* e = std::getenv("WF_ABLATE");
  return e ? std::atoi(e) : 0;
#else
  return 0;
#endif
}

template <int P, int BX, int BY>
static int launch_march_t(int nx, int ny, int nz, int lz, int lz0, const double* d_G6blk, const double* d_D,
                          const DMat& dm, double coeff, const double* d_x, double* d_y, const int32_t* d_items,
                          int nitems, hipStream_t s)
{
  const int ncols = ((nx + BX - 1) / BX) * ((ny + BY - 1) / BY);
  const int nseg = 1 + (std::max(nz - lz0, 0) + lz - 1) / lz;
  const int nwg = d_items ? nitems : ncols * nseg;
  if (nwg == 0) return WF_OK;
  hipLaunchKernelGGL((k_stiffness_march<P, BX, BY>), dim3((unsigned)nwg), dim3(256), 0, s, nx, ny, nz, lz, lz0,
                     reinterpret_cast<const double2*>(d_G6blk), d_D, dm, coeff, d_x, d_y, d_items, march_ablate());
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("stiffness_march launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

// The (BX, BY) column cross-sections compiled per degree.  index = variant.
bool march_variant(int P, int variant, int* bx, int* by)
{
  static const int tab[8][3][2] = {
      {{0, 0}, {0, 0}, {0, 0}},
      {{8, 8}, {4, 4}, {8, 4}},   // P1: 64 / 16 / 32 cells
      {{5, 5}, {3, 3}, {7, 4}},   // P2: 25 / 9 / 28 cells
      {{4, 4}, {3, 3}, {4, 2}},   // P3: 16 / 9 / 8 cells
      {{3, 3}, {5, 2}, {2, 2}},   // P4: 9 / 10 / 4 cells
      {{0, 0}, {0, 0}, {0, 0}},   // P5..P7: the k-split kernel (stiffness_march_ks.hip); this one ran at one wave per
      {{0, 0}, {0, 0}, {0, 0}},   // SIMD there (two geometry register sets per column do not fit 256 VGPRs) and
      {{0, 0}, {0, 0}, {0, 0}},   // measured 0.26-0.51 ms at 10 M dofs against 0.17-0.22 ms
  };
  if (P < 1 || P > 4 || variant < 0 || variant > 2) return false;
  *bx = tab[P][variant][0];
  *by = tab[P][variant][1];
  return true;
}

#define WF_MARCH_CASE(PP, V, BXX, BYY) \
  if (P == PP && variant == V) return launch_march_t<PP, BXX, BYY>(nx, ny, nz, lz, lz0, d_G6blk, d_D, dm, coeff, d_x, d_y, d_items, nitems, s);

int launch_stiffness_march(int P, int variant, int nx, int ny, int nz, int lz, int lz0, const double* d_G6blk,
                           const double* d_D, const DMat& dm, double coeff, const double* d_x, double* d_y,
                           const int32_t* d_items, int nitems, hipStream_t s)
{
  if ((size_t)nx * ny * nz == 0) return WF_OK;
  WF_MARCH_CASE(1, 0, 8, 8) WF_MARCH_CASE(1, 1, 4, 4) WF_MARCH_CASE(1, 2, 8, 4)
  WF_MARCH_CASE(2, 0, 5, 5) WF_MARCH_CASE(2, 1, 3, 3) WF_MARCH_CASE(2, 2, 7, 4)
  WF_MARCH_CASE(3, 0, 4, 4) WF_MARCH_CASE(3, 1, 3, 3) WF_MARCH_CASE(3, 2, 4, 2)
  WF_MARCH_CASE(4, 0, 3, 3) WF_MARCH_CASE(4, 1, 5, 2) WF_MARCH_CASE(4, 2, 2, 2)
  set_error("stiffness_march: unsupported degree/variant");
  return WF_ERR_UNSUPPORTED;
}

}  // namespace wf

#ifdef WF_MARCH_TRACE
extern "C" int wf_debug_march_trace(unsigned long long* host, size_t n)
{
  void* sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(wf::g_march_trace)) != hipSuccess) return -1;
  if (n > sizeof(wf::g_march_trace) / 8) n = sizeof(wf::g_march_trace) / 8;
  return hipMemcpy(host, sym, n * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
