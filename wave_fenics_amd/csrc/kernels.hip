// libwavehip device kernels (gfx950 / MI355X only).
//
// Hot path: the sum-factorised hexahedral stiffness operator
//     y += -c0^2 * sum_cells  P_c^T  D^T (G .* (D P_c x))
// replacing the dense per-cell loop of common/operators.hpp:113-133 (skernel)
// and the gather/kernel/scatter launch triple of common/cuda/mass.hpp:76-95.
//
// Thread mapping ("column threads", 64-wide waves): one thread owns the column
// (i, j, *) of one cell and keeps its n = P+1 values along z in registers; a
// 256-thread workgroup processes CB = floor(256 / n^2) cells at once (P4: 10
// cells, 250 of 256 lanes busy).  x- and y-direction contractions go through an
// LDS tile of the cell's dofs, the z-direction contraction stays in registers.
// The per-point symmetric geometry tensor (6 doubles) is the dominant HBM
// stream; it is stored pre-blocked per workgroup so that every lane issues
// 16-byte loads that are contiguous across the wave, and all loads of a batch
// are issued before any arithmetic.
#include <cstdlib>

#include "common.h"

namespace wf {

// --------------------------------------------------------------------------
// small helpers
// --------------------------------------------------------------------------
__device__ __forceinline__ double clamp101(double v)
{
  // xt::isclose(v, b) with rtol 1e-5, atol 1e-8, applied for b = -1, 0, 1
  // (common/precomputation.hpp:105-107)
  if (fabs(v + 1.0) <= 1e-8 + 1e-5) v = -1.0;
  if (fabs(v) <= 1e-8) v = 0.0;
  if (fabs(v - 1.0) <= 1e-8 + 1e-5) v = 1.0;
  return v;
}

__device__ __forceinline__ double det3(const double* A)
{
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6])
         + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// Jacobian, |det J| w and G = (J^-1 detJ) J^-T at one point of a trilinear
// hexahedron (common/precomputation.hpp:83-100).  xv: the cell's 8 vertices
// [v][3], vertex v = a + 2b + 4c; X: reference point.
__device__ __forceinline__ void hex_point_geometry(const double (*xv)[3], double X0, double X1, double X2,
                                                   double w, int use_fabs, int do_clamp, double* G9,
                                                   double* detJw)
{
  double J[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) J[i] = 0.0;
  const double f0[2] = {1.0 - X0, X0}, f1[2] = {1.0 - X1, X1}, f2[2] = {1.0 - X2, X2};
  const double g[2] = {-1.0, 1.0};
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const int a = v & 1, b = (v >> 1) & 1, c = (v >> 2) & 1;
    // the reference clamps the tabulated cmap derivatives to -1/0/1
    // (precomputation.hpp:56-58)
    double d0 = g[a] * f1[b] * f2[c];
    double d1 = f0[a] * g[b] * f2[c];
    double d2 = f0[a] * f1[b] * g[c];
    if (do_clamp) {
      d0 = clamp101(d0);
      d1 = clamp101(d1);
      d2 = clamp101(d2);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      J[i * 3 + 0] += xv[v][i] * d0;
      J[i * 3 + 1] += xv[v][i] * d1;
      J[i * 3 + 2] += xv[v][i] * d2;
    }
  }
  double d = det3(J);
  const double idet = 1.0 / d;
  if (use_fabs) d = fabs(d);
  d *= w;
  *detJw = d;
  if (G9) {
    double Ji[9];
    Ji[0] = (J[4] * J[8] - J[5] * J[7]) * idet;
    Ji[1] = (J[2] * J[7] - J[1] * J[8]) * idet;
    Ji[2] = (J[1] * J[5] - J[2] * J[4]) * idet;
    Ji[3] = (J[5] * J[6] - J[3] * J[8]) * idet;
    Ji[4] = (J[0] * J[8] - J[2] * J[6]) * idet;
    Ji[5] = (J[2] * J[3] - J[0] * J[5]) * idet;
    Ji[6] = (J[3] * J[7] - J[4] * J[6]) * idet;
    Ji[7] = (J[1] * J[6] - J[0] * J[7]) * idet;
    Ji[8] = (J[0] * J[4] - J[1] * J[3]) * idet;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += (Ji[i * 3 + k] * d) * Ji[j * 3 + k];
        G9[i * 3 + j] = do_clamp ? clamp101(s) : s;
      }
  }
}

// Blocked symmetric geometry layout shared by the stiffness kernels:
//   G6blk[batch][k][p][t][e],  t = cl * n^2 + (j * n + i) < NT = CB * n^2,
//   (p, e) -> component: (0,0)=G00 (0,1)=G01 (1,0)=G02 (1,1)=G11 (2,0)=G12 (2,1)=G22
__device__ __forceinline__ size_t g6_index(int n, int NT, size_t batch, int k, int p, int t, int e)
{
  return ((((batch * n + k) * 3 + p) * (size_t)NT + t) * 2) + e;
}

__device__ __forceinline__ void store_g6(double* G6blk, int n, int NT, size_t batch, int k, int t,
                                         const double* G9)
{
  G6blk[g6_index(n, NT, batch, k, 0, t, 0)] = G9[0];
  G6blk[g6_index(n, NT, batch, k, 0, t, 1)] = G9[1];
  G6blk[g6_index(n, NT, batch, k, 1, t, 0)] = G9[2];
  G6blk[g6_index(n, NT, batch, k, 1, t, 1)] = G9[4];
  G6blk[g6_index(n, NT, batch, k, 2, t, 0)] = G9[5];
  G6blk[g6_index(n, NT, batch, k, 2, t, 1)] = G9[8];
}

// --------------------------------------------------------------------------
// geometry kernels (setup; a2/a13)
// --------------------------------------------------------------------------
__global__ void k_geometry_hex(int n, int CB, int ncells, const double* __restrict__ xverts,
                               const int32_t* __restrict__ geom_dofmap, const double* __restrict__ pts,
                               const double* __restrict__ wts, int use_fabs, int do_clamp,
                               double* __restrict__ G9out, double* __restrict__ G6blk,
                               double* __restrict__ detJ)
{
  const int nd = n * n * n;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)ncells * nd) return;
  const int c = (int)(gid / nd), q = (int)(gid % nd);
  const int i = q % n, j = (q / n) % n, k = q / (n * n);
  double xv[8][3];
  for (int v = 0; v < 8; ++v) {
    const int32_t vi = geom_dofmap[(size_t)c * 8 + v];
    xv[v][0] = xverts[(size_t)vi * 3 + 0];
    xv[v][1] = xverts[(size_t)vi * 3 + 1];
    xv[v][2] = xverts[(size_t)vi * 3 + 2];
  }
  double G9[9], d;
  const bool wantG = (G9out != nullptr) || (G6blk != nullptr);
  hex_point_geometry(xv, pts[i], pts[j], pts[k], wts[i] * wts[j] * wts[k], use_fabs, do_clamp,
                     wantG ? G9 : nullptr, &d);
  if (detJ) detJ[gid] = d;
  if (G9out)
    for (int m = 0; m < 9; ++m) G9out[gid * 9 + m] = G9[m];
  if (G6blk) {
    const int NT = CB * n * n;
    store_g6(G6blk, n, NT, c / CB, k, (c % CB) * n * n + j * n + i, G9);
  }
}

// the same for a subset of slots of the blocked layout: cell c of the list goes to slot slot_of[c]
// (batch = slot / CB); used by the indexed marching operator, whose slot array has gaps
// cell_sign (may be null): -1 for cells whose vertices were handed over in a reflected frame (lattice
// plan, generic_plan.cpp).  G = K K^T |det J| w does not depend on the frame; without the fabs
// (spectral_mass.hpp:58-64) det J changes sign under a reflection and is corrected here.
__global__ void k_geometry_hex_slots(int n, int CB, int ncells, const double* __restrict__ xverts,
                                     const int32_t* __restrict__ geom_dofmap, const int32_t* __restrict__ slot_of,
                                     const int8_t* __restrict__ cell_sign,
                                     const double* __restrict__ pts, const double* __restrict__ wts, int use_fabs,
                                     int do_clamp, double* __restrict__ G6blk)
{
  const int nd = n * n * n;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)ncells * nd) return;
  const int c = (int)(gid / nd), q = (int)(gid % nd);
  const int i = q % n, j = (q / n) % n, k = q / (n * n);
  double xv[8][3];
  for (int v = 0; v < 8; ++v) {
    const int32_t vi = geom_dofmap[(size_t)c * 8 + v];
    xv[v][0] = xverts[(size_t)vi * 3 + 0];
    xv[v][1] = xverts[(size_t)vi * 3 + 1];
    xv[v][2] = xverts[(size_t)vi * 3 + 2];
  }
  double G9[9], d;
  hex_point_geometry(xv, pts[i], pts[j], pts[k], wts[i] * wts[j] * wts[k], use_fabs, do_clamp, G9, &d);
  if (!use_fabs && cell_sign && cell_sign[c] < 0)
    for (int m = 0; m < 9; ++m) G9[m] = -G9[m];
  const int slot = slot_of[c];
  store_g6(G6blk, n, CB * n * n, slot / CB, k, (slot % CB) * n * n + j * n + i, G9);
}

// reference-layout G[ncells][nq][3][3] -> blocked symmetric layout (upper triangle)
__global__ void k_pack_G6(int n, int CB, int ncells, const double* __restrict__ G9in,
                          double* __restrict__ G6blk)
{
  const int nd = n * n * n;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (size_t)ncells * nd) return;
  const int c = (int)(gid / nd), q = (int)(gid % nd);
  const int i = q % n, j = (q / n) % n, k = q / (n * n);
  double G9[9];
  for (int m = 0; m < 9; ++m) G9[m] = G9in[gid * 9 + m];
  store_g6(G6blk, n, CB * n * n, c / CB, k, (c % CB) * n * n + j * n + i, G9);
}

// Structured box: one thread per (cell, point); the cell's vertices come from the
// vertex lattice.  Writes the blocked G of the box stiffness kernel (block =
// bx*by*bz cells, padded blocks stay zero) and/or accumulates the lumped mass
// diagonal m[g] += |det J| w on the dof lattice.
__global__ void k_geometry_box(int n, int nx, int ny, int nz, int bx, int by, int bz,
                               const double* __restrict__ xverts, const double* __restrict__ pts,
                               const double* __restrict__ wts, int use_fabs, int do_clamp,
                               double* __restrict__ G6blk, double* __restrict__ mdiag)
{
  const int P = n - 1, nd = n * n * n;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ncells = (size_t)nx * ny * nz;
  if (gid >= ncells * nd) return;
  const int c = (int)(gid / nd), q = (int)(gid % nd);
  const int i = q % n, j = (q / n) % n, k = q / (n * n);
  const int cx = c % nx, cy = (c / nx) % ny, cz = c / (nx * ny);
  double xv[8][3];
  for (int v = 0; v < 8; ++v) {
    const size_t vi = (size_t)(cx + (v & 1)) + (size_t)(nx + 1) * ((cy + ((v >> 1) & 1)) + (size_t)(ny + 1) * (cz + ((v >> 2) & 1)));
    xv[v][0] = xverts[vi * 3 + 0];
    xv[v][1] = xverts[vi * 3 + 1];
    xv[v][2] = xverts[vi * 3 + 2];
  }
  double G9[9], d;
  hex_point_geometry(xv, pts[i], pts[j], pts[k], wts[i] * wts[j] * wts[k], use_fabs, do_clamp,
                     G6blk ? G9 : nullptr, &d);
  if (mdiag) {
    const size_t NX = (size_t)P * nx + 1, NY = (size_t)P * ny + 1;
    const size_t g = (size_t)(P * cx + i) + NX * ((size_t)(P * cy + j) + NY * (size_t)(P * cz + k));
    unsafeAtomicAdd(&mdiag[g], d);
  }
  if (G6blk) {
    const int nbx = (nx + bx - 1) / bx, nby = (ny + by - 1) / by;
    const int Bx = cx / bx, By = cy / by, Bz = cz / bz;
    const size_t blk = (size_t)Bx + (size_t)nbx * (By + (size_t)nby * Bz);
    const int cl = (cx % bx) + bx * ((cy % by) + by * (cz % bz));
    const int CB = bx * by * bz;
    store_g6(G6blk, n, CB * n * n, blk, k, cl * n * n + j * n + i, G9);
  }
}

}  // namespace wf
#include "stiffness_core.h"  // stiffness_column<P>
namespace wf {

// --------------------------------------------------------------------------
// generic stiffness: arbitrary tensor-ordered dofmap, atomic scatter
// --------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void k_stiffness_generic(int ncells, const int32_t* __restrict__ dofmap,
                                                           const double2* __restrict__ G6blk,
                                                           const double* __restrict__ dD, DMat dm,
                                                           double coeff, const double* __restrict__ x,
                                                           double* __restrict__ y, int ablate_arg)
{
  [[maybe_unused]] const int ablate = WF_ABLATE_FLAGS(ablate_arg);
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  constexpr int CB = 256 / n2, NT = CB * n2;
  constexpr int NFLAT = (CB * nd + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* U = smem;                 // [CB][nd]
  double* Fr = U + CB * nd;         // [CB][nd]
  double* Fs = Fr + CB * nd;        // [CB][nd]
  double* sD = Fs + CB * nd;        // [n][n]

  const int t = threadIdx.x;
  const size_t batch = blockIdx.x;
  const int cell0 = (int)batch * CB;
  const bool active = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;

  // 1. issue the whole geometry stream of this batch first (16 B per lane,
  //    contiguous across the wave); padded cells read zeros.
  double2 g[n][3];
  if (active) {
    const double2* gp = G6blk + (batch * n * 3) * (size_t)NT + t;
    if (ablate & 2) gp = G6blk + t;   // diagnostic: every workgroup re-reads batch 0 (L2-resident)
#pragma unroll
    for (int k = 0; k < n; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) g[k][p] = load_stream(gp + (size_t)(k * 3 + p) * NT);
  }
  if (t < n * n) sD[t] = dD[t];

  // 2. gather: flat, coalesced dofmap reads; indices stay in registers for the scatter
  int32_t idx[NFLAT];
  const int nvalid = min(CB, ncells - cell0) * nd;
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    idx[m] = -1;
    if (pos < nvalid) {
      idx[m] = dofmap[(size_t)cell0 * nd + pos];
      U[pos] = (ablate & 4) ? 1.0 + idx[m] : x[idx[m]];
    } else if (pos < CB * nd) {
      U[pos] = 0.0;
    }
  }
  __syncthreads();

  double out[n];
  stiffness_column<P>(U + cl * nd, n2, n, Fr + cl * nd, Fs + cl * nd, sD, dm, g, coeff, i, j, active, out, ablate);

  // 3. element results back through LDS (Fr is free after the second barrier
  //    only for this thread's own entries -> use U, whose reads all precede the
  //    first barrier inside stiffness_column), then flat atomic scatter-add.
  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) U[cl * nd + k * n2 + ji] = out[k];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    if (idx[m] >= 0) {
      if (ablate & 1) {   // diagnostic: no scatter (store kept live, never taken)
        if (U[pos] == 1.2345e300) y[idx[m]] = U[pos];
      } else {
        unsafeAtomicAdd(&y[idx[m]], U[pos]);
      }
    }
  }
}

// --------------------------------------------------------------------------
// structured box stiffness: implicit dofmap, block dof tile in LDS, atomics only
// on dofs shared with a neighbouring block
// --------------------------------------------------------------------------
// --------------------------------------------------------------------------
// generic stiffness, batch-unique form (default for arbitrary dofmaps): the host
// lists the unique dofs of every batch of CB cells once (uniq, sorted) and the
// position of every element-local dof in that list (loc, uint16).  x is read once
// per unique dof, the cells of the batch are summed per unique dof in LDS
// (ds_add_f64) and y receives ONE atomic per unique dof, in ascending address
// order -- fewer and better-shaped atomic requests than the element-wise scatter
// (P4, lexicographic numbering: 1025 instead of 1250 per batch, runs of 41
// contiguous doubles instead of 5).
// --------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void k_stiffness_generic_u(int ncells, const int32_t* __restrict__ uoff,
                                                             const int32_t* __restrict__ uniq,
                                                             const uint16_t* __restrict__ loc,
                                                             const double2* __restrict__ G6blk,
                                                             const double* __restrict__ dD, DMat dm,
                                                             double coeff, const double* __restrict__ x,
                                                             double* __restrict__ y, int ablate_arg)
{
  [[maybe_unused]] const int ablate = WF_ABLATE_FLAGS(ablate_arg);
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  constexpr int CB = 256 / n2, NT = CB * n2;
  constexpr int NFLAT = (CB * nd + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* U = smem;                 // [CB][nd]
  double* Fr = U + CB * nd;         // [CB][nd]; also the unique-dof tile before phase 1 and after phase 2
  double* Fs = Fr + CB * nd;        // [CB][nd]
  double* sD = Fs + CB * nd;        // [n][n]

  const int t = threadIdx.x;
  const size_t batch = blockIdx.x;
  const int cell0 = (int)batch * CB;
  const bool active = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;

  double2 g[n][3];
  if (active) {
    const double2* gp = G6blk + (batch * n * 3) * (size_t)NT + t;
    if (ablate & 2) gp = G6blk + t;
#pragma unroll
    for (int k = 0; k < n; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) g[k][p] = load_stream(gp + (size_t)(k * 3 + p) * NT);
  }
  if (t < n * n) sD[t] = dD[t];

  const int u0 = uoff[batch], nu = uoff[batch + 1] - u0;
  for (int u = t; u < nu; u += 256) Fr[u] = (ablate & 4) ? 1.0 + u : x[uniq[u0 + u]];
  uint16_t lc[NFLAT];
  const int nvalid = min(CB, ncells - cell0) * nd;
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    lc[m] = pos < nvalid ? loc[(size_t)cell0 * nd + pos] : (uint16_t)0xFFFF;
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    if (pos < CB * nd) U[pos] = lc[m] != 0xFFFF ? Fr[lc[m]] : 0.0;
  }
  __syncthreads();

  double out[n];
  stiffness_column<P>(U + cl * nd, n2, n, Fr + cl * nd, Fs + cl * nd, sD, dm, g, coeff, i, j, active, out, ablate);

  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) U[cl * nd + k * n2 + ji] = out[k];
  }
  __syncthreads();   // phase 2 has read Fr/Fs, U holds the element results
  for (int u = t; u < nu; u += 256) Fr[u] = 0.0;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    if (lc[m] != 0xFFFF) atomicAdd(&Fr[lc[m]], U[pos]);
  }
  __syncthreads();
  for (int u = t; u < nu; u += 256) {
    if (ablate & 1) {
      if (Fr[u] == 1.2345e300) y[uniq[u0 + u]] = Fr[u];
    } else {
      unsafeAtomicAdd(&y[uniq[u0 + u]], Fr[u]);
    }
  }
}

// The same kernel as a persistent workgroup that walks batches and carries the NEXT batch's unique-dof indices in
// registers: the x gather of a batch is a chain of three dependent loads (uoff -> uniq -> x), and the ablation masks
// showed it costs as much as the whole geometry stream (P4, 10 M dofs: 0.261 ms; without the x loads 0.202, with the
// geometry served from L2 0.204).  With the indices fetched one batch ahead the x loads are issued at the top of a
// batch together with its geometry: one memory latency per batch instead of two and a half.  All loads of the loop
// are unconditional on clamped addresses (a guard is a branch; at its join the compiler waits for every pending
// load), the indices double as the scatter addresses of the batch.
template <int P>
__global__ __launch_bounds__(256) void k_stiffness_generic_up(int ncells, int nbatch, const int32_t* __restrict__ uoff,
                                                              const int32_t* __restrict__ uniq, const uint16_t* __restrict__ loc,
                                                              const double2* __restrict__ G6blk, const double* __restrict__ dD,
                                                              DMat dm, double coeff, const double* __restrict__ x,
                                                              double* __restrict__ y)
{
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  constexpr int CB = 256 / n2, NT = CB * n2;
  constexpr int NFLAT = (CB * nd + 255) / 256;   // element-local dofs per thread; also bounds the unique dofs per thread
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* U = smem;                 // [CB][nd]
  double* Fr = U + CB * nd;         // [CB][nd]; also the unique-dof tile before phase 1 and after phase 2
  double* Fs = Fr + CB * nd;        // [CB][nd]
  double* sD = Fs + CB * nd;        // [n][n]

  const int t = threadIdx.x;
  const bool active = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;
  if (t < n * n) sD[t] = dD[t];

  // unique-dof indices of a batch -> registers (entry u = t + 256 m; past the list: its last entry)
  auto load_uniq = [&](int32_t (&uq)[NFLAT], int& nu, int b) {
    const int u0 = uoff[b];
    nu = uoff[b + 1] - u0;
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      uq[m] = uniq[u0 + (u < nu ? u : nu - 1)];
    }
  };
  int32_t uqa[NFLAT], uqb[NFLAT];
  int nua = 0, nub = 0;
  int batch = blockIdx.x;
  if (batch < nbatch) load_uniq(uqa, nua, batch);

  auto one = [&](int32_t (&uq)[NFLAT], int nu, int32_t (&uqn)[NFLAT], int& nun, int b) {
    const int cell0 = b * CB;
    // (a) this batch's x values, geometry and local positions; the next batch's indices
    double xr[NFLAT];
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) xr[m] = x[uq[m]];
    double2 g[n][3];
    {
      const double2* gp = G6blk + ((size_t)b * n * 3) * (size_t)NT + (active ? t : NT - 1);
#pragma unroll
      for (int k = 0; k < n; ++k)
#pragma unroll
        for (int p = 0; p < 3; ++p) g[k][p] = load_stream(gp + (size_t)(k * 3 + p) * NT);
    }
    uint16_t lc[NFLAT];
    const int nvalid = min(CB, ncells - cell0) * nd;
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int pos = t + 256 * m;
      const uint16_t v = loc[(size_t)cell0 * nd + (pos < nvalid ? pos : 0)];
      lc[m] = pos < nvalid ? v : (uint16_t)0xFFFF;
    }
    const int bn = b + (int)gridDim.x;
    load_uniq(uqn, nun, bn < nbatch ? bn : b);
    // (b) unique tile -> LDS -> element-local layout
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      if (u < nu) Fr[u] = xr[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int pos = t + 256 * m;
      if (pos < CB * nd) U[pos] = lc[m] != 0xFFFF ? Fr[lc[m]] : 0.0;
    }
    __syncthreads();
    double out[n];
    stiffness_column<P>(U + cl * nd, n2, n, Fr + cl * nd, Fs + cl * nd, sD, dm, g, coeff, i, j, active, out, 0);
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) U[cl * nd + k * n2 + ji] = out[k];
    }
    __syncthreads();   // phase 2 has read Fr/Fs, U holds the element results
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      if (u < nu) Fr[u] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int pos = t + 256 * m;
      if (lc[m] != 0xFFFF) atomicAdd(&Fr[lc[m]], U[pos]);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      if (u < nu) unsafeAtomicAdd(&y[uq[m]], Fr[u]);
    }
    __syncthreads();   // Fr is the next batch's unique tile
  };
  while (batch < nbatch) {
    one(uqa, nua, uqb, nub, batch);
    batch += gridDim.x;
    if (batch >= nbatch) break;
    one(uqb, nub, uqa, nua, batch);
    batch += gridDim.x;
  }
}

// Fully pipelined form (P <= 4: two geometry register sets fit at two workgroups per CU): everything batch b + s needs (s = the grid size) --
// geometry, x values, local positions -- is requested while batch b computes, its unique-dof indices one batch
// earlier, the two scalars that locate them one batch earlier still.  There is no `has_next` anywhere: a batch past
// the end is the last batch again (redundant reads, never stored), so the wait counts the compiler derives stay
// exact (stiffness_march.hip on what a uniform branch around a prefetch costs).
template <int P>
__global__ __launch_bounds__(256) void k_stiffness_generic_up2(int ncells, int nbatch, const int32_t* __restrict__ uoff,
                                                               const int32_t* __restrict__ uniq, const uint16_t* __restrict__ loc,
                                                               const double2* __restrict__ G6blk, const double* __restrict__ dD,
                                                               DMat dm, double coeff, const double* __restrict__ x,
                                                               double* __restrict__ y)
{
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  constexpr int CB = 256 / n2, NT = CB * n2;
  constexpr int NFLAT = (CB * nd + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* U = smem;
  double* Fr = U + CB * nd;
  double* Fs = Fr + CB * nd;
  double* sD = Fs + CB * nd;

  const int t = threadIdx.x;
  const bool active = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;
  if (t < n * n) sD[t] = dD[t];
  const int stride = (int)gridDim.x;
  auto clampb = [&](int b) { return b < nbatch ? b : nbatch - 1; };

  // unique-dof indices of the batch whose list is uniq[u0 .. u0 + nu) -> registers (past the list: its last entry)
  auto load_uniq = [&](int32_t (&uq)[NFLAT], int u0, int nu) {
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      uq[m] = uniq[u0 + (u < nu ? u : nu - 1)];
    }
  };
  auto load_batch = [&](double (&xr)[NFLAT], double2 (&g)[n][3], uint16_t (&lc)[NFLAT], const int32_t (&uq)[NFLAT], int b) {
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) xr[m] = x[uq[m]];
    const double2* gp = G6blk + ((size_t)b * n * 3) * (size_t)NT + (active ? t : NT - 1);
#pragma unroll
    for (int k = 0; k < n; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) g[k][p] = load_stream(gp + (size_t)(k * 3 + p) * NT);
    const int cell0 = b * CB, nvalid = min(CB, ncells - cell0) * nd;
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int pos = t + 256 * m;
      const uint16_t v = loc[(size_t)cell0 * nd + (pos < nvalid ? pos : 0)];
      lc[m] = pos < nvalid ? v : (uint16_t)0xFFFF;
    }
  };

  int batch = blockIdx.x;
  if (batch >= nbatch) return;
  // list locations (scalars) of batches b, b + s, b + 2 s; vectors in two sets that swap roles (loop unrolled by two)
  int b1 = clampb(batch + stride), b2 = clampb(batch + 2 * stride);
  int u0c = uoff[batch], nuc = uoff[batch + 1] - u0c;
  int u0n = uoff[b1], nun = uoff[b1 + 1] - u0n;
  int u0nn = uoff[b2], nunn = uoff[b2 + 1] - u0nn;
  int32_t uqA[NFLAT], uqB[NFLAT];
  double xA[NFLAT], xB[NFLAT];
  double2 gA[n][3], gB[n][3];
  uint16_t lcA[NFLAT], lcB[NFLAT];
  load_uniq(uqA, u0c, nuc);
  load_uniq(uqB, u0n, nun);
  load_batch(xA, gA, lcA, uqA, batch);

  // batch b sits in (xr, g, lc); uqn holds the indices of batch b + s, into which (xn, gn, lcn) are loaded; uq (the
  // indices of batch b, no longer needed: its scatter re-reads them from L2) is refilled with those of batch b + 2 s
  auto one = [&](int32_t (&uq)[NFLAT], double (&xr)[NFLAT], double2 (&g)[n][3], uint16_t (&lc)[NFLAT], int32_t (&uqn)[NFLAT],
                 double (&xn)[NFLAT], double2 (&gn)[n][3], uint16_t (&lcn)[NFLAT], int b) {
    const int nu = nuc, u0 = u0c;
    // (a) unique tile -> LDS (consumes xr) -> element-local layout
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      if (u < nu) Fr[u] = xr[m];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int pos = t + 256 * m;
      if (pos < CB * nd) U[pos] = lc[m] != 0xFFFF ? Fr[lc[m]] : 0.0;
    }
    __syncthreads();
    // (b) this batch's scatter addresses again (L2), then the requests for batch b + s and the indices of b + 2 s,
    //     then the list location of b + 3 s: all in flight during the phases
    int32_t us[NFLAT];
    load_uniq(us, u0, nu);
    load_batch(xn, gn, lcn, uqn, clampb(b + stride));
    load_uniq(uq, u0nn, nunn);
    const int b3 = clampb(b + 3 * stride);
    const int u0n3 = uoff[b3], nun3 = uoff[b3 + 1] - u0n3;
    // (c) element kernel
    double out[n];
    stiffness_column<P>(U + cl * nd, n2, n, Fr + cl * nd, Fs + cl * nd, sD, dm, g, coeff, i, j, active, out, 0);
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k) U[cl * nd + k * n2 + ji] = out[k];
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      if (u < nu) Fr[u] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int pos = t + 256 * m;
      if (lc[m] != 0xFFFF) atomicAdd(&Fr[lc[m]], U[pos]);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NFLAT; ++m) {
      const int u = t + 256 * m;
      if (u < nu) unsafeAtomicAdd(&y[us[m]], Fr[u]);
    }
    __syncthreads();
    u0c = u0n;
    nuc = nun;
    u0n = u0nn;
    nun = nunn;
    u0nn = u0n3;
    nunn = nun3;
  };
  while (true) {
    one(uqA, xA, gA, lcA, uqB, xB, gB, lcB, batch);
    batch += stride;
    if (batch >= nbatch) break;
    one(uqB, xB, gB, lcB, uqA, xA, gA, lcA, batch);
    batch += stride;
    if (batch >= nbatch) break;
  }
}

template <int P>
__global__ __launch_bounds__(256) void k_stiffness_box(int nx, int ny, int nz, int bx, int by, int bz,
                                                       const double2* __restrict__ G6blk,
                                                       const double* __restrict__ dD, DMat dm,
                                                       double coeff, const double* __restrict__ x,
                                                       double* __restrict__ y, int ablate_arg)
{
  [[maybe_unused]] const int ablate = WF_ABLATE_FLAGS(ablate_arg);
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  const int CB = bx * by * bz, NT = CB * n2;
  const int TX = P * bx + 1, TY = P * by + 1, TZ = P * bz + 1;
  const int tile = TX * TY * TZ;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ut = smem;                         // [TZ][TY][TX] x tile, later the y tile
  double* Fr = Ut + ((tile + 1) & ~1);       // [CB][nd]
  double* Fs = Fr + CB * nd;                 // [CB][nd]
  double* sD = Fs + CB * nd;                 // [n][n]

  const int t = threadIdx.x;
  const int nbx = (nx + bx - 1) / bx, nby = (ny + by - 1) / by;
  const int Bx = blockIdx.x % nbx, By = (blockIdx.x / nbx) % nby, Bz = blockIdx.x / (nbx * nby);
  const bool inrange = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;
  const int lx = cl % bx, ly = (cl / bx) % by, lz = cl / (bx * by);
  const int cx = Bx * bx + lx, cy = By * by + ly, cz = Bz * bz + lz;
  const bool active = inrange && cx < nx && cy < ny && cz < nz;

  double2 g[n][3];
  if (inrange) {
    const double2* gp = G6blk + ((size_t)blockIdx.x * n * 3) * (size_t)NT + t;
    if (ablate & 2) gp = G6blk + t;
#pragma unroll
    for (int k = 0; k < n; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) g[k][p] = load_stream(gp + (size_t)(k * 3 + p) * NT);
  }
  if (t < n * n) sD[t] = dD[t];

  // dof tile of this block inside the global lattice (clipped at the mesh end)
  const int NX = P * nx + 1, NY = P * ny + 1, NZ = P * nz + 1;
  const int I0 = P * Bx * bx, J0 = P * By * by, K0 = P * Bz * bz;
  const int EX = min(TX, NX - I0), EY = min(TY, NY - J0), EZ = min(TZ, NZ - K0);
  for (int pos = t; pos < tile; pos += 256) {
    const int I = pos % TX, J = (pos / TX) % TY, K = pos / (TX * TY);
    double v = 0.0;
    if (I < EX && J < EY && K < EZ)
      v = (ablate & 4) ? 1.0 + pos : x[(size_t)(I0 + I) + (size_t)NX * ((size_t)(J0 + J) + (size_t)NY * (K0 + K))];
    Ut[pos] = v;
  }
  __syncthreads();

  double out[n];
  const double* Uc = Ut + (P * lz) * TX * TY + (P * ly) * TX + P * lx;
  stiffness_column<P>(Uc, TX * TY, TX, Fr + cl * nd, Fs + cl * nd, sD, dm, g, coeff, i, j, active, out, ablate);

  // y tile: zero, accumulate the cells of the block with LDS atomics
  __syncthreads();
  for (int pos = t; pos < tile; pos += 256) Ut[pos] = 0.0;
  __syncthreads();
  if (active) {
    double* Yc = Ut + (P * lz) * TX * TY + (P * ly) * TX + P * lx;
#pragma unroll
    for (int k = 0; k < n; ++k) atomicAdd(&Yc[k * TX * TY + j * TX + i], out[k]);
  }
  __syncthreads();
  // flush: dofs interior to the block (or on the mesh boundary) are exclusive to
  // this workgroup -> plain read-modify-write; faces shared with a neighbouring
  // block -> global atomic add.
  const bool sx0 = Bx > 0, sy0 = By > 0, sz0 = Bz > 0;
  const bool sx1 = I0 + TX < NX, sy1 = J0 + TY < NY, sz1 = K0 + TZ < NZ;
  for (int pos = t; pos < tile; pos += 256) {
    const int I = pos % TX, J = (pos / TX) % TY, K = pos / (TX * TY);
    if (I < EX && J < EY && K < EZ) {
      const size_t gidx = (size_t)(I0 + I) + (size_t)NX * ((size_t)(J0 + J) + (size_t)NY * (K0 + K));
      const bool shared = (sx0 && I == 0) || (sx1 && I == TX - 1) || (sy0 && J == 0) || (sy1 && J == TY - 1)
                          || (sz0 && K == 0) || (sz1 && K == TZ - 1);
      const double v = Ut[pos];
      if (ablate & 1) {
        if (v == 1.2345e300) y[gidx] = v;
      } else if (shared)
        unsafeAtomicAdd(&y[gidx], v);
      else
        y[gidx] += v;
    }
  }
}

// --------------------------------------------------------------------------
// lumped (spectral) mass: fused gather * detJ -> scatter-add
// replaces the three launches of common/cuda/spectral_mass.hpp:84-89
// --------------------------------------------------------------------------
__global__ void k_mass_lumped(int64_t nentries, const int32_t* __restrict__ dofmap,
                              const double* __restrict__ detJ, const double* __restrict__ x,
                              double* __restrict__ y)
{
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nentries) {
    const int32_t d = dofmap[e];
    unsafeAtomicAdd(&y[d], x[d] * detJ[e]);
  }
}

// batch-unique form of the lumped mass (default): x once per unique dof, the
// batch summed in LDS, one global atomic per unique dof
__global__ __launch_bounds__(256) void k_mass_lumped_u(int ncells, int nd, int CB, const int32_t* __restrict__ uoff,
                                                       const int32_t* __restrict__ uniq,
                                                       const uint16_t* __restrict__ loc,
                                                       const double* __restrict__ detJ,
                                                       const double* __restrict__ x, double* __restrict__ y)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int t = threadIdx.x;
  const int nbatch = (ncells + CB - 1) / CB;
  for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const int u0 = uoff[batch], nu = uoff[batch + 1] - u0;
    double* Xu = smem;
    double* Yu = smem + nu;
    __syncthreads();
    for (int u = t; u < nu; u += 256) {
      Xu[u] = x[uniq[u0 + u]];
      Yu[u] = 0.0;
    }
    __syncthreads();
    const size_t e0 = (size_t)batch * CB * nd;
    const int ne = min(CB, ncells - batch * CB) * nd;
    for (int p = t; p < ne; p += 256) {
      const int l = loc[e0 + p];
      atomicAdd(&Yu[l], Xu[l] * detJ[e0 + p]);
    }
    __syncthreads();
    for (int u = t; u < nu; u += 256) unsafeAtomicAdd(&y[uniq[u0 + u]], Yu[u]);
  }
}

// --------------------------------------------------------------------------
// dense mass Phi^T D Phi with a tensor-product rule, sum-factorised:
// replaces common/cuda/mass_kernel.cu:5-46 and the DGEMM pair of
// demo/gpu_operator/main.cpp:149-155 / demo/gpu_tsmm (k >> m ~ n).  A workgroup
// processes CB cells at once (P2: 32 cells) so that all 256 lanes work in each of
// the six 1-D contraction passes; the passes ping-pong between two LDS tiles.
// phi1: [m][n] row-major (m 1-D quadrature points, n = P+1 1-D nodes).
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mass_dense(int n, int m, int CB, int ncells,
                                                    const int32_t* __restrict__ dofmap,
                                                    const int32_t* __restrict__ uoff,   // batch-unique lists or null
                                                    const int32_t* __restrict__ uniq,
                                                    const uint16_t* __restrict__ loc,
                                                    const double* __restrict__ phi1,
                                                    const double* __restrict__ detJ,
                                                    const double* __restrict__ x, double* __restrict__ y)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int mx = max(n, m), mx3 = mx * mx * mx;
  double* A = smem;                  // ping  [CB][mx^3]
  double* B = A + CB * mx3;          // pong  [CB][mx^3]
  double* sphi = B + CB * mx3;       // [m][n]
  double* Xu = sphi + m * n;         // [CB * nd] unique-dof tile (batch-unique form)
  const int t = threadIdx.x;
  const int nd = n * n * n, nq = m * m * m;
  for (int p = t; p < m * n; p += 256) sphi[p] = phi1[p];
  const int nbatch = (ncells + CB - 1) / CB;
  for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const int c0 = batch * CB;
    const int nc = min(CB, ncells - c0);
    __syncthreads();
    int u0 = 0, nu = 0;
    if (uoff) {
      u0 = uoff[batch];
      nu = uoff[batch + 1] - u0;
      for (int u = t; u < nu; u += 256) Xu[u] = x[uniq[u0 + u]];
      __syncthreads();
      for (int p = t; p < nc * nd; p += 256) {
        const int c = p / nd, l = p - c * nd;
        A[c * mx3 + l] = Xu[loc[(size_t)c0 * nd + p]];
      }
      __syncthreads();
      for (int u = t; u < nu; u += 256) Xu[u] = 0.0;
    } else {
      for (int p = t; p < nc * nd; p += 256) {
        const int c = p / nd, l = p - c * nd;
        A[c * mx3 + l] = x[dofmap[(size_t)c0 * nd + p]];
      }
    }
    __syncthreads();
    // forward x: B[c][k][j][qi] = sum_i phi[qi][i] A[c][k][j][i]
    for (int p = t; p < nc * n * n * m; p += 256) {
      const int c = p / (n * n * m), r = p - c * (n * n * m);
      const int qi = r % m, kj = r / m;
      double s = 0.0;
      for (int a = 0; a < n; ++a) s += sphi[qi * n + a] * A[c * mx3 + kj * n + a];
      B[c * mx3 + r] = s;
    }
    __syncthreads();
    // forward y: A[c][k][qj][qi] = sum_j phi[qj][j] B[c][k][j][qi]
    for (int p = t; p < nc * n * m * m; p += 256) {
      const int c = p / (n * m * m), r = p - c * (n * m * m);
      const int qi = r % m, qj = (r / m) % m, k = r / (m * m);
      double s = 0.0;
      for (int a = 0; a < n; ++a) s += sphi[qj * n + a] * B[c * mx3 + (k * n + a) * m + qi];
      A[c * mx3 + r] = s;
    }
    __syncthreads();
    // forward z and D: B[c][qk][qj][qi] = detJ * sum_k phi[qk][k] A[c][k][qj][qi]
    for (int p = t; p < nc * nq; p += 256) {
      const int c = p / nq, r = p - c * nq;
      const int qji = r % (m * m), qk = r / (m * m);
      double s = 0.0;
      for (int a = 0; a < n; ++a) s += sphi[qk * n + a] * A[c * mx3 + a * m * m + qji];
      B[c * mx3 + r] = s * detJ[(size_t)(c0 + c) * nq + r];
    }
    __syncthreads();
    // backward z: A[c][k][qj][qi] = sum_qk phi[qk][k] B[c][qk][qj][qi]
    for (int p = t; p < nc * n * m * m; p += 256) {
      const int c = p / (n * m * m), r = p - c * (n * m * m);
      const int qji = r % (m * m), k = r / (m * m);
      double s = 0.0;
      for (int a = 0; a < m; ++a) s += sphi[a * n + k] * B[c * mx3 + a * m * m + qji];
      A[c * mx3 + r] = s;
    }
    __syncthreads();
    // backward y: B[c][k][j][qi] = sum_qj phi[qj][j] A[c][k][qj][qi]
    for (int p = t; p < nc * n * n * m; p += 256) {
      const int c = p / (n * n * m), r = p - c * (n * n * m);
      const int qi = r % m, j = (r / m) % n, k = r / (m * n);
      double s = 0.0;
      for (int a = 0; a < m; ++a) s += sphi[a * n + j] * A[c * mx3 + (k * m + a) * m + qi];
      B[c * mx3 + r] = s;
    }
    __syncthreads();
    // backward x and scatter: y[dof] += sum_qi phi[qi][i] B[c][k][j][qi]
    for (int p = t; p < nc * nd; p += 256) {
      const int c = p / nd, l = p - c * nd;
      const int i = l % n, kj = l / n;
      double s = 0.0;
      for (int a = 0; a < m; ++a) s += sphi[a * n + i] * B[c * mx3 + kj * m + a];
      if (uoff)
        atomicAdd(&Xu[loc[(size_t)c0 * nd + p]], s);
      else
        unsafeAtomicAdd(&y[dofmap[(size_t)c0 * nd + p]], s);
    }
    if (uoff) {
      __syncthreads();
      for (int u = t; u < nu; u += 256) unsafeAtomicAdd(&y[uniq[u0 + u]], Xu[u]);
    }
  }
}

// --------------------------------------------------------------------------
// dense mass, column-thread form for square tables (nq1 == P+1: the GLL-collocated
// rule of demo/gpu_operator_monolithic and the Gauss rule of degree 2P of
// demo/gpu_operator): compile-time sizes, one thread per (i, j) column, the z
// contractions in registers, x/y contractions through LDS, batch-unique
// gather/scatter.  y += Phi^T (detJ .* (Phi x)),  Phi = phi1 (x) phi1 (x) phi1.
// --------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void k_mass_dense_col(int ncells, const int32_t* __restrict__ uoff,
                                                        const int32_t* __restrict__ uniq,
                                                        const uint16_t* __restrict__ loc,
                                                        const double* __restrict__ phi1,   // [n][n]: phi1[q][a]
                                                        const double* __restrict__ detJ,   // [ncells][n^3]
                                                        const double* __restrict__ x, double* __restrict__ y)
{
  constexpr int n = P + 1, n2 = n * n, nd = n * n2;
  constexpr int CB = 256 / n2, NT = CB * n2;
  constexpr int NFLAT = (CB * nd + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* U = smem;               // [CB][nd]
  double* A = U + CB * nd;        // [CB][nd]; also the unique-dof tile before/after the contractions
  double* sP = A + CB * nd;       // [n][n]
  const int t = threadIdx.x;
  const size_t batch = blockIdx.x;
  const int cell0 = (int)batch * CB;
  const bool active = t < NT;
  const int cl = t / n2, ji = t % n2, j = ji / n, i = ji % n;
  if (t < n * n) sP[t] = phi1[t];
  const int u0 = uoff[batch], nu = uoff[batch + 1] - u0;
  for (int u = t; u < nu; u += 256) A[u] = x[uniq[u0 + u]];
  uint16_t lc[NFLAT];
  const int nvalid = min(CB, ncells - cell0) * nd;
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    lc[m] = pos < nvalid ? loc[(size_t)cell0 * nd + pos] : (uint16_t)0xFFFF;
  }
  double dj_[n];   // detJ of this thread's column (quadrature points (i, j, *)); zero for padded cells
#pragma unroll
  for (int k = 0; k < n; ++k) dj_[k] = (active && cell0 + cl < ncells) ? detJ[(size_t)(cell0 + cl) * nd + k * n2 + ji] : 0.0;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    if (pos < CB * nd) U[pos] = lc[m] != 0xFFFF ? A[lc[m]] : 0.0;
  }
  __syncthreads();
  double* Uc = U + cl * nd;
  double* Ac = A + cl * nd;
  double v[n];
  // forward x: A[k][j][qi = i] = sum_a phi[i][a] U[k][j][a]
  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) {
      double s_ = 0.0;
#pragma unroll
      for (int a = 0; a < n; ++a) s_ += sP[i * n + a] * Uc[k * n2 + j * n + a];
      v[k] = s_;
    }
#pragma unroll
    for (int k = 0; k < n; ++k) Ac[k * n2 + ji] = v[k];
  }
  __syncthreads();
  // forward y (thread = (qi = i, qj = j)), then z, detJ, backward z in registers
  if (active) {
    double vy[n];
#pragma unroll
    for (int k = 0; k < n; ++k) {
      double s_ = 0.0;
#pragma unroll
      for (int a = 0; a < n; ++a) s_ += sP[j * n + a] * Ac[k * n2 + a * n + i];
      vy[k] = s_;
    }
    double w[n];
#pragma unroll
    for (int q = 0; q < n; ++q) {
      double s_ = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) s_ += sP[q * n + k] * vy[k];
      w[q] = s_ * dj_[q];
    }
#pragma unroll
    for (int k = 0; k < n; ++k) {
      double s_ = 0.0;
#pragma unroll
      for (int q = 0; q < n; ++q) s_ += sP[q * n + k] * w[q];
      v[k] = s_;
    }
  }
  __syncthreads();   // all reads of A (forward y) done
  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) Uc[k * n2 + ji] = v[k];   // B1[k][qj][qi]
  }
  __syncthreads();
  // backward y: B2[k][j][qi = i] = sum_qj phi[qj][j] B1[k][qj][i]
  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) {
      double s_ = 0.0;
#pragma unroll
      for (int q = 0; q < n; ++q) s_ += sP[q * n + j] * Uc[k * n2 + q * n + i];
      v[k] = s_;
    }
#pragma unroll
    for (int k = 0; k < n; ++k) Ac[k * n2 + ji] = v[k];
  }
  __syncthreads();
  // backward x: out[k][j][i] = sum_qi phi[qi][i] B2[k][j][qi]
  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) {
      double s_ = 0.0;
#pragma unroll
      for (int q = 0; q < n; ++q) s_ += sP[q * n + i] * Ac[k * n2 + j * n + q];
      v[k] = s_;
    }
  }
  __syncthreads();   // all reads of A done: A becomes the unique-dof sum tile, U the element results
  if (active) {
#pragma unroll
    for (int k = 0; k < n; ++k) Uc[k * n2 + ji] = v[k];
  }
  for (int u = t; u < nu; u += 256) A[u] = 0.0;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < NFLAT; ++m) {
    const int pos = t + 256 * m;
    if (lc[m] != 0xFFFF) atomicAdd(&A[lc[m]], U[pos]);
  }
  __syncthreads();
  for (int u = t; u < nu; u += 256) unsafeAtomicAdd(&y[uniq[u0 + u]], A[u]);
}

template <int P>
static int launch_mass_dense_col_t(int ncells, const int32_t* d_uoff, const int32_t* d_uniq, const uint16_t* d_loc,
                                   const double* d_phi1, const double* d_detJ, const double* d_x, double* d_y,
                                   hipStream_t s)
{
  constexpr int n = P + 1, nd = n * n * n, CB = 256 / (n * n);
  const unsigned nb = (unsigned)((ncells + CB - 1) / CB);
  const size_t lds = (size_t)(2 * CB * nd + n * n) * sizeof(double);
  hipLaunchKernelGGL(k_mass_dense_col<P>, dim3(nb), dim3(256), lds, s, ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ,
                     d_x, d_y);
  return WF_OK;
}

int launch_mass_dense_col(int P, int ncells, const int32_t* d_uoff, const int32_t* d_uniq, const uint16_t* d_loc,
                          const double* d_phi1, const double* d_detJ, const double* d_x, double* d_y, hipStream_t s)
{
  if (ncells == 0) return WF_OK;
  int rc = WF_ERR_UNSUPPORTED;
  switch (P) {
    case 1: rc = launch_mass_dense_col_t<1>(ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ, d_x, d_y, s); break;
    case 2: rc = launch_mass_dense_col_t<2>(ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ, d_x, d_y, s); break;
    case 3: rc = launch_mass_dense_col_t<3>(ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ, d_x, d_y, s); break;
    case 4: rc = launch_mass_dense_col_t<4>(ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ, d_x, d_y, s); break;
    case 5: rc = launch_mass_dense_col_t<5>(ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ, d_x, d_y, s); break;
    case 6: rc = launch_mass_dense_col_t<6>(ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ, d_x, d_y, s); break;
    case 7: rc = launch_mass_dense_col_t<7>(ncells, d_uoff, d_uniq, d_loc, d_phi1, d_detJ, d_x, d_y, s); break;
  }
  if (rc != WF_OK) {
    set_error("mass_dense_col: degree must be 1..7");
    return rc;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("mass_dense_col launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

// --------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------
// Diagnostic ablation mask (profiling only; WF_ABLATE unset or 0 in production):
// 1 = no scatter, 2 = geometry served from L2, 4 = no x gather, 8 = no contractions.
static int ablate_flags()
{
#ifdef WF_DIAG
  const char* e = std::getenv("WF_ABLATE");
  return e ? std::atoi(e) : 0;
#else
  return 0;
#endif
}

static inline unsigned grid_for(size_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

#define WF_LAUNCH_CHECK()                                                       \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      set_error(std::string("kernel launch failed: ") + hipGetErrorString(_e)); \
      return WF_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

int launch_geometry_hex(int P, int ncells, const double* d_xverts, const int32_t* d_geom_dofmap,
                        const double* d_pts, const double* d_wts, int use_fabs, int clamp, double* d_G9,
                        double* d_G6blk, double* d_detJ, hipStream_t s)
{
  const int n = P + 1;
  const size_t N = (size_t)ncells * n * n * n;
  if (N == 0) return WF_OK;
  hipLaunchKernelGGL(k_geometry_hex, dim3(grid_for(N, 256)), dim3(256), 0, s, n, cells_per_batch(P), ncells,
                     d_xverts, d_geom_dofmap, d_pts, d_wts, use_fabs, clamp, d_G9, d_G6blk, d_detJ);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

int launch_geometry_hex_slots(int P, int CB, int ncells, const double* d_xverts, const int32_t* d_geom_dofmap,
                              const int32_t* d_slot_of, const uint8_t* d_sign, const double* d_pts, const double* d_wts,
                              int use_fabs, int clamp, double* d_G6blk, hipStream_t s)
{
  const int n = P + 1;
  const size_t N = (size_t)ncells * n * n * n;
  if (N == 0) return WF_OK;
  hipLaunchKernelGGL(k_geometry_hex_slots, dim3(grid_for(N, 256)), dim3(256), 0, s, n, CB, ncells,
                     d_xverts, d_geom_dofmap, d_slot_of, reinterpret_cast<const int8_t*>(d_sign), d_pts, d_wts, use_fabs,
                     clamp, d_G6blk);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

int launch_pack_G6(int P, int CB, int ncells, const double* d_G9, double* d_G6blk, hipStream_t s)
{
  const int n = P + 1;
  const size_t N = (size_t)ncells * n * n * n;
  if (N == 0) return WF_OK;
  hipLaunchKernelGGL(k_pack_G6, dim3(grid_for(N, 256)), dim3(256), 0, s, n, CB, ncells, d_G9,
                     d_G6blk);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

int launch_geometry_box(int P, int nx, int ny, int nz, int bx, int by, int bz, const double* d_xverts,
                        const double* d_pts, const double* d_wts, int use_fabs, int clamp, double* d_G6blk,
                        double* d_mdiag, hipStream_t s)
{
  const int n = P + 1;
  const size_t N = (size_t)nx * ny * nz * n * n * n;
  if (N == 0) return WF_OK;
  hipLaunchKernelGGL(k_geometry_box, dim3(grid_for(N, 256)), dim3(256), 0, s, n, nx, ny, nz, bx, by, bz,
                     d_xverts, d_pts, d_wts, use_fabs, clamp, d_G6blk, d_mdiag);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

template <int P>
static int launch_stiffness_generic_t(int ncells, const int32_t* d_dofmap, const double* d_G6blk,
                                      const double* d_D, const DMat& dm, double coeff, const double* d_x,
                                      double* d_y, hipStream_t s)
{
  constexpr int n = P + 1, nd = n * n * n, CB = 256 / (n * n);
  const unsigned nb = (unsigned)((ncells + CB - 1) / CB);
  const size_t lds = (size_t)(3 * CB * nd + n * n) * sizeof(double);
  hipLaunchKernelGGL(k_stiffness_generic<P>, dim3(nb), dim3(256), lds, s, ncells, d_dofmap,
                     reinterpret_cast<const double2*>(d_G6blk), d_D, dm, coeff, d_x, d_y, ablate_flags());
  WF_LAUNCH_CHECK();
  return WF_OK;
}

#ifndef WF_GENERIC_PIPELINED
#define WF_GENERIC_PIPELINED 1
#endif
#ifndef WF_GENERIC_UP2_MAXP
#define WF_GENERIC_UP2_MAXP 4   // degrees that use the fully pipelined kernel (two geometry register sets; P5, P6 need 256 VGPRs + AGPRs: one wave per SIMD)
#endif
template <int P>
static int launch_stiffness_generic_u_t(int ncells, const int32_t* d_uoff, const int32_t* d_uniq,
                                        const uint16_t* d_loc, const double* d_G6blk, const double* d_D,
                                        const DMat& dm, double coeff, const double* d_x, double* d_y, hipStream_t s)
{
  constexpr int n = P + 1, nd = n * n * n, CB = 256 / (n * n);
  const unsigned nb = (unsigned)((ncells + CB - 1) / CB);
  const size_t lds = (size_t)(3 * CB * nd + n * n) * sizeof(double);
#if WF_GENERIC_PIPELINED
  if (ablate_flags() == 0) {
    // persistent workgroups, as many as fit the chip at this kernel's register / LDS use (4 per CU at P <= 4)
    int per_cu = 0;
    if constexpr (P <= WF_GENERIC_UP2_MAXP) {
      WF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stiffness_generic_up2<P>, 256, lds));
      const unsigned grid = std::min<unsigned>(nb, (unsigned)std::max(1, per_cu) * 256u);
      hipLaunchKernelGGL(k_stiffness_generic_up2<P>, dim3(grid), dim3(256), lds, s, ncells, (int)nb, d_uoff, d_uniq, d_loc,
                         reinterpret_cast<const double2*>(d_G6blk), d_D, dm, coeff, d_x, d_y);
    } else {
      WF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_stiffness_generic_up<P>, 256, lds));
      const unsigned grid = std::min<unsigned>(nb, (unsigned)std::max(1, per_cu) * 256u);
      hipLaunchKernelGGL(k_stiffness_generic_up<P>, dim3(grid), dim3(256), lds, s, ncells, (int)nb, d_uoff, d_uniq, d_loc,
                         reinterpret_cast<const double2*>(d_G6blk), d_D, dm, coeff, d_x, d_y);
    }
    WF_LAUNCH_CHECK();
    return WF_OK;
  }
#endif
  hipLaunchKernelGGL(k_stiffness_generic_u<P>, dim3(nb), dim3(256), lds, s, ncells, d_uoff, d_uniq, d_loc,
                     reinterpret_cast<const double2*>(d_G6blk), d_D, dm, coeff, d_x, d_y, ablate_flags());
  WF_LAUNCH_CHECK();
  return WF_OK;
}

int launch_stiffness_generic_u(int P, int ncells, const int32_t* d_uoff, const int32_t* d_uniq, const uint16_t* d_loc,
                               const double* d_G6blk, const double* d_D, const DMat& dm, double coeff,
                               const double* d_x, double* d_y, hipStream_t s)
{
  if (ncells == 0) return WF_OK;
  switch (P) {
    case 1: return launch_stiffness_generic_u_t<1>(ncells, d_uoff, d_uniq, d_loc, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 2: return launch_stiffness_generic_u_t<2>(ncells, d_uoff, d_uniq, d_loc, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 3: return launch_stiffness_generic_u_t<3>(ncells, d_uoff, d_uniq, d_loc, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 4: return launch_stiffness_generic_u_t<4>(ncells, d_uoff, d_uniq, d_loc, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 5: return launch_stiffness_generic_u_t<5>(ncells, d_uoff, d_uniq, d_loc, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 6: return launch_stiffness_generic_u_t<6>(ncells, d_uoff, d_uniq, d_loc, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 7: return launch_stiffness_generic_u_t<7>(ncells, d_uoff, d_uniq, d_loc, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
  }
  set_error("stiffness: degree must be 1..7");
  return WF_ERR_UNSUPPORTED;
}

int launch_stiffness_generic(int P, int ncells, const int32_t* d_dofmap, const double* d_G6blk,
                             const double* d_D, const DMat& dm, double coeff, const double* d_x, double* d_y,
                             hipStream_t s)
{
  if (ncells == 0) return WF_OK;
  switch (P) {
    case 1: return launch_stiffness_generic_t<1>(ncells, d_dofmap, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 2: return launch_stiffness_generic_t<2>(ncells, d_dofmap, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 3: return launch_stiffness_generic_t<3>(ncells, d_dofmap, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 4: return launch_stiffness_generic_t<4>(ncells, d_dofmap, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 5: return launch_stiffness_generic_t<5>(ncells, d_dofmap, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 6: return launch_stiffness_generic_t<6>(ncells, d_dofmap, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 7: return launch_stiffness_generic_t<7>(ncells, d_dofmap, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
  }
  set_error("stiffness: degree must be 1..7");
  return WF_ERR_UNSUPPORTED;
}

template <int P>
static int launch_stiffness_box_t(int nx, int ny, int nz, int bx, int by, int bz, const double* d_G6blk,
                                  const double* d_D, const DMat& dm, double coeff, const double* d_x,
                                  double* d_y, hipStream_t s)
{
  constexpr int n = P + 1, nd = n * n * n;
  const int CB = bx * by * bz;
  const int tile = (P * bx + 1) * (P * by + 1) * (P * bz + 1);
  const unsigned nb = (unsigned)(((nx + bx - 1) / bx) * ((ny + by - 1) / by) * ((nz + bz - 1) / bz));
  const size_t lds = (size_t)(((tile + 1) & ~1) + 2 * CB * nd + n * n) * sizeof(double);
  hipLaunchKernelGGL(k_stiffness_box<P>, dim3(nb), dim3(256), lds, s, nx, ny, nz, bx, by, bz,
                     reinterpret_cast<const double2*>(d_G6blk), d_D, dm, coeff, d_x, d_y, ablate_flags());
  WF_LAUNCH_CHECK();
  return WF_OK;
}

int launch_stiffness_box(int P, int nx, int ny, int nz, int bx, int by, int bz, const double* d_G6blk,
                         const double* d_D, const DMat& dm, double coeff, const double* d_x, double* d_y,
                         hipStream_t s)
{
  if ((size_t)nx * ny * nz == 0) return WF_OK;
  if (bx * by * bz * (P + 1) * (P + 1) > 256) {
    set_error("stiffness_box: block does not fit a 256-thread workgroup");
    return WF_ERR_INVALID;
  }
  switch (P) {
    case 1: return launch_stiffness_box_t<1>(nx, ny, nz, bx, by, bz, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 2: return launch_stiffness_box_t<2>(nx, ny, nz, bx, by, bz, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 3: return launch_stiffness_box_t<3>(nx, ny, nz, bx, by, bz, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 4: return launch_stiffness_box_t<4>(nx, ny, nz, bx, by, bz, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 5: return launch_stiffness_box_t<5>(nx, ny, nz, bx, by, bz, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 6: return launch_stiffness_box_t<6>(nx, ny, nz, bx, by, bz, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
    case 7: return launch_stiffness_box_t<7>(nx, ny, nz, bx, by, bz, d_G6blk, d_D, dm, coeff, d_x, d_y, s);
  }
  set_error("stiffness_box: degree must be 1..7");
  return WF_ERR_UNSUPPORTED;
}

int launch_mass_lumped(int64_t nentries, const int32_t* d_dofmap, const double* d_detJ, const double* d_x,
                       double* d_y, hipStream_t s)
{
  if (nentries == 0) return WF_OK;
  hipLaunchKernelGGL(k_mass_lumped, dim3(grid_for((size_t)nentries, 256)), dim3(256), 0, s, nentries, d_dofmap,
                     d_detJ, d_x, d_y);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

int launch_mass_lumped_u(int ncells, int nd, int CB, const int32_t* d_uoff, const int32_t* d_uniq,
                         const uint16_t* d_loc, const double* d_detJ, const double* d_x, double* d_y, hipStream_t s)
{
  if (ncells == 0) return WF_OK;
  const int nbatch = (ncells + CB - 1) / CB;
  const size_t lds = (size_t)2 * CB * nd * sizeof(double);
  hipLaunchKernelGGL(k_mass_lumped_u, dim3((unsigned)std::min(nbatch, 256 * 8)), dim3(256), lds, s, ncells, nd, CB,
                     d_uoff, d_uniq, d_loc, d_detJ, d_x, d_y);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

// cells per workgroup of k_mass_dense: enough that every contraction pass fills 256
// lanes (measured: P2 >= 8, P4 2..8, P6 4)
int mass_dense_cells_per_batch(int mx)
{
  return std::max(1, std::min(1400 / (mx * mx * mx), 32));
}

int launch_mass_dense(int P, int nq1, int ncells, const int32_t* d_dofmap, const int32_t* d_uoff,
                      const int32_t* d_uniq, const uint16_t* d_loc, int CBu, const double* d_phi1,
                      const double* d_detJ, const double* d_x, double* d_y, hipStream_t s)
{
  if (ncells == 0) return WF_OK;
  const int n = P + 1, mx = n > nq1 ? n : nq1, mx3 = mx * mx * mx;
  // cells per workgroup: enough that every contraction pass fills 256 lanes, within ~32 KB of LDS
  const int CB = d_uoff ? CBu : mass_dense_cells_per_batch(mx);   // the unique lists fix the batch size
  const size_t lds = (size_t)(2 * CB * mx3 + nq1 * n + (d_uoff ? CB * n * n * n : 0)) * sizeof(double);
  if (lds > 160 * 1024) {
    set_error("mass_dense: tables do not fit LDS");
    return WF_ERR_UNSUPPORTED;
  }
  const unsigned nb = (unsigned)std::min<int64_t>((ncells + CB - 1) / CB, 256 * 8);
  if (lds > 64 * 1024)
    WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mass_dense), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds));
  hipLaunchKernelGGL(k_mass_dense, dim3(nb), dim3(256), lds, s, n, nq1, CB, ncells, d_dofmap, d_uoff, d_uniq, d_loc,
                     d_phi1, d_detJ, d_x, d_y);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

}  // namespace wf
