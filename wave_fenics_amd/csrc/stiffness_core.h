// Per-thread core of the sum-factorised hexahedral stiffness operator, shared by
// the generic, box and marching kernels.
#pragma once
#include "common.h"

namespace wf {

// Non-temporal 16-byte load for data that is read exactly once per apply (the geometry
// stream, 80 % of the operator's bytes): it does not displace x and y from the XCD's L2 and
// the Infinity Cache.  Measured at cfg2 (P4, 10.2 M dofs): box apply 0.254 -> 0.230 ms
// standalone; inside the RK4 loop, where nine vectors compete for the caches, 0.280 -> 0.235 ms.
__device__ __forceinline__ double2 load_stream(const double2* p)
{
  typedef double d2v __attribute__((ext_vector_type(2)));
  const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p));
  return make_double2(v.x, v.y);
}

// --------------------------------------------------------------------------
// stiffness: per-thread core shared by the generic, box and marching kernels
// --------------------------------------------------------------------------
// U: LDS dofs of this thread's cell, addressed U[k*sk + j*sj + i] (strides in
// doubles; the generic kernel uses the compact cell layout sk = n^2, sj = n,
// the box kernel addresses the cell inside the block's dof tile).
// Fr, Fs: LDS scratch of the cell, compact layout.  sD: LDS copy of D.
// Phase 1: reference gradient at the quadrature points, times G -> Fr, Fs (LDS) and ft
// (registers).  A workgroup barrier separates it from phase 2, which applies D^T:
// out[k] = (K_cell u)[i, j, k].  The geometry registers g are dead after phase 1.
// P7: the z rows of D come from LDS (broadcast reads) -- as 128 SGPRs they spill (224 v_readlane per layer in the
// batch kernel: 0.189 -> 0.1755 ms with LDS rows; P6: 0.201 -> 0.207, P5 unchanged, so P <= 6 keeps the SGPRs)
#ifndef WF_CORE_DZ_LDS
#define WF_CORE_DZ_LDS 1
#endif
template <int P>
__device__ __forceinline__ void stiffness_phase1(const double* __restrict__ U, int sk, int sj,
                                                 double* __restrict__ Fr, double* __restrict__ Fs,
                                                 const double* __restrict__ sD, const DMat& dm,
                                                 const double2 (&g)[P + 1][3], double coeff, int i, int j,
                                                 bool active, double (&ft)[P + 1])
{
  constexpr int n = P + 1, n2 = n * n;
  if (active) {
    double ru[n];
#pragma unroll
    for (int k = 0; k < n; ++k) ru[k] = U[k * sk + j * sj + i];
    double di[n], dj[n];
#pragma unroll
    for (int a = 0; a < n; ++a) {
      di[a] = sD[i * n + a];
      dj[a] = sD[j * n + a];
    }
#pragma unroll
    for (int k = 0; k < n; ++k) {
      double ur = 0.0, us = 0.0, ut = 0.0;
#pragma unroll
      for (int a = 0; a < n; ++a) {
        ur += di[a] * U[k * sk + j * sj + a];
        us += dj[a] * U[k * sk + a * sj + i];
        ut += (WF_CORE_DZ_LDS && P >= 7 ? sD[k * n + a] : dm.v[k * n + a]) * ru[a];
      }
      const double g00 = g[k][0].x, g01 = g[k][0].y, g02 = g[k][1].x, g11 = g[k][1].y,
                   g12 = g[k][2].x, g22 = g[k][2].y;
      // operators.hpp:126-128: fw = coeff * (G row . w)
      const double fr = coeff * (g00 * ur + g01 * us + g02 * ut);
      const double fs = coeff * (g01 * ur + g11 * us + g12 * ut);
      ft[k] = coeff * (g02 * ur + g12 * us + g22 * ut);
      Fr[k * n2 + j * n + i] = fr;
      Fs[k * n2 + j * n + i] = fs;
    }
  }
}

template <int P>
__device__ __forceinline__ void stiffness_phase2(const double* __restrict__ Fr, const double* __restrict__ Fs,
                                                 const double* __restrict__ sD, const DMat& dm,
                                                 const double (&ft)[P + 1], int i, int j, bool active,
                                                 double (&out)[P + 1])
{
  constexpr int n = P + 1, n2 = n * n;
  if (active) {
    double dti[n], dtj[n];
#pragma unroll
    for (int a = 0; a < n; ++a) {
      dti[a] = sD[a * n + i];
      dtj[a] = sD[a * n + j];
    }
#pragma unroll
    for (int k = 0; k < n; ++k) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < n; ++a) {
        s += dti[a] * Fr[k * n2 + j * n + a];
        s += dtj[a] * Fs[k * n2 + a * n + i];
        s += (WF_CORE_DZ_LDS && P >= 7 ? sD[a * n + k] : dm.v[a * n + k]) * ft[a];
      }
      out[k] = s;
    }
  }
}

// Both phases with the barrier between them.  Output: out[k] = (K_cell u)[i, j, k].
template <int P>
__device__ __forceinline__ void stiffness_column(const double* __restrict__ U, int sk, int sj,
                                                 double* __restrict__ Fr, double* __restrict__ Fs,
                                                 const double* __restrict__ sD, const DMat& dm,
                                                 const double2 (&g)[P + 1][3], double coeff, int i,
                                                 int j, bool active, double (&out)[P + 1], int ablate = 0)
{
  constexpr int n = P + 1;
  double ft[n];
  if (ablate & 8) {   // diagnostic: no contractions, keep every input live
    if (active) {
#pragma unroll
      for (int k = 0; k < n; ++k)
        out[k] = U[k * sk + j * sj + i] + g[k][0].x + g[k][0].y + g[k][1].x + g[k][1].y + g[k][2].x + g[k][2].y;
    }
    __syncthreads();
    return;
  }
  stiffness_phase1<P>(U, sk, sj, Fr, Fs, sD, dm, g, coeff, i, j, active, ft);
  __syncthreads();
  stiffness_phase2<P>(Fr, Fs, sD, dm, ft, i, j, active, out);
}

}  // namespace wf
