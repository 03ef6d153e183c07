/*
 * wave_oracle.c -- CPU restatement of the wave-fenics operator hot loops.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the
 * "cpu_baseline" of bench.py.  Nothing in the product path (wave_fenics_amd/)
 * may import, link or execute it.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or
 * fixtures for this path, and it cannot be built here
 * (needs DOLFINx + Basix + xtensor + FFCx).  The functions below restate the
 * reference loops line by line; they are checked against analytic known-answer
 * tests written for this repo (tests/test_oracle_kat.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).
 *
 * Build: see oracle/Makefile.  Two builds of the same source are produced:
 *   libwave_oracle.so       -O2 (strict IEEE; the parity checker)
 *   libwave_oracle_fast.so  the reference's own flags
 *                           (-Ofast -march=native -mprefer-vector-width=512,
 *                            demo/cpu_planar3d/CMakeLists.txt:27; timing only)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* common/operators.hpp:113-133  skernel<T>
 * A[i] += sum_q (coeff * G_q * (dphi_q . w)) . dphi_q[i], coeff = -c0^2.
 * dphi is the dense table [3][nq][nd] (row-major), G is [nq][3][3] row-major.
 * The reference hard-codes c0 = 1500 (operators.hpp:114); it is a parameter
 * here and every caller passes 1500 unless a test says otherwise. */
static void skernel(double* A, const double* w, double c0, const double* G,
                    const double* dphi, int nq, int nd)
{
  const double coeff = -1.0 * c0 * c0;
  const double* d0 = dphi;
  const double* d1 = dphi + (size_t)nq * nd;
  const double* d2 = dphi + (size_t)2 * nq * nd;
  for (int iq = 0; iq < nq; iq++) {
    const double* _G = G + iq * 9;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
    for (int ic = 0; ic < nd; ic++) {
      w0 += w[ic] * d0[(size_t)iq * nd + ic];
      w1 += w[ic] * d1[(size_t)iq * nd + ic];
      w2 += w[ic] * d2[(size_t)iq * nd + ic];
    }
    const double fw0 = coeff * (_G[0] * w0 + _G[1] * w1 + _G[2] * w2);
    const double fw1 = coeff * (_G[3] * w0 + _G[4] * w1 + _G[5] * w2);
    const double fw2 = coeff * (_G[6] * w0 + _G[7] * w1 + _G[8] * w2);
    for (int i = 0; i < nd; i++) {
      A[i] += fw0 * d0[(size_t)iq * nd + i] + fw1 * d1[(size_t)iq * nd + i]
              + fw2 * d2[(size_t)iq * nd + i];
    }
  }
}

/* common/operators.hpp:183-200  StiffnessOperator::operator()
 * y += K x over cells [cell_begin, cell_end); y is accumulated, never zeroed.
 * dofmap is [ncells][nd] int32, G is [ncells][nq][3][3]. */
void oracle_stiffness_apply(int cell_begin, int cell_end, int nd, int nq,
                            const int32_t* dofmap, const double* G,
                            const double* dphi, double c0, const double* x,
                            double* y)
{
  double* _x = (double*)malloc(sizeof(double) * nd);
  double* _y = (double*)malloc(sizeof(double) * nd);
  for (int cell = cell_begin; cell < cell_end; ++cell) {
    const int32_t* cell_dofs = dofmap + (size_t)cell * nd;
    for (int i = 0; i < nd; i++)
      _x[i] = x[cell_dofs[i]];
    memset(_y, 0, sizeof(double) * nd);
    const double* G_cell = G + (size_t)cell * nq * 9;
    skernel(_y, _x, c0, G_cell, dphi, nq, nd);
    for (int i = 0; i < nd; i++)
      y[cell_dofs[i]] += _y[i];
  }
  free(_x);
  free(_y);
}

/* common/operators.hpp:36-40 mkernel + :86-108 MassOperatorCPU::operator()
 * Lumped (GLL-collocated) mass: _x[i] = x[dofs[perm[i]]]; _y[q] = _x[q]*detJ[c][q];
 * y[dofs[perm[i]]] += _y[i].  Requires nq == nd. */
void oracle_mass_apply(int cell_begin, int cell_end, int nd, int nq,
                       const int32_t* dofmap, const int32_t* perm,
                       const double* detJ, const double* x, double* y)
{
  double* _x = (double*)malloc(sizeof(double) * nd);
  double* _y = (double*)malloc(sizeof(double) * nd);
  for (int cell = cell_begin; cell < cell_end; ++cell) {
    const int32_t* cell_dofs = dofmap + (size_t)cell * nd;
    for (int i = 0; i < nd; i++)
      _x[i] = x[cell_dofs[perm[i]]];
    memset(_y, 0, sizeof(double) * nd);
    const double* detJ_ptr = detJ + (size_t)cell * nq;
    for (int iq = 0; iq < nq; ++iq)
      _y[iq] = _x[iq] * detJ_ptr[iq];
    for (int i = 0; i < nd; i++)
      y[cell_dofs[perm[i]]] += _y[i];
  }
  free(_x);
  free(_y);
}

/* common/cuda/mass_kernel.cu:5-37 _mass_apply + common/cuda/mass.hpp:76-95
 * Dense mass action y += Phi^T (detJ .* (Phi x_e)) with Phi [nq][nd] row-major.
 * Also the cuBLAS path of demo/gpu_operator/main.cpp:144-160 (B, D, B^T). */
void oracle_dense_mass_apply(int cell_begin, int cell_end, int nd, int nq,
                             const int32_t* dofmap, const double* phi,
                             const double* detJ, const double* x, double* y)
{
  double* xe = (double*)malloc(sizeof(double) * nd);
  double* xq = (double*)malloc(sizeof(double) * nq);
  for (int cell = cell_begin; cell < cell_end; ++cell) {
    const int32_t* cell_dofs = dofmap + (size_t)cell * nd;
    for (int i = 0; i < nd; i++)
      xe[i] = x[cell_dofs[i]];
    for (int q = 0; q < nq; q++) {
      double wq = 0.0;
      for (int j = 0; j < nd; j++)
        wq += xe[j] * phi[(size_t)q * nd + j];
      xq[q] = detJ[(size_t)cell * nq + q] * wq;
    }
    for (int i = 0; i < nd; i++) {
      double yi = 0.0;
      for (int q = 0; q < nq; q++)
        yi += xq[q] * phi[(size_t)q * nd + i];
      y[cell_dofs[i]] += yi;
    }
  }
  free(xe);
  free(xq);
}

/* 3x3 determinant and inverse as dolfinx::math::det / math::inv
 * (called from common/precomputation.hpp:95-96). */
static double det3(const double* A)
{
  return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6])
         + A[2] * (A[3] * A[7] - A[4] * A[6]);
}
static void inv3(const double* A, double* B)
{
  const double idet = 1.0 / det3(A);
  B[0] = (A[4] * A[8] - A[5] * A[7]) * idet;
  B[1] = (A[2] * A[7] - A[1] * A[8]) * idet;
  B[2] = (A[1] * A[5] - A[2] * A[4]) * idet;
  B[3] = (A[5] * A[6] - A[3] * A[8]) * idet;
  B[4] = (A[0] * A[8] - A[2] * A[6]) * idet;
  B[5] = (A[2] * A[3] - A[0] * A[5]) * idet;
  B[6] = (A[3] * A[7] - A[4] * A[6]) * idet;
  B[7] = (A[1] * A[6] - A[0] * A[7]) * idet;
  B[8] = (A[0] * A[4] - A[1] * A[3]) * idet;
}

static inline double clamp101(double v)
{
  /* xt::isclose(a, b): |a-b| <= atol + rtol*|b|, rtol 1e-5, atol 1e-8
   * (common/precomputation.hpp:105-107; applied in the order -1, 0, 1). */
  if (fabs(v + 1.0) <= 1e-8 + 1e-5) v = -1.0;
  if (fabs(v) <= 1e-8) v = 0.0;
  if (fabs(v - 1.0) <= 1e-8 + 1e-5) v = 1.0;
  return v;
}

/* common/precomputation.hpp:69-107  precompute_geometric_data, cell loop.
 * xv [nverts][3], geom_dofmap [ncells][nnodes], dphi_cmap [3][nq][nnodes]
 * (already clamped by the caller as in :55-58), weights [nq].
 * Out: G [ncells][nq][3][3], detJ [ncells][nq].
 * If use_fabs == 0 the determinant keeps its sign (the generic path,
 * common/precompute.hpp:102-116 used by spectral_mass.hpp:58-64). */
void oracle_geometry(int ncells, int nq, int nnodes, const double* xv,
                     const int32_t* geom_dofmap, const double* dphi_cmap,
                     const double* weights, int use_fabs, int do_clamp,
                     double* G, double* detJ)
{
  for (int c = 0; c < ncells; c++) {
    const int32_t* x_dofs = geom_dofmap + (size_t)c * nnodes;
    for (int q = 0; q < nq; q++) {
      double J[9], Ji[9];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double s = 0.0;
          const double* dp = dphi_cmap + ((size_t)j * nq + q) * nnodes;
          for (int n = 0; n < nnodes; n++)
            s += xv[(size_t)x_dofs[n] * 3 + i] * dp[n];
          J[i * 3 + j] = s;
        }
      double d = det3(J);
      if (use_fabs) d = fabs(d);
      d *= weights[q];
      detJ[(size_t)c * nq + q] = d;
      if (G) {
        inv3(J, Ji);
        double* g = G + ((size_t)c * nq + q) * 9;
        /* dot(J_inv * detJ, transpose(J_inv)) : precomputation.hpp:99-100 */
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++)
              s += (Ji[i * 3 + k] * d) * Ji[j * 3 + k];
            g[i * 3 + j] = do_clamp ? clamp101(s) : s;
          }
      }
    }
  }
}

/* Sum-factorised CPU variant of the same stiffness operator ("Baseline B" of
 * BASELINE.md section 3): informative only, never substituted for the dense
 * reference form.  D is the 1-D collocation derivative matrix [n][n]
 * (D[q][a] = l_a'(xi_q)); local index l = i + n*(j + n*k). */
void oracle_stiffness_apply_sumfact(int cell_begin, int cell_end, int n,
                                    const int32_t* dofmap, const double* G,
                                    const double* D, double c0, const double* x,
                                    double* y)
{
  const int nd = n * n * n;
  const double coeff = -1.0 * c0 * c0;
  double* u = (double*)malloc(sizeof(double) * nd * 4);
  double* f0 = u + nd;
  double* f1 = f0 + nd;
  double* f2 = f1 + nd;
  for (int cell = cell_begin; cell < cell_end; ++cell) {
    const int32_t* cd = dofmap + (size_t)cell * nd;
    for (int i = 0; i < nd; i++) u[i] = x[cd[i]];
    const double* Gc = G + (size_t)cell * nd * 9;
    for (int k = 0; k < n; k++)
      for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
          double w0 = 0, w1 = 0, w2 = 0;
          for (int a = 0; a < n; a++) {
            w0 += D[i * n + a] * u[a + n * (j + n * k)];
            w1 += D[j * n + a] * u[i + n * (a + n * k)];
            w2 += D[k * n + a] * u[i + n * (j + n * a)];
          }
          const int q = i + n * (j + n * k);
          const double* g = Gc + q * 9;
          f0[q] = coeff * (g[0] * w0 + g[1] * w1 + g[2] * w2);
          f1[q] = coeff * (g[3] * w0 + g[4] * w1 + g[5] * w2);
          f2[q] = coeff * (g[6] * w0 + g[7] * w1 + g[8] * w2);
        }
    for (int k = 0; k < n; k++)
      for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
          double s = 0;
          for (int a = 0; a < n; a++) {
            s += D[a * n + i] * f0[a + n * (j + n * k)];
            s += D[a * n + j] * f1[i + n * (a + n * k)];
            s += D[a * n + k] * f2[i + n * (j + n * a)];
          }
          y[cd[i + n * (j + n * k)]] += s;
        }
  }
  free(u);
}
