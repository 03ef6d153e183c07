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
// NU: unique dofs of a batch per thread (numax <= NU * 64 NW), a compile-time bound for the
// register-staged gather of the NEXT batch.
template <int QT, int KT, int DT, int NW, int NU>
__global__ __launch_bounds__(64 * NW) void k_stiffness_dense(int nd, int nq, int nbatch, int numax,
                                                         const double* __restrict__ Tg,      // [3*16*QT][KP] padded table
                                                         const double* __restrict__ wq,      // [16*QT] weights (0 beyond nq)
                                                         const double* __restrict__ Cg,      // [nbatch*16*NW][6]
                                                         const uint16_t* __restrict__ locT,  // [nbatch][4*KT][16*NW]
                                                         const int32_t* __restrict__ uoff,   // [nbatch+1]
                                                         const int32_t* __restrict__ uniq,   // unique dofs of all batches
                                                         const uint8_t* __restrict__ clampb, // [nbatch] 1: some w_q C_c of the batch lies in a clamp window
                                                         double coeff, int do_clamp, const double* __restrict__ x,
                                                         double* __restrict__ y, int ablate)
{
  constexpr int NQP = 16 * QT, KP = 4 * KT + 1, NT = 64 * NW, NCB = 16 * NW;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* T = smem;                 // [3*NQP][KP]
  double* sw = T + 3 * NQP * KP;    // [NQP]
  double* Xu = sw + NQP;            // [numax]  x at the unique dofs of the batch
  double* Yu = Xu + numax;          // [numax]  sum of the cell results per unique dof

  (void)nq;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lc = lane & 15, lg = lane >> 4;
  for (int p = t; p < 3 * NQP * KP; p += NT) T[p] = Tg[p];
  for (int p = t; p < NQP; p += NT) sw[p] = wq[p];

  // Software pipeline over the batches of this (persistent) workgroup: while batch b is in its
  // MFMA section, the gather of batch b + G is in flight in registers (x values xr, local
  // indices and geometry) and the unique-dof list of batch b + 2G is being fetched (uq), so the
  // only memory latency left on the critical path is the scatter's.  Inside the MFMA section no
  // global load is consumed (loads retire in order).
  const int G = gridDim.x;
  auto load_uq = [&](int32_t (&uq)[NU], int b) {
    const int u0 = b < nbatch ? uoff[b] : 0, nu = b < nbatch ? uoff[b + 1] - u0 : 0;
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      const int u = t + NT * m;
      uq[m] = u < nu ? uniq[u0 + u] : -1;
    }
  };
  auto load_x = [&](double (&xr)[NU], const int32_t (&uq)[NU]) {
#pragma unroll
    for (int m = 0; m < NU; ++m) xr[m] = uq[m] >= 0 ? ((ablate & 4) ? 1.0 + m : x[uq[m]]) : 0.0;
  };
  auto load_cell = [&](uint16_t (&loc)[KT], double (&C)[6], int b) {
    if (b >= nbatch) return;
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) loc[ks] = locT[((size_t)b * 4 * KT + 4 * ks + lg) * NCB + wave * 16 + lc];
    const double* cp = Cg + ((size_t)b * NCB + wave * 16 + lc) * 6;
#pragma unroll
    for (int e = 0; e < 6; ++e) C[e] = cp[e];
  };
  int32_t uq_cur[NU], uq_nxt[NU];     // unique dofs (this thread's share) of the batch being computed / of the next one
  double xr[NU];
  uint16_t loc[KT], locn[KT];
  double C[6], Cn[6];
  // prologue: first batch straight into LDS, indices of the second
  load_uq(uq_cur, blockIdx.x);
  load_x(xr, uq_cur);
  load_cell(loc, C, blockIdx.x);
  load_uq(uq_nxt, blockIdx.x + G);
#pragma unroll
  for (int m = 0; m < NU; ++m) {
    const int u = t + NT * m;
    if (u < numax) {
      Xu[u] = xr[m];
      Yu[u] = 0.0;
    }
  }

  for (int batch = blockIdx.x; batch < nbatch; batch += G) {
    // the -1/0/1 clamp of G (precomputation.hpp:105-107) is the identity unless a product w_q C_c
    // falls into one of its windows; the host marks the batches where that happens (same
    // double-precision products), every other batch skips ~200 VALU instructions per slab
    const bool clamp_here = do_clamp && clampb[batch];
    __syncthreads();   // Xu holds this batch's x values, Yu is zero

    // B operands of the first product: this lane's dof values, one per k-step
    double ub[KT];
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) ub[ks] = (4 * ks + lg) < nd ? Xu[loc[ks]] : 0.0;
    // gather of the next batch (registers) and index list of the one after it
    int32_t uq_nn[NU];
    load_x(xr, uq_nxt);
    load_cell(locn, Cn, batch + G);
    load_uq(uq_nn, batch + 2 * G);

    double4_t Y[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) Y[dt] = double4_t{0.0, 0.0, 0.0, 0.0};

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
      const double* Ta = T + (16 * qt + lc) * KP + lg;   // + e NQP KP + 4 ks
      double ac[3], an[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) ac[e] = Ta[e * NQP * KP];
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        if (ks + 1 < KT) {
#pragma unroll
          for (int e = 0; e < 3; ++e) an[e] = Ta[e * NQP * KP + 4 * (ks + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them otherwise)
#pragma unroll
        for (int e = 0; e < 3; ++e) W[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[e], ub[ks], W[e], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 3; ++e) ac[e] = an[e];
      }
      // first A operands of the second product: in flight during the lane-local geometry product
      // (row = e NQP + 16 qt + 4 r + lg, column d = 16 dt + lc; columns >= 4 KT are never stored: pad = 0)
      const double* Tb = T + (16 * qt + lg) * KP + lc;   // + (e NQP + 4 r) KP + 16 dt
      double bc[DT], bn[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) bc[dt] = (16 * dt + lc) < 4 * KT ? Tb[16 * dt] : 0.0;
      // ---- F = coeff * G W (lane-local: the three directions of one (q, cell) share lane and register)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + lg + 4 * r;
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
        if (st + 1 < 12) {
          const int en = (st + 1) / 4, rn = (st + 1) % 4;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) bn[dt] = (16 * dt + lc) < 4 * KT ? Tb[(en * NQP + 4 * rn) * KP + 16 * dt] : 0.0;
        }
        const double b = W[e][r];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) Y[dt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bc[dt], b, Y[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) bc[dt] = bn[dt];
      }
    }
    // ---- per-batch accumulation over unique dofs, then one atomic per unique dof
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ks = 4 * dt + r;   // d = 16 dt + lg + 4 r = 4 ks + lg
        if (ks < KT && 4 * ks + lg < nd) atomicAdd(&Yu[loc[ks < KT ? ks : 0]], Y[dt][r]);
      }
    __syncthreads();   // Yu complete; every wave has taken its operands out of Xu
#pragma unroll
    for (int m = 0; m < NU; ++m) {
      const int u = t + NT * m;
      if (uq_cur[m] >= 0) {
        if (ablate & 1) {
          if (Yu[u] == 1.2345e300) y[uq_cur[m]] = Yu[u];
        } else {
          unsafeAtomicAdd(&y[uq_cur[m]], Yu[u]);
        }
      }
      if (u < numax) {   // same thread, same entries: next batch's x values in, sums back to zero
        Xu[u] = xr[m];
        Yu[u] = 0.0;
      }
      uq_cur[m] = uq_nxt[m];
      uq_nxt[m] = uq_nn[m];
    }
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) loc[ks] = locn[ks];
#pragma unroll
    for (int e = 0; e < 6; ++e) C[e] = Cn[e];
  }
}

struct DenseOpData {
  int nd = 0, nq = 0, QT = 0, KT = 0, DT = 0, nbatch = 0, numax = 0, nw = 4;
  double* d_T = nullptr;
  double* d_w = nullptr;
  double* d_C = nullptr;
  uint16_t* d_locT = nullptr;
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
  (void)hipFree(d->d_locT);
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
  const int NQP = 16 * d->QT, KP = 4 * d->KT + 1;
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
  std::vector<uint16_t> locT((size_t)nbatch * 4 * d->KT * NCB, 0);
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
        locT[((size_t)b * 4 * d->KT + k) * NCB + c] = (uint16_t)u;
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
  if ((rc = up(&d->d_locT, locT, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_uoff, uoff, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_uniq, uniq, &d->bytes)) != WF_OK) return rc;
  if ((rc = up(&d->d_clampb, clampb, &d->bytes)) != WF_OK) return rc;
  *out = d.release();
  return WF_OK;
}

size_t dense_bytes(const DenseOpData* d) { return d ? d->bytes : 0; }

template <int QT, int KT, int DT, int NW, int NU>
static int launch_dense_t(const DenseOpData* d, double coeff, int do_clamp, const double* d_x, double* d_y,
                          hipStream_t s)
{
  constexpr int NQP = 16 * QT, KP = 4 * KT + 1;
  const size_t lds = ((size_t)3 * NQP * KP + NQP + 2 * d->numax) * sizeof(double);
  if (lds > 160 * 1024) {
    set_error("stiffness_dense: tables do not fit LDS");
    return WF_ERR_UNSUPPORTED;
  }
  auto kern = k_stiffness_dense<QT, KT, DT, NW, NU>;
  if (lds > 64 * 1024)
    WF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
  const unsigned nb = (unsigned)std::min(d->nbatch, 256 * 2);   // persistent: the table is staged into LDS once per workgroup
  hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * NW), lds, s, d->nd, d->nq, d->nbatch, d->numax, d->d_T, d->d_w, d->d_C,
                     d->d_locT, d->d_uoff, d->d_uniq, d->d_clampb, coeff, do_clamp, d_x, d_y,
                     std::getenv("WF_ABLATE") ? std::atoi(std::getenv("WF_ABLATE")) : 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("stiffness_dense launch failed: ") + hipGetErrorString(e));
    return WF_ERR_HIP;
  }
  return WF_OK;
}

// NU = 5 covers the unique dofs of 64 well-numbered P4 cells (1154 on the Kuhn box); 9 is the worst case 64 * 35
#define WF_DENSE_CASE(Q, K, D)                                                                                   \
  if (d->QT == Q && d->KT == K && d->DT == D)                                                                    \
    return d->numax <= 5 * 256 ? launch_dense_t<Q, K, D, 4, 5>(d, coeff, do_clamp, d_x, d_y, s)                   \
                               : launch_dense_t<Q, K, D, 4, 9>(d, coeff, do_clamp, d_x, d_y, s);

// Compiled shapes: Lagrange P1..P4 on the tetrahedron with the m = p Gauss-Jacobi
// rule (nd, nq) = (4,1) (10,8) (20,27) (35,64), plus P4 with the m = 3 rule.
int launch_stiffness_dense(const DenseOpData* d, double coeff, int do_clamp, const double* d_x, double* d_y,
                           hipStream_t s)
{
  if (d->nbatch == 0) return WF_OK;
  WF_DENSE_CASE(1, 1, 1)
  WF_DENSE_CASE(1, 3, 1)
  WF_DENSE_CASE(2, 5, 2)
  WF_DENSE_CASE(4, 9, 3)
  WF_DENSE_CASE(2, 9, 3)
  set_error("stiffness_dense: (nd, nq) shape not compiled (supported: tetrahedron P1..P4)");
  return WF_ERR_UNSUPPORTED;
}

}  // namespace wf
