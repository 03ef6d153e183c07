// Matrix-free conjugate gradients on the operators of this library (BP1 of
// demo/gpu_cg: device::cg, demo/gpu_cg/CUDA/cg.hpp:38-121, with the vector kernels of
// CUDA/streaming.hpp:29-137 and the scalar MPI_Allreduce of cg.hpp:15-24).
//
// Same signature idea (x, b, matvec, kmax, rtol; returns the iteration count) and the
// same stopping rule (||r||^2 / ||r0||^2 < rtol^2, cg.hpp:103), but the textbook
// algorithm: the reference loop reverse-updates the search direction instead of the
// product (cg.hpp:84), treats cublasDnrm2's result as a squared norm (cg.hpp:59,97) and
// updates the direction with axpy(1, p, r) (cg.hpp:117) -- none of which is reproduced.
//
// All scalars stay on the device: alpha and beta are formed inside the vector kernels
// from the reduction results, so one iteration is
//   [fwd halo] y = A p [rev halo] ; pAp = <p, y> ; {x += a p ; r -= a y ; rr' = <r, r>} ; p = r + b p
// = the operator + 88 B/dof, with ONE host read (8 bytes) per iteration for the stopping test.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

namespace wf {

namespace {

constexpr int kS_rr = 0, kS_pAp = 1, kS_rr_new = 2, kS_count = 4;

inline unsigned grid_for(int64_t n)
{
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 2048 ? (g ? g : 1) : 2048);
}

__device__ inline void block_add(double s, double* target)
{
  __shared__ double part[4];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(target, part[0] + part[1] + part[2] + part[3]);
}

// target += <x, y>
__global__ void __launch_bounds__(256)
k_cg_dot(int64_t n, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ target)
{
  double s = 0.0;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) s += x[g] * y[g];
  block_add(s, target);
}

// a[idx[i]] = 0 (and b[idx[i]] = 0): the ghost entries of the product and of the residual are kept at
// exactly zero, so that whole-array reductions count every global dof once (the reference reduces
// over size_local entries, streaming.hpp:93,111; here ghosts are interleaved with owned dofs)
__global__ void __launch_bounds__(256) k_cg_zero_ghosts(int32_t n, const int32_t* __restrict__ idx, double* a, double* b)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    a[idx[g]] = 0.0;
    if (b) b[idx[g]] = 0.0;
  }
}

// r = b - y  (initial residual; y = A x0)
__global__ void __launch_bounds__(256) k_cg_residual(int64_t n, const double* __restrict__ b, const double* __restrict__ y,
                                                     double* __restrict__ r, double* __restrict__ p)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    const double v = b[g] - y[g];
    r[g] = v;
    p[g] = v;
  }
}

// alpha = rr / pAp ; x += alpha p ; r -= alpha y ; rr_new += <r, r>
__global__ void __launch_bounds__(256)
k_cg_update(int64_t n, const double* __restrict__ sc, const double* __restrict__ p, const double* __restrict__ y,
            double* __restrict__ x, double* __restrict__ r, double* __restrict__ rr_new)
{
  const double alpha = sc[kS_rr] / sc[kS_pAp];
  double s = 0.0;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    x[g] += alpha * p[g];
    const double v = r[g] - alpha * y[g];
    r[g] = v;
    s += v * v;
  }
  block_add(s, rr_new);
}

// beta = rr_new / rr ; p = r + beta p
__global__ void __launch_bounds__(256)
k_cg_direction(int64_t n, const double* __restrict__ sc, const double* __restrict__ r, double* __restrict__ p)
{
  const double beta = sc[kS_rr_new] / sc[kS_rr];
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
    p[g] = r[g] + beta * p[g];
}

// rr <- rr_new ; rr_new, pAp <- 0   (one thread)
__global__ void k_cg_roll(double* sc)
{
  sc[kS_rr] = sc[kS_rr_new];
  sc[kS_rr_new] = 0.0;
  sc[kS_pAp] = 0.0;
}

struct Bufs {
  double *r = nullptr, *p = nullptr, *y = nullptr, *sc = nullptr;
  ~Bufs()
  {
    (void)hipFree(r);
    (void)hipFree(p);
    (void)hipFree(y);
    (void)hipFree(sc);
  }
};

}  // namespace
}  // namespace wf

using namespace wf;

// defined in comm.hip
extern "C" int wf_updater_ghosts(const wf_updater* u, const int32_t** d_ghost_pos, int32_t* nghost);

extern "C" int wf_cg(const wf_cg_desc* d, double* d_x, const double* d_b, int* iterations, double* rel_residual, void* stream)
{
  WF_REQUIRE(d && d_x && d_b, "wf_cg: null argument");
  WF_REQUIRE(d->n >= 0 && d->kmax >= 0 && d->rtol >= 0.0, "wf_cg: bad size / kmax / rtol");
  WF_REQUIRE(d->op || d->matvec, "wf_cg: needs an operator handle or a matvec callback");
  WF_REQUIRE(!d->updater || d->comm, "wf_cg: an updater needs its communicator (global reductions)");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = d->n;
  if (iterations) *iterations = 0;
  if (rel_residual) *rel_residual = 0.0;
  if (n == 0 && !d->comm) return WF_OK;
  Bufs B;
  const size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(double);
  WF_HIP_CHECK(hipMalloc((void**)&B.r, bytes));
  WF_HIP_CHECK(hipMalloc((void**)&B.p, bytes));
  WF_HIP_CHECK(hipMalloc((void**)&B.y, bytes));
  WF_HIP_CHECK(hipMalloc((void**)&B.sc, kS_count * sizeof(double)));
  WF_HIP_CHECK(hipMemsetAsync(B.sc, 0, kS_count * sizeof(double), s));
  const int32_t* d_ghost = nullptr;
  int32_t nghost = 0;
  int rc;
  if (d->updater && (rc = wf_updater_ghosts(d->updater, &d_ghost, &nghost)) != WF_OK) return rc;

  // y = A v with the halo exchanges of a partitioned mesh around it (LinearGLL.hpp:164-176)
  auto matvec = [&](double* v, double* y) -> int {
    int e;
    if (d->updater && (e = wf_updater_fwd(d->updater, v, s)) != WF_OK) return e;
    WF_HIP_CHECK(hipMemsetAsync(y, 0, bytes, s));     // the operators accumulate (y += A v)
    if (d->matvec) {
      if ((e = d->matvec(d->user, v, y, s)) != WF_OK) {
        set_error("wf_cg: matvec callback failed");
        return e;
      }
    } else if ((e = wf_op_apply(d->op, v, y, s)) != WF_OK)
      return e;
    if (d->updater && (e = wf_updater_rev(d->updater, y, s)) != WF_OK) return e;
    if (nghost) hipLaunchKernelGGL(k_cg_zero_ghosts, dim3(grid_for(nghost)), dim3(256), 0, s, nghost, d_ghost, y, (double*)nullptr);
    return WF_OK;
  };
  // global <a, b> into sc[slot]: one of the two operands is zero on the ghost entries, so the
  // whole-array sum is the sum over owned entries; then the scalar all-reduce (cg.hpp:15-24)
  auto finish_dot = [&](const double* a, const double* b, int slot, bool local_done) -> int {
    if (!local_done && n) hipLaunchKernelGGL(k_cg_dot, dim3(grid_for(n)), dim3(256), 0, s, n, a, b, B.sc + slot);
    if (d->comm) return wf_comm_allreduce(d->comm, WF_SUM, 1, B.sc + slot, B.sc + slot, s);
    return WF_OK;
  };
  auto read = [&](int slot, double* v) -> int {
    WF_HIP_CHECK(hipMemcpyAsync(v, B.sc + slot, sizeof(double), hipMemcpyDeviceToHost, s));
    WF_HIP_CHECK(hipStreamSynchronize(s));
    return WF_OK;
  };

  // r = b - A x0 ; p = r ; rr = <r, r>
  if ((rc = matvec(d_x, B.y)) != WF_OK) return rc;
  if (n) hipLaunchKernelGGL(k_cg_residual, dim3(grid_for(n)), dim3(256), 0, s, n, d_b, B.y, B.r, B.p);
  if (nghost) hipLaunchKernelGGL(k_cg_zero_ghosts, dim3(grid_for(nghost)), dim3(256), 0, s, nghost, d_ghost, B.r, B.p);
  if ((rc = finish_dot(B.r, B.r, kS_rr, false)) != WF_OK) return rc;
  double rr0 = 0.0, rr = 0.0;
  if ((rc = read(kS_rr, &rr0)) != WF_OK) return rc;
  rr = rr0;
  const double rtol2 = d->rtol * d->rtol;
#ifdef WF_DIAG
  const bool verbose = std::getenv("WF_CG_VERBOSE") != nullptr;
#else
  constexpr bool verbose = false;
#endif
  if (verbose) std::fprintf(stderr, "wf_cg: n = %lld rnorm0 = %.17g\n", (long long)n, rr0);
  int k = 0;
  if (rr0 > 0.0) {
    while (k < d->kmax) {
      ++k;
      if ((rc = matvec(B.p, B.y)) != WF_OK) return rc;
      if ((rc = finish_dot(B.p, B.y, kS_pAp, false)) != WF_OK) return rc;
      if (n) hipLaunchKernelGGL(k_cg_update, dim3(grid_for(n)), dim3(256), 0, s, n, B.sc, B.p, B.y, d_x, B.r, B.sc + kS_rr_new);
      if ((rc = finish_dot(B.r, B.r, kS_rr_new, true)) != WF_OK) return rc;
      if ((rc = read(kS_rr_new, &rr)) != WF_OK) return rc;
      if (verbose) std::fprintf(stderr, "wf_cg: it %d rnorm = %.17g\n", k, rr);   // cg.hpp:101 LOG(INFO) << "rnorm = "
      if (!(rr == rr)) {
        set_error("wf_cg: residual is not finite (operator not positive definite?)");
        return WF_ERR_INVALID;
      }
      if (rr / rr0 < rtol2) break;                                   // cg.hpp:103
      if (n) hipLaunchKernelGGL(k_cg_direction, dim3(grid_for(n)), dim3(256), 0, s, n, B.sc, B.r, B.p);
      hipLaunchKernelGGL(k_cg_roll, dim3(1), dim3(1), 0, s, B.sc);
    }
  }
  WF_HIP_CHECK(hipStreamSynchronize(s));
  if (iterations) *iterations = k;
  if (rel_residual) *rel_residual = rr0 > 0.0 ? std::sqrt(rr / rr0) : 0.0;
  return WF_OK;
}
