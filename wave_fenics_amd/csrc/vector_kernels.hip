// Streaming kernels of the RK4 loop and the free gather/scatter/transform
// kernels (a8, a9, a15, a16).  All are HBM-bound; every kernel is
// a grid-stride loop capped at 256 CUs x 8 workgroups, 16-byte accesses where
// the operands allow it.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace wf {

static inline unsigned capped_grid(size_t n, unsigned block)
{
  size_t g = (n + block - 1) / block;
  const size_t cap = 256u * 8u;
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

#define WF_LAUNCH_CHECK()                                                       \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      set_error(std::string("kernel launch failed: ") + hipGetErrorString(_e)); \
      return WF_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

// common/cuda/scatter.cu:5-11
__global__ void k_gather(int32_t N, const int32_t* __restrict__ idx, const double* __restrict__ in,
                         double* __restrict__ out)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += (int64_t)gridDim.x * blockDim.x)
    out[g] = in[idx[g]];
}
// common/cuda/scatter.cu:39-45 (hardware global_atomic_add_f64, no CAS loop)
__global__ void k_scatter_add(int32_t N, const int32_t* __restrict__ idx, const double* __restrict__ in,
                              double* __restrict__ out)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += (int64_t)gridDim.x * blockDim.x)
    unsafeAtomicAdd(&out[idx[g]], in[g]);
}
__global__ void k_scatter_set(int32_t N, const int32_t* __restrict__ idx, const double* __restrict__ in,
                              double* __restrict__ out)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += (int64_t)gridDim.x * blockDim.x)
    out[idx[g]] = in[g];
}
// common/cuda/transform.cu:6-11
__global__ void k_transform1(int32_t N, const double* __restrict__ in, const double* __restrict__ detJ,
                             double* __restrict__ out)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < N; g += (int64_t)gridDim.x * blockDim.x)
    out[g] = in[g] * detJ[g];
}

__global__ void k_fill(int64_t n, double v, double* __restrict__ out)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
    out[g] = v;
}
// kernels::axpy of common/LinearGLL.hpp:28-33: r = x*alpha + y (r may alias y)
__global__ void k_axpy(int64_t n, double alpha, const double* x, const double* y, double* r)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
    r[g] = x[g] * alpha + y[g];
}
__global__ void k_scale(int64_t n, double alpha, double* x)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
    x[g] *= alpha;
}
// common/LinearGLL.hpp:189-190: out = b / m
__global__ void k_div(int64_t n, const double* b, const double* m, double* out)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
    out[g] = b[g] / m[g];
}
// The same with 16-byte accesses; the diagonal m and the result bypass the caches (non-temporal): inside the
// RK4 loop / the bench step they are streamed once per pass, while b (= y of the stiffness apply) and the
// apply's x are what the next stiffness apply touches again and should keep the Infinity Cache
// (cfg2: 4 x 82 MB of vectors do not fit its 256 MB, x + y do).
__global__ void k_div2(int64_t npairs, const double* __restrict__ b, const double* __restrict__ m, double* __restrict__ out)
{
  typedef double d2v __attribute__((ext_vector_type(2)));
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += (int64_t)gridDim.x * blockDim.x) {
    const d2v bb = reinterpret_cast<const d2v*>(b)[g];
    const d2v mm = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(m) + g);
    d2v r;
    r.x = bb.x / mm.x;
    r.y = bb.y / mm.y;
    __builtin_nontemporal_store(r, reinterpret_cast<d2v*>(out) + g);
  }
}
__global__ void k_mult_add(int64_t n, const double* __restrict__ m, const double* __restrict__ x, double* y)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
    y[g] += m[g] * x[g];
}
// y += m .* x with 16-byte accesses, the diagonal streamed past the caches
__global__ void k_mult_add2(int64_t npairs, const double* __restrict__ m, const double* __restrict__ x, double* __restrict__ y)
{
  typedef double d2v __attribute__((ext_vector_type(2)));
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < npairs; g += (int64_t)gridDim.x * blockDim.x) {
    const d2v mm = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(m) + g);
    const d2v xx = reinterpret_cast<const d2v*>(x)[g];
    d2v yy = reinterpret_cast<d2v*>(y)[g];
    yy.x += mm.x * xx.x;
    yy.y += mm.y * xx.y;
    reinterpret_cast<d2v*>(y)[g] = yy;
  }
}
// common/cuda/la.hpp:87-103 inner_product: wave shuffle -> LDS -> one atomic per workgroup
__global__ void k_dot(int64_t n, const double* __restrict__ x, const double* __restrict__ y,
                      double* __restrict__ result)
{
  __shared__ double part[4];
  double s = 0.0;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x)
    s += x[g] * y[g];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(result, part[0] + part[1] + part[2] + part[3]);
}
// diagonal boundary operator, LinearGLL.hpp:175 / forms.ufl:19-24
__global__ void k_boundary(int32_t n1, const int32_t* __restrict__ idx1, const double* __restrict__ m1,
                           double s1, int32_t n2, const int32_t* __restrict__ idx2,
                           const double* __restrict__ m2, double s2, const double* __restrict__ v,
                           double* __restrict__ b)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < (int64_t)n1 + n2;
       g += (int64_t)gridDim.x * blockDim.x) {
    if (g < n1) {
      unsafeAtomicAdd(&b[idx1[g]], s1 * m1[g]);
    } else {
      const int64_t h = g - n1;
      const int32_t d = idx2[h];
      unsafeAtomicAdd(&b[d], s2 * m2[h] * v[d]);
    }
  }
}

// Fused tail of one RK4 stage (common/LinearGLL.hpp:182-191 divide, :260 f0,
// :264-265 solution update) plus the head of the next stage (:250-254) and the
// zeroing of b for the next stiffness apply (:173), so that the vector algebra
// between two stiffness applies is ONE pass:
//   kv = b / m ; ku = vn
//   u_ = ku*(dt*b_i) + u_read ; v_ = kv*(dt*b_i) + v_read ; b = 0
//   un = ku*(dt*a_next) + u0  ; vn_next = kv*(dt*a_next) + v0              (if has_next)
// HBM traffic per dof: stage 0 (u_read = u0, v_read = vn = v0) 4 reads + 5 writes = 72 B,
// stages 1-2 7 + 5 = 96 B, stage 3 5 + 3 = 64 B -- 82 B/stage on average against the
// 208 B/dof of the reference's separate copy/axpy/fill/transform passes.
// u_read/v_read may alias u_/v_ (stages 1..3) or be u0/v0 (stage 0).
// VEC = 2: 16-byte accesses (all pointers 16-byte aligned, n counted in pairs).
template <int VEC, bool HAS_NEXT, bool NT = false>
__global__ void __launch_bounds__(256)
k_rk4_stage(int64_t n, double bdt, double adt_next, double* b, const double* m, const double* vn,
            const double* u_read, const double* v_read, double* u_, double* v_, const double* u0,
            const double* v0, double* un, double* vn_next)
{
  using V = typename std::conditional<VEC == 2, double2, double>::type;
  typedef double d2v __attribute__((ext_vector_type(2)));
  // NT: everything except the two vectors the next stiffness apply touches (un, b) bypasses the caches
  auto ld = [](const double* p, int64_t g) -> V {
    if constexpr (NT && VEC == 2) {
      const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p) + g);
      return make_double2(v.x, v.y);
    } else
      return reinterpret_cast<const V*>(p)[g];
  };
  auto st = [](double* p, int64_t g, V v) { reinterpret_cast<V*>(p)[g] = v; };
  auto stnt = [](double* p, int64_t g, V v) {
    if constexpr (NT && VEC == 2) {
      d2v w;
      w.x = v.x;
      w.y = v.y;
      __builtin_nontemporal_store(w, reinterpret_cast<d2v*>(p) + g);
    } else
      reinterpret_cast<V*>(p)[g] = v;
  };
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (int64_t)gridDim.x * blockDim.x) {
    const V bb = ld(b, g), mm = ld(m, g), ku = ld(vn, g), ur = ld(u_read, g), vr = ld(v_read, g);
    V kv, uo, vo, zero;
    if constexpr (VEC == 2) {
      kv = make_double2(bb.x / mm.x, bb.y / mm.y);
      uo = make_double2(ku.x * bdt + ur.x, ku.y * bdt + ur.y);
      vo = make_double2(kv.x * bdt + vr.x, kv.y * bdt + vr.y);
      zero = make_double2(0.0, 0.0);
    } else {
      kv = bb / mm;
      uo = ku * bdt + ur;
      vo = kv * bdt + vr;
      zero = 0.0;
    }
    if constexpr (HAS_NEXT) {
      const V a0 = ld(u0, g), c0 = ld(v0, g);
      V un_, vn_;
      if constexpr (VEC == 2) {
        un_ = make_double2(ku.x * adt_next + a0.x, ku.y * adt_next + a0.y);
        vn_ = make_double2(kv.x * adt_next + c0.x, kv.y * adt_next + c0.y);
      } else {
        un_ = ku * adt_next + a0;
        vn_ = kv * adt_next + c0;
      }
      st(un, g, un_);
      stnt(vn_next, g, vn_);
    }
    stnt(u_, g, uo);
    stnt(v_, g, vo);
    st(b, g, zero);
  }
}

}  // namespace wf

using namespace wf;

extern "C" {

int wf_gather(int32_t N, const int32_t* d_indices, const double* d_in, double* d_out, void* stream)
{
  if (N <= 0) return WF_OK;
  hipLaunchKernelGGL(k_gather, dim3(capped_grid(N, 256)), dim3(256), 0, (hipStream_t)stream, N, d_indices, d_in,
                     d_out);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_scatter_add(int32_t N, const int32_t* d_indices, const double* d_in, double* d_out, void* stream)
{
  if (N <= 0) return WF_OK;
  hipLaunchKernelGGL(k_scatter_add, dim3(capped_grid(N, 256)), dim3(256), 0, (hipStream_t)stream, N, d_indices,
                     d_in, d_out);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_scatter_set(int32_t N, const int32_t* d_indices, const double* d_in, double* d_out, void* stream)
{
  if (N <= 0) return WF_OK;
  hipLaunchKernelGGL(k_scatter_set, dim3(capped_grid(N, 256)), dim3(256), 0, (hipStream_t)stream, N, d_indices,
                     d_in, d_out);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_transform1(int32_t N, const double* d_in, const double* d_detJ, double* d_out, void* stream)
{
  if (N <= 0) return WF_OK;
  hipLaunchKernelGGL(k_transform1, dim3(capped_grid(N, 256)), dim3(256), 0, (hipStream_t)stream, N, d_in, d_detJ,
                     d_out);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_copy(int64_t n, const double* d_in, double* d_out, void* stream)
{
  if (n <= 0) return WF_OK;
  WF_HIP_CHECK(hipMemcpyAsync(d_out, d_in, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
  return WF_OK;
}
int wf_fill(int64_t n, double value, double* d_out, void* stream)
{
  if (n <= 0) return WF_OK;
  hipLaunchKernelGGL(k_fill, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, value, d_out);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_axpy(int64_t n, double alpha, const double* d_x, const double* d_y, double* d_r, void* stream)
{
  if (n <= 0) return WF_OK;
  hipLaunchKernelGGL(k_axpy, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, alpha, d_x, d_y,
                     d_r);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_scale(int64_t n, double alpha, double* d_x, void* stream)
{
  if (n <= 0) return WF_OK;
  hipLaunchKernelGGL(k_scale, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, alpha, d_x);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_pointwise_div(int64_t n, const double* d_b, const double* d_m, double* d_out, void* stream)
{
  if (n <= 0) return WF_OK;
  const bool vec = ((reinterpret_cast<uintptr_t>(d_b) | reinterpret_cast<uintptr_t>(d_m) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0
                   && d_out != d_b && d_out != d_m;
  const int64_t nv = vec ? n / 2 : 0;
  if (nv) hipLaunchKernelGGL(k_div2, dim3(capped_grid(nv, 256)), dim3(256), 0, (hipStream_t)stream, nv, d_b, d_m, d_out);
  if (n > 2 * nv)
    hipLaunchKernelGGL(k_div, dim3(capped_grid(n - 2 * nv, 256)), dim3(256), 0, (hipStream_t)stream, n - 2 * nv, d_b + 2 * nv,
                       d_m + 2 * nv, d_out + 2 * nv);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_pointwise_mult_add(int64_t n, const double* d_m, const double* d_x, double* d_y, void* stream)
{
  if (n <= 0) return WF_OK;
  const bool vec = ((reinterpret_cast<uintptr_t>(d_m) | reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y)) & 15) == 0
                   && d_y != d_x && d_y != d_m;
  const int64_t nv = vec ? n / 2 : 0;
  if (nv) hipLaunchKernelGGL(k_mult_add2, dim3(capped_grid(nv, 256)), dim3(256), 0, (hipStream_t)stream, nv, d_m, d_x, d_y);
  if (n > 2 * nv)
    hipLaunchKernelGGL(k_mult_add, dim3(capped_grid(n - 2 * nv, 256)), dim3(256), 0, (hipStream_t)stream, n - 2 * nv, d_m + 2 * nv,
                       d_x + 2 * nv, d_y + 2 * nv);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_dot(int64_t n, const double* d_x, const double* d_y, double* d_result, void* stream)
{
  WF_HIP_CHECK(hipMemsetAsync(d_result, 0, sizeof(double), (hipStream_t)stream));
  if (n <= 0) return WF_OK;
  hipLaunchKernelGGL(k_dot, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, d_x, d_y, d_result);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_rk4_stage(int64_t n, double bdt, double adt_next, int has_next, double* d_b, const double* d_m,
                 const double* d_vn, const double* d_u_read, const double* d_v_read, double* d_u, double* d_v,
                 const double* d_u0, const double* d_v0, double* d_un, double* d_vn_next, void* stream)
{
  if (n <= 0) return WF_OK;
  WF_REQUIRE(d_b && d_m && d_vn && d_u_read && d_v_read && d_u && d_v, "wf_rk4_stage: null vector");
  WF_REQUIRE(!has_next || (d_u0 && d_v0 && d_un && d_vn_next), "wf_rk4_stage: next-stage vectors missing");
  MarkerScope mk("wf_rk4_stage");
  WF_REQUIRE(!has_next || d_vn_next != d_vn, "wf_rk4_stage: vn_next must not alias vn");
  hipStream_t st = (hipStream_t)stream;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool vec = aligned(d_b) && aligned(d_m) && aligned(d_vn) && aligned(d_u_read) && aligned(d_v_read) && aligned(d_u)
             && aligned(d_v);
  if (has_next) vec = vec && aligned(d_u0) && aligned(d_v0) && aligned(d_un) && aligned(d_vn_next);
  const int64_t nv = vec ? n / 2 : 0;   // pairs handled by the 16-byte kernel; the rest (at most one entry) scalar
  constexpr bool nt = true;   // streaming (non-temporal) accesses for everything but un and b (measured 0.155 -> 0.135 ms)
#define WF_STAGE(VEC, NEXT, cnt, off)                                                                          \
  if (nt && VEC == 2)                                                                                          \
    hipLaunchKernelGGL((k_rk4_stage<VEC, NEXT, true>), dim3(capped_grid((cnt), 256)), dim3(256), 0, st, (cnt), bdt, \
                     adt_next, d_b + (off), d_m + (off), d_vn + (off), d_u_read + (off), d_v_read + (off),      \
                     d_u + (off), d_v + (off), NEXT ? d_u0 + (off) : nullptr, NEXT ? d_v0 + (off) : nullptr,   \
                     NEXT ? d_un + (off) : nullptr, NEXT ? d_vn_next + (off) : nullptr);                        \
  else                                                                                                         \
  hipLaunchKernelGGL((k_rk4_stage<VEC, NEXT>), dim3(capped_grid((cnt), 256)), dim3(256), 0, st, (cnt), bdt,     \
                     adt_next, d_b + (off), d_m + (off), d_vn + (off), d_u_read + (off), d_v_read + (off),      \
                     d_u + (off), d_v + (off), NEXT ? d_u0 + (off) : nullptr, NEXT ? d_v0 + (off) : nullptr,   \
                     NEXT ? d_un + (off) : nullptr, NEXT ? d_vn_next + (off) : nullptr)
  if (nv > 0) {
    if (has_next) WF_STAGE(2, true, nv, 0); else WF_STAGE(2, false, nv, 0);
    WF_LAUNCH_CHECK();
  }
  const int64_t rest = n - 2 * nv;
  if (rest > 0) {
    if (has_next) WF_STAGE(1, true, rest, 2 * nv); else WF_STAGE(1, false, rest, 2 * nv);
    WF_LAUNCH_CHECK();
  }
#undef WF_STAGE
  return WF_OK;
}

int wf_boundary_apply(int32_t n1, const int32_t* d_idx1, const double* d_m1, double s1, int32_t n2,
                      const int32_t* d_idx2, const double* d_m2, double s2, const double* d_v, double* d_b,
                      void* stream)
{
  const int64_t n = (int64_t)n1 + n2;
  if (n <= 0) return WF_OK;
  hipLaunchKernelGGL(k_boundary, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n1, d_idx1, d_m1,
                     s1, n2, d_idx2, d_m2, s2, d_v, d_b);
  WF_LAUNCH_CHECK();
  return WF_OK;
}

}  // extern "C"
