// Streaming kernels of the RK4 loop and the free gather/scatter/transform
// kernels (a8, a9, a15, a16).  All are HBM-bound; every kernel is
// a grid-stride loop capped at 256 CUs x 8 workgroups, 16-byte accesses where
// the operands allow it.
#include <cstdlib>
#include <memory>
#include <type_traits>
#include <vector>
#include <type_traits>

#include "common.h"

namespace wf {

static inline unsigned capped_grid(size_t n, unsigned block)
{
  // One entry per thread up to 2^22 workgroups (the loops below are grid-stride, so any n is covered).  A grid capped
  // at 8 workgroups per CU, every thread looping ~10 times, measured 10 % slower on the streaming kernels in the
  // sustained RK4 loop (stage kernel 0.152 -> 0.136 ms, step 1.372 -> 1.308 ms): fresh waves keep more independent
  // requests in flight than the iterations of a resident one.
  size_t g = (n + block - 1) / block;
  const size_t cap = (size_t)1 << 22;
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}
// reductions (one atomic per workgroup): 8 workgroups per CU
static inline unsigned reduction_grid(size_t n, unsigned block)
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
// 16-byte forms of fill / axpy / scale (aligned arrays; the odd last entry by thread 0 of the last workgroup)
__global__ void k_fill2(int64_t npairs, int tail, double v, double* out)
{
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < npairs) reinterpret_cast<double2*>(out)[g] = make_double2(v, v);
  if (tail && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[2 * npairs] = v;
}
__global__ void k_axpy2(int64_t npairs, int tail, double alpha, const double* x, const double* y, double* r)
{
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < npairs) {   // r may alias y (or x): every thread reads its entry before it writes it
    const double2 xx = reinterpret_cast<const double2*>(x)[g], yy = reinterpret_cast<const double2*>(y)[g];
    reinterpret_cast<double2*>(r)[g] = make_double2(xx.x * alpha + yy.x, xx.y * alpha + yy.y);
  }
  if (tail && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) r[2 * npairs] = x[2 * npairs] * alpha + y[2 * npairs];
}
__global__ void k_scale2(int64_t npairs, int tail, double alpha, double* x)
{
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < npairs) {
    double2 v = reinterpret_cast<double2*>(x)[g];
    v.x *= alpha;
    v.y *= alpha;
    reinterpret_cast<double2*>(x)[g] = v;
  }
  if (tail && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) x[2 * npairs] *= alpha;
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
// `tail` (0 or 1): the odd last entry, done by the first thread of the last block in the same launch (as its own
// launch it cost 4 us of the 44 us the mass-inverse takes in the bench step).
__global__ void k_div2(int64_t npairs, int tail, const double* __restrict__ b, const double* __restrict__ m, double* __restrict__ out)
{
  typedef double d2v __attribute__((ext_vector_type(2)));
  if (tail && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[2 * npairs] = b[2 * npairs] / m[2 * npairs];
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
__global__ void k_mult_add2(int64_t npairs, int tail, const double* __restrict__ m, const double* __restrict__ x, double* __restrict__ y)
{
  typedef double d2v __attribute__((ext_vector_type(2)));
  if (tail && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) y[2 * npairs] += m[2 * npairs] * x[2 * npairs];
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
// BC: instead of zeroing b for the next stiffness apply, leave the NEXT right-hand side's diagonal boundary
// term in it (LinearGLL.hpp:175, forms.ufl:19-24): b[i] = s1 c1[r] + s2 c2[r] v'[i] for the dofs of the
// boundary sets, v' = the v argument of the next f1 (vn_next, or the updated solution after the last stage).
// The sets are a bitmap over the dofs (one 64-bit word per 64 dofs, L2-resident), a running count per word and
// the two coefficient arrays over the union set in dof order: 0.2 B/dof read instead of a fifth launch per stage.
struct BoundaryPlanDev {
  const unsigned long long* mask = nullptr;   // [ceil(n / 64)]
  const uint32_t* prefix = nullptr;           // [ceil(n / 64)] boundary dofs in front of the word
  const double* c1 = nullptr;                 // [nb] facet mass of Gamma_1 (0 for dofs only in Gamma_2)
  const double* c2 = nullptr;                 // [nb] facet mass of Gamma_2
  double s1 = 0.0, s2 = 0.0;
};

struct StageArgs {
  double bdt, adt_next;
  double* b;
  const double* m;
  const double* vn;
  const double* u_read;
  const double* v_read;
  double* u_;
  double* v_;
  const double* u0;
  const double* v0;
  double* un;
  double* vn_next;
  BoundaryPlanDev bc;
};

// entry g of the stage (VEC doubles wide) of arrays that start at dof `off`
template <int VEC, bool HAS_NEXT, bool NT, bool BC>
__device__ __forceinline__ void stage_entry(const StageArgs& a, int64_t g, int64_t off)
{
  using V = typename std::conditional<VEC == 2, double2, double>::type;
  typedef double d2v __attribute__((ext_vector_type(2)));
  // NT: everything except the two vectors the next stiffness apply touches (un, b) bypasses the caches
  auto ld = [off](const double* p, int64_t g) -> V {
    if constexpr (NT && VEC == 2) {
      const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(p + off) + g);
      return make_double2(v.x, v.y);
    } else
      return reinterpret_cast<const V*>(p + off)[g];
  };
  auto st = [off](double* p, int64_t g, V v) { reinterpret_cast<V*>(p + off)[g] = v; };
  auto stnt = [off](double* p, int64_t g, V v) {
    if constexpr (NT && VEC == 2) {
      d2v w;
      w.x = v.x;
      w.y = v.y;
      __builtin_nontemporal_store(w, reinterpret_cast<d2v*>(p + off) + g);
    } else
      reinterpret_cast<V*>(p + off)[g] = v;
  };
  const double bdt = a.bdt, adt_next = a.adt_next;
  const V bb = ld(a.b, g), mm = ld(a.m, g), ku = ld(a.vn, g), ur = ld(a.u_read, g), vr = ld(a.v_read, g);
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
  V vnext = vo;   // the v the next right-hand side sees
  if constexpr (HAS_NEXT) {
    const V a0 = ld(a.u0, g), c0 = ld(a.v0, g);
    V un_, vn_;
    if constexpr (VEC == 2) {
      un_ = make_double2(ku.x * adt_next + a0.x, ku.y * adt_next + a0.y);
      vn_ = make_double2(kv.x * adt_next + c0.x, kv.y * adt_next + c0.y);
    } else {
      un_ = ku * adt_next + a0;
      vn_ = kv * adt_next + c0;
    }
    st(a.un, g, un_);
    stnt(a.vn_next, g, vn_);
    vnext = vn_;
  }
  stnt(a.u_, g, uo);
  stnt(a.v_, g, vo);
  if constexpr (BC) {
    const BoundaryPlanDev& bc = a.bc;
    const int64_t d0 = off + (int64_t)VEC * g;          // first dof of this entry
    unsigned long long word;
    bool any = true;
    if constexpr (VEC == 2) {
      // a wave covers 128 consecutive dofs = two mask words (off is 0 and g a multiple of 64 in lane 0): fetch
      // them with scalar loads and skip the whole block when the wave holds no boundary dof (97 % of the waves)
      const int64_t w0 = __builtin_amdgcn_readfirstlane((int)(d0 >> 6));
      const unsigned long long wa = bc.mask[w0], wb = bc.mask[w0 + 1];   // mask has one spare word at the end
      any = (wa | wb) != 0ull;
      word = (d0 >> 6) == w0 ? wa : wb;
    } else {
      word = bc.mask[d0 >> 6];
    }
    if (any) {
      const unsigned bits = (unsigned)(word >> (d0 & 63)) & (VEC == 2 ? 3u : 1u);   // VEC == 2: d0 is even, both dofs in one word
      if (bits) {
        const unsigned long long below = word & ((1ull << (d0 & 63)) - 1ull);
        int64_t r = (int64_t)bc.prefix[d0 >> 6] + __popcll(below);
        if constexpr (VEC == 2) {
          if (bits & 1u) {
            zero.x = bc.s1 * bc.c1[r] + bc.s2 * bc.c2[r] * vnext.x;
            ++r;
          }
          if (bits & 2u) zero.y = bc.s1 * bc.c1[r] + bc.s2 * bc.c2[r] * vnext.y;
        } else {
          zero = bc.s1 * bc.c1[r] + bc.s2 * bc.c2[r] * vnext;
        }
      }
    }
  }
  st(a.b, g, zero);
}

// nvec entries of width VEC, then `ntail` single entries (at most VEC - 1: the odd last dof of a 16-byte sweep) by
// the first threads of the last block -- one launch whatever the vector length
template <int VEC, bool HAS_NEXT, bool NT = false, bool BC = false>
__global__ void __launch_bounds__(256) k_rk4_stage(int64_t nvec, int ntail, StageArgs a)
{
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < nvec; g += (int64_t)gridDim.x * blockDim.x)
    stage_entry<VEC, HAS_NEXT, NT, BC>(a, g, 0);
  if (VEC > 1 && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < ntail)
    stage_entry<1, HAS_NEXT, false, BC>(a, threadIdx.x, (int64_t)VEC * nvec);
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
  const int64_t np2 = n / 2;
  if ((reinterpret_cast<uintptr_t>(d_out) & 15) == 0 && np2 > 0 && np2 < ((int64_t)1 << 30))
    hipLaunchKernelGGL(k_fill2, dim3((unsigned)((np2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, np2, (int)(n - 2 * np2), value, d_out);
  else
    hipLaunchKernelGGL(k_fill, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, value, d_out);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_axpy(int64_t n, double alpha, const double* d_x, const double* d_y, double* d_r, void* stream)
{
  if (n <= 0) return WF_OK;
  const int64_t np2 = n / 2;
  if (((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y) | reinterpret_cast<uintptr_t>(d_r)) & 15) == 0 && np2 > 0 &&
      np2 < ((int64_t)1 << 30))
    hipLaunchKernelGGL(k_axpy2, dim3((unsigned)((np2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, np2, (int)(n - 2 * np2), alpha, d_x,
                       d_y, d_r);
  else
    hipLaunchKernelGGL(k_axpy, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, alpha, d_x, d_y, d_r);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_scale(int64_t n, double alpha, double* d_x, void* stream)
{
  if (n <= 0) return WF_OK;
  const int64_t np2 = n / 2;
  if ((reinterpret_cast<uintptr_t>(d_x) & 15) == 0 && np2 > 0 && np2 < ((int64_t)1 << 30))
    hipLaunchKernelGGL(k_scale2, dim3((unsigned)((np2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, np2, (int)(n - 2 * np2), alpha, d_x);
  else
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
  if (nv)
    hipLaunchKernelGGL(k_div2, dim3(capped_grid(nv, 256)), dim3(256), 0, (hipStream_t)stream, nv, (int)(n - 2 * nv), d_b, d_m, d_out);
  else
    hipLaunchKernelGGL(k_div, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, d_b, d_m, d_out);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_pointwise_mult_add(int64_t n, const double* d_m, const double* d_x, double* d_y, void* stream)
{
  if (n <= 0) return WF_OK;
  const bool vec = ((reinterpret_cast<uintptr_t>(d_m) | reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y)) & 15) == 0
                   && d_y != d_x && d_y != d_m;
  const int64_t nv = vec ? n / 2 : 0;
  if (nv)
    hipLaunchKernelGGL(k_mult_add2, dim3(capped_grid(nv, 256)), dim3(256), 0, (hipStream_t)stream, nv, (int)(n - 2 * nv), d_m, d_x, d_y);
  else
    hipLaunchKernelGGL(k_mult_add, dim3(capped_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, d_m, d_x, d_y);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
int wf_dot(int64_t n, const double* d_x, const double* d_y, double* d_result, void* stream)
{
  WF_HIP_CHECK(hipMemsetAsync(d_result, 0, sizeof(double), (hipStream_t)stream));
  if (n <= 0) return WF_OK;
  hipLaunchKernelGGL(k_dot, dim3(reduction_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, n, d_x, d_y, d_result);
  WF_LAUNCH_CHECK();
  return WF_OK;
}
}  // extern "C"

struct wf_boundary {
  int64_t n = 0;
  int32_t nb = 0, n1 = 0, n2 = 0;
  unsigned long long* d_mask = nullptr;
  uint32_t* d_prefix = nullptr;
  double *d_c1 = nullptr, *d_c2 = nullptr;
  // the plain index form as well (first right-hand side of a run, reference-order loop)
  int32_t *d_idx1 = nullptr, *d_idx2 = nullptr;
  double *d_m1 = nullptr, *d_m2 = nullptr;
};

namespace {

int rk4_stage_impl(int64_t n, double bdt, double adt_next, int has_next, double* d_b, const double* d_m, const double* d_vn,
                   const double* d_u_read, const double* d_v_read, double* d_u, double* d_v, const double* d_u0,
                   const double* d_v0, double* d_un, double* d_vn_next, const wf_boundary* bcp, double s1_next, double s2,
                   void* stream)
{
  if (n <= 0) return WF_OK;
  WF_REQUIRE(d_b && d_m && d_vn && d_u_read && d_v_read && d_u && d_v, "wf_rk4_stage: null vector");
  WF_REQUIRE(!has_next || (d_u0 && d_v0 && d_un && d_vn_next), "wf_rk4_stage: next-stage vectors missing");
  WF_REQUIRE(!has_next || d_vn_next != d_vn, "wf_rk4_stage: vn_next must not alias vn");
  WF_REQUIRE(!bcp || bcp->n == n, "wf_rk4_stage_bc: the boundary plan was built for another vector length");
  MarkerScope mk("wf_rk4_stage");
  hipStream_t st = (hipStream_t)stream;
  auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  bool vec = aligned(d_b) && aligned(d_m) && aligned(d_vn) && aligned(d_u_read) && aligned(d_v_read) && aligned(d_u)
             && aligned(d_v);
  if (has_next) vec = vec && aligned(d_u0) && aligned(d_v0) && aligned(d_un) && aligned(d_vn_next);
  wf::StageArgs a{};
  a.bdt = bdt;
  a.adt_next = adt_next;
  a.b = d_b;
  a.m = d_m;
  a.vn = d_vn;
  a.u_read = d_u_read;
  a.v_read = d_v_read;
  a.u_ = d_u;
  a.v_ = d_v;
  a.u0 = d_u0;
  a.v0 = d_v0;
  a.un = d_un;
  a.vn_next = d_vn_next;
  const bool use_bc = bcp && bcp->nb > 0;
  if (use_bc) {
    a.bc.mask = bcp->d_mask;
    a.bc.prefix = bcp->d_prefix;
    a.bc.c1 = bcp->d_c1;
    a.bc.c2 = bcp->d_c2;
    a.bc.s1 = s1_next;
    a.bc.s2 = s2;
  }
  // 16-byte sweep (streaming accesses for everything but un and b: 0.155 -> 0.135 ms at cfg2) with the odd last dof
  // in the same launch, or the scalar kernel for unaligned vectors
  const int64_t nvec = vec ? n / 2 : n;
  const int ntail = vec ? (int)(n - 2 * nvec) : 0;
  const unsigned grid = capped_grid((size_t)std::max<int64_t>(nvec, 1), 256);
#define WF_STAGE(VEC, NEXT, NT_, BC_) \
  hipLaunchKernelGGL((wf::k_rk4_stage<VEC, NEXT, NT_, BC_>), dim3(grid), dim3(256), 0, st, nvec, ntail, a)
  if (vec) {
    if (has_next) { if (use_bc) WF_STAGE(2, true, true, true); else WF_STAGE(2, true, true, false); }
    else          { if (use_bc) WF_STAGE(2, false, true, true); else WF_STAGE(2, false, true, false); }
  } else {
    if (has_next) { if (use_bc) WF_STAGE(1, true, false, true); else WF_STAGE(1, true, false, false); }
    else          { if (use_bc) WF_STAGE(1, false, false, true); else WF_STAGE(1, false, false, false); }
  }
#undef WF_STAGE
  WF_LAUNCH_CHECK();
  return WF_OK;
}

}  // namespace

extern "C" {

int wf_rk4_stage(int64_t n, double bdt, double adt_next, int has_next, double* d_b, const double* d_m,
                 const double* d_vn, const double* d_u_read, const double* d_v_read, double* d_u, double* d_v,
                 const double* d_u0, const double* d_v0, double* d_un, double* d_vn_next, void* stream)
{
  return rk4_stage_impl(n, bdt, adt_next, has_next, d_b, d_m, d_vn, d_u_read, d_v_read, d_u, d_v, d_u0, d_v0, d_un, d_vn_next,
                        nullptr, 0.0, 0.0, stream);
}

int wf_rk4_stage_bc(int64_t n, double bdt, double adt_next, int has_next, double* d_b, const double* d_m,
                    const double* d_vn, const double* d_u_read, const double* d_v_read, double* d_u, double* d_v,
                    const double* d_u0, const double* d_v0, double* d_un, double* d_vn_next, const wf_boundary* bc,
                    double s1_next, double s2, void* stream)
{
  WF_REQUIRE(bc != nullptr, "wf_rk4_stage_bc: null boundary plan");
  return rk4_stage_impl(n, bdt, adt_next, has_next, d_b, d_m, d_vn, d_u_read, d_v_read, d_u, d_v, d_u0, d_v0, d_un, d_vn_next, bc,
                        s1_next, s2, stream);
}

int wf_boundary_destroy(wf_boundary* bc)
{
  if (!bc) return WF_OK;
  (void)hipFree(bc->d_mask);
  (void)hipFree(bc->d_prefix);
  (void)hipFree(bc->d_c1);
  (void)hipFree(bc->d_c2);
  (void)hipFree(bc->d_idx1);
  (void)hipFree(bc->d_idx2);
  (void)hipFree(bc->d_m1);
  (void)hipFree(bc->d_m2);
  delete bc;
  return WF_OK;
}

int wf_boundary_create(int64_t n, int32_t n1, const int32_t* h_idx1, const double* h_m1, int32_t n2, const int32_t* h_idx2,
                       const double* h_m2, wf_boundary** out)
{
  WF_REQUIRE(out && n >= 0 && n1 >= 0 && n2 >= 0, "wf_boundary_create: bad argument");
  *out = nullptr;
  WF_REQUIRE((n1 == 0 || (h_idx1 && h_m1)) && (n2 == 0 || (h_idx2 && h_m2)), "wf_boundary_create: null array");
  for (int32_t i = 0; i < n1; ++i) WF_REQUIRE(h_idx1[i] >= 0 && h_idx1[i] < n, "wf_boundary_create: index out of range");
  for (int32_t i = 0; i < n2; ++i) WF_REQUIRE(h_idx2[i] >= 0 && h_idx2[i] < n, "wf_boundary_create: index out of range");
  const size_t nw = (size_t)((n + 63) / 64) + 1;   // + 1: the stage kernel reads two words per wave
  std::vector<unsigned long long> mask(nw, 0ull);
  for (int32_t i = 0; i < n1; ++i) mask[h_idx1[i] >> 6] |= 1ull << (h_idx1[i] & 63);
  for (int32_t i = 0; i < n2; ++i) mask[h_idx2[i] >> 6] |= 1ull << (h_idx2[i] & 63);
  std::vector<uint32_t> prefix(nw, 0u);
  uint32_t run = 0;
  for (size_t w = 0; w < nw; ++w) {
    prefix[w] = run;
    run += (uint32_t)__builtin_popcountll(mask[w]);
  }
  const int32_t nb = (int32_t)run;
  auto rank = [&](int32_t d) {
    return (size_t)prefix[d >> 6] + (size_t)__builtin_popcountll(mask[d >> 6] & ((1ull << (d & 63)) - 1ull));
  };
  std::vector<double> c1((size_t)nb, 0.0), c2((size_t)nb, 0.0);
  for (int32_t i = 0; i < n1; ++i) c1[rank(h_idx1[i])] += h_m1[i];   // a repeated index accumulates, like wf_boundary_apply
  for (int32_t i = 0; i < n2; ++i) c2[rank(h_idx2[i])] += h_m2[i];
  std::unique_ptr<wf_boundary, int (*)(wf_boundary*)> bc(new wf_boundary, wf_boundary_destroy);
  bc->n = n;
  bc->nb = nb;
  bc->n1 = n1;
  bc->n2 = n2;
  auto up = [](auto** d, const auto* h, size_t cnt) -> int {
    *d = nullptr;
    if (cnt == 0) return WF_OK;
    WF_HIP_CHECK(hipMalloc((void**)d, cnt * sizeof(**d)));
    WF_HIP_CHECK(hipMemcpy(*d, h, cnt * sizeof(**d), hipMemcpyHostToDevice));
    return WF_OK;
  };
  int rc;
  if ((rc = up(&bc->d_mask, mask.data(), nw)) != WF_OK || (rc = up(&bc->d_prefix, prefix.data(), nw)) != WF_OK
      || (rc = up(&bc->d_c1, c1.data(), (size_t)nb)) != WF_OK || (rc = up(&bc->d_c2, c2.data(), (size_t)nb)) != WF_OK
      || (rc = up(&bc->d_idx1, h_idx1, (size_t)n1)) != WF_OK || (rc = up(&bc->d_m1, h_m1, (size_t)n1)) != WF_OK
      || (rc = up(&bc->d_idx2, h_idx2, (size_t)n2)) != WF_OK || (rc = up(&bc->d_m2, h_m2, (size_t)n2)) != WF_OK)
    return rc;
  *out = bc.release();
  return WF_OK;
}

// b[idx1[i]] += s1 m1[i]; b[idx2[i]] += s2 m2[i] v[idx2[i]] from a plan (the first right-hand side of a fused run)
int wf_boundary_apply_plan(const wf_boundary* bc, double s1, double s2, const double* d_v, double* d_b, void* stream)
{
  WF_REQUIRE(bc && d_v && d_b, "wf_boundary_apply_plan: null argument");
  return wf_boundary_apply(bc->n1, bc->d_idx1, bc->d_m1, s1, bc->n2, bc->d_idx2, bc->d_m2, s2, d_v, d_b, stream);
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
