// k_mass_march<P, BX, BY>: dense (sum-factorised) mass operator y += Phi^T diag(det J w) Phi x with a square
// 1-D table Phi = phi1 (x) phi1 (x) phi1 on the lattice columns of ANY dofmap -- MassOperator::apply,
// common/cuda/mass.hpp:76-95 + mass_kernel.cu:5-37, the DGEMM pair of demo/gpu_operator/main.cpp:144-160
// (k >> m ~ n), marching through the columns like the stiffness kernels (plan of generic_plan.cpp).
//
// What differs from k_march_idx<OP_MASS> (stiffness_march_idx.hip), which it replaces for the non-collocated
// rules (the collocated ones are a diagonal, api.hip):
//  * a cell lives inside ONE wave (n^2 = (P+1)^2 lanes; floor(64 / n^2) cells per wave), so the five passes of
//    the element kernel exchange their data through wave-private LDS scratch with no workgroup barrier --
//    LDS operations of one wave execute in order;
//  * every pass is a "pencil" contraction in registers: a lane reads the n values of one line of the cell,
//    multiplies by the 1-D table held in VGPRs (n^2 doubles, the same in every lane: as scalar operands
//    they would need 2 n^2 SGPRs) and writes n values -- 2 n LDS accesses per lane and pass instead of
//    the n^2 reads of a column thread that fetches a whole row for every output:
//        X  : lane (j, k)   A[k][j][qi]   = sum_i  phi[qi][i] U[k][j][i]
//        Y  : lane (qi, k)  B[k][qj][qi]  = sum_j  phi[qj][j] A[k][j][qi]
//        Z  : lane (qi, qj) w[qk] = detJ[qk] sum_k phi[qk][k] B[k][qj][qi];  A[k][qj][qi] = sum_qk phi[qk][k] w[qk]
//        Y^T: lane (qi, k)  B[k][j][qi]   = sum_qj phi[qj][j] A[k][qj][qi]
//        X^T: lane (j, k)   out[k][j][i]  = sum_qi phi[qi][i] B[k][j][qi]
//  * the x planes, the per-cell results and the carried z-shared plane are double-buffered in LDS: ONE
//    workgroup barrier per layer (between writing a layer's results / the next layer's x planes and
//    flushing them); the flush of layer l runs beside the passes of layer l + 1.
// HBM-bound by its bytes (8 B of det J w per point + x + y), latency-bound in practice.
#include "stiffness_core.h"

namespace wf {

template <int P, int BX, int BY>
struct MassLayout {
  static constexpr int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY;
  static constexpr int CW = 64 / n2;                      // cells per wave
  static constexpr int NWV = (CB + CW - 1) / CW;          // waves per workgroup
  static constexpr int WG = 64 * NWV;
  static constexpr int TX = P * BX + 1, TY = P * BY + 1, TP = TX * TY;
  static constexpr int oUx = 0;                           // [2][(P + 1) TP]
  static constexpr int oO = oUx + 2 * (P + 1) * TP;       // [2][CB P n2]
  static constexpr int oCy = oO + 2 * CB * P * n2;        // [2][CB n2]
  static constexpr int oA = oCy + 2 * CB * n2;            // [CB nd] x 2 (A, B), wave-private per cell
  static constexpr int ndoubles = ((oA + 2 * CB * nd + 1) / 2) * 2;
  static_assert(CW >= 1, "a cell does not fit a wave");
};

size_t mass_march_lds_bytes(int P, int BX, int BY, int lz)
{
  const int n = P + 1, n2 = n * n, nd = n * n2, CB = BX * BY, TP = (P * BX + 1) * (P * BY + 1);
  const size_t d = (size_t)2 * (P + 1) * TP + (size_t)2 * CB * P * n2 + (size_t)2 * CB * n2 + (size_t)2 * CB * nd + 2;
  return d * sizeof(double) + (size_t)(P * lz + 1) * TP * sizeof(int32_t);
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
  int32_t* sIdx = reinterpret_cast<int32_t*>(ms_smem + L::ndoubles);   // [(P nl + 1)][TP] dof offsets, -1 = none

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cw = lane / n2, pq = lane % n2, p0 = pq % n, p1 = pq / n;
  const int cl = wave * CW + cw;                       // cell of the layer
  const bool active = cw < CW && cl < CB;
  const int lx = cl % BX, ly = cl / BX;
  double* A = ms_smem + L::oA + (active ? cl : 0) * nd;
  double* B = A + CB * nd;
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
    for (int k = 0; k < n; ++k) d[k] = __builtin_nontemporal_load(dp + (size_t)k * (CB * n2));
  };

  // ---- prologue ---------------------------------------------------------------------------
  {
    const int32_t* __restrict__ pat = a.pat_off + (size_t)a.item_pattern[item] * a.tile_size;
    for (int e = t; e < (P * nl + 1) * TP; e += WG) sIdx[e] = pat[e];
  }
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
  __syncthreads();

  const int ucell = (P * ly) * TX + P * lx;

  auto flush = [&](const double* Ob, int l) {
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      const int pos = t + WG * m;
      if (pos >= P * TP) continue;
      const int32_t off = sIdx[(P * l) * TP + pos];
      if (off < 0) continue;
      const int pl = pos / TP, r = pos % TP, J = r / TX, I = r % TX;
      const int ca = I / P, ia = I % P, cb = J / P, jb = J % P;
      double v = 0.0;
      if (cb < BY) {
        if (ca < BX) v += Ob[((cb * BX + ca) * P + pl) * n2 + jb * n + ia];
        if (ia == 0 && ca > 0) v += Ob[((cb * BX + ca - 1) * P + pl) * n2 + jb * n + P];
      }
      if (jb == 0 && cb > 0) {
        if (ca < BX) v += Ob[(((cb - 1) * BX + ca) * P + pl) * n2 + P * n + ia];
        if (ia == 0 && ca > 0) v += Ob[(((cb - 1) * BX + ca - 1) * P + pl) * n2 + P * n + P];
      }
      unsafeAtomicAdd(a.y + gbase + off, v);
    }
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

  auto layer = [&](double (&dcur)[n], double (&dnext)[n], int l, int b) {
    const bool has_next = l + 1 < nl;
    const double* Ub = Ux + b * (P + 1) * TP;
    double* Un = Ux + (b ^ 1) * (P + 1) * TP;
    double* Ob = O + b * (CB * P * n2);
    const int ln = has_next ? l + 1 : l;
    // (a) next layer's x planes and det J: in flight during the passes (unconditional loads on clamped addresses)
    double xn[NPOS];
#pragma unroll
    for (int m = 0; m < NPOS; ++m) {
      const int pos = t + WG * m;
      const int32_t off = pos < P * TP ? sIdx[(P * ln + 1) * TP + pos] : -1;
      xn[m] = a.x[gbase + (off >= 0 ? off : 0)];
    }
    if (has_next) load_d(dnext, ln);

    // (b) the five passes of the element kernel, wave-private
    if (active) {
      double in[n], out[n];
      // X: lane (j, k) = (p0, p1)
#pragma unroll
      for (int c = 0; c < n; ++c) in[c] = Ub[ucell + p1 * TP + p0 * TX + c];
      fwd(in, out);
#pragma unroll
      for (int q = 0; q < n; ++q) A[(p1 * n + p0) * n + q] = out[q];
      wave_sync();
      // Y: lane (qi, k) = (p0, p1)
#pragma unroll
      for (int c = 0; c < n; ++c) in[c] = A[(p1 * n + c) * n + p0];
      fwd(in, out);
#pragma unroll
      for (int q = 0; q < n; ++q) B[(p1 * n + q) * n + p0] = out[q];
      wave_sync();
      // Z: lane (qi, qj) = (p0, p1): forward, times det J w, transposed
#pragma unroll
      for (int c = 0; c < n; ++c) in[c] = B[(c * n + p1) * n + p0];
      fwd(in, out);
#pragma unroll
      for (int q = 0; q < n; ++q) out[q] *= dcur[q];
      bwd(out, in);
#pragma unroll
      for (int c = 0; c < n; ++c) A[(c * n + p1) * n + p0] = in[c];
      wave_sync();
      // Y^T: lane (qi, k) = (p0, p1)
#pragma unroll
      for (int q = 0; q < n; ++q) in[q] = A[(p1 * n + q) * n + p0];
      bwd(in, out);
#pragma unroll
      for (int c = 0; c < n; ++c) B[(p1 * n + c) * n + p0] = out[c];
      wave_sync();
      // X^T: lane (j, k) = (p0, p1); planes 0..P-1 -> O, plane P -> carry, plane 0 picks up the previous carry
#pragma unroll
      for (int q = 0; q < n; ++q) in[q] = B[(p1 * n + p0) * n + q];
      bwd(in, out);
      if (p1 == 0) {
#pragma unroll
        for (int c = 0; c < n; ++c) out[c] += Cy[(b ^ 1) * (CB * n2) + cl * n2 + p0 * n + c];
      }
      double* dst = p1 < P ? Ob + (cl * P + p1) * n2 + p0 * n : Cy + b * (CB * n2) + cl * n2 + p0 * n;
#pragma unroll
      for (int c = 0; c < n; ++c) dst[c] = out[c];
    }
    // (c) x planes of the next layer -> the other buffer (the consumer of xn)
    if (has_next) {
#pragma unroll
      for (int m = 0; m < NCP; ++m) {
        const int pos = t + WG * m;
        if (pos < TP) Un[pos] = Ub[P * TP + pos];
      }
#pragma unroll
      for (int m = 0; m < NPOS; ++m) {
        const int pos = t + WG * m;
        if (pos < P * TP) Un[TP + pos] = sIdx[(P * ln + 1) * TP + pos] >= 0 ? xn[m] : 0.0;
      }
    }
    __syncthreads();   // the one workgroup barrier of the layer
    // (d) flush: runs beside the next layer's passes
    flush(Ob, l);
  };
  for (int l = 0; l < nl; l += 2) {
    layer(dA, dB, l, 0);
    if (l + 1 < nl) layer(dB, dA, l + 1, 1);
  }

  // ---- epilogue: the last (carried) plane ---------------------------------------------------
  {
    const double* Cb = Cy + ((nl - 1) & 1) * (CB * n2);
#pragma unroll
    for (int m = 0; m < NCP; ++m) {
      const int pos = t + WG * m;
      if (pos >= TP) continue;
      const int32_t off = sIdx[(P * nl) * TP + pos];
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

// column cross-sections: whole cells per wave (floor(64 / n^2)), four waves
#define WF_MASS_SHAPES(X) X(1, 8, 8) X(2, 7, 4) X(3, 4, 4) X(4, 4, 2) X(5, 2, 2) X(6, 2, 2) X(7, 2, 2)

void mass_march_shape(int P, int* bx, int* by)
{
#define X(PP, BXX, BYY) \
  if (P == PP) {        \
    *bx = BXX;          \
    *by = BYY;          \
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
