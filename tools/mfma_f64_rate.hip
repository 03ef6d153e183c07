// Microbenchmark (diagnostic, not part of the library): sustained rate of
// v_mfma_f64_16x16x4_f64 on MI355X as a function of independent accumulator chains per wave
// and waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, int iters, double a0, double b0)
{
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345) out[0] = s;
}
template <int NACC>
void run(int waves_per_simd)
{
  double* d;
  hipMalloc(&d, 8);
  const int iters = 20000;
  const int threads = 256 * waves_per_simd;   // one workgroup per CU, 4 SIMDs
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, 100, 1.0, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 256.0 * 4 * waves_per_simd;
  printf("acc chains %d, waves/SIMD %d: %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", NACC, waves_per_simd,
         flop / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)NACC * iters * waves_per_simd));
  hipFree(d);
}
int main()
{
  for (int w = 1; w <= 2; ++w) {
    run<1>(w);
    run<2>(w);
    run<4>(w);
    run<8>(w);
  }
  return 0;
}
