// Microbenchmark (diagnostic, not part of the library): sustained HBM read rate of MI355X for the
// access shape of the stiffness operator's geometry stream -- 16-byte non-temporal loads, 256-thread
// workgroups, several loads in flight per thread -- over a 1.2 GB array (larger than the 256 MB
// Infinity Cache), and the same with a 1/5 write stream mixed in (the operator's y traffic).
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_read_rate.hip -o examples/bin/hbm_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2v __attribute__((ext_vector_type(2)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const d2v* __restrict__ a, size_t n, double* __restrict__ out)
{
  double s = 0.0;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + 256 * (U - 1) < n; i += stride) {
    d2v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + 256 * u) : a[i + 256 * u];
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
  }
  if (s == 1.2345e300) out[0] = s;
}

template <int U>
__global__ __launch_bounds__(256) void k_read_write(const d2v* __restrict__ a, size_t n, d2v* __restrict__ w)
{
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + 256 * (U - 1) < n; i += stride) {
    d2v v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + i + 256 * u);
    d2v s = v[0];
#pragma unroll
    for (int u = 1; u < U; ++u) s += v[u];
    w[i / U] = s;   // one 16-byte store per U loads
  }
}

template <typename F>
static double time_ms(F&& f)
{
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  float best = 1e30f;
  for (int r = 0; r < 10; ++r) {
    (void)hipEventRecord(e0, nullptr);
    f();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

int main()
{
  const size_t bytes = (size_t)1200 << 20, n = bytes / 16;
  d2v *a, *w;
  double* out;
  (void)hipMalloc(&a, bytes);
  (void)hipMalloc(&w, bytes / 4);
  (void)hipMalloc(&out, 8);
  (void)hipMemset(a, 0, bytes);
  for (int wgs : {512, 1024, 2048, 4096, 16384}) {
    double t4 = time_ms([&] { hipLaunchKernelGGL((k_read<4, true>), dim3(wgs), dim3(256), 0, nullptr, a, n, out); });
    double t8 = time_ms([&] { hipLaunchKernelGGL((k_read<8, true>), dim3(wgs), dim3(256), 0, nullptr, a, n, out); });
    double t8t = time_ms([&] { hipLaunchKernelGGL((k_read<8, false>), dim3(wgs), dim3(256), 0, nullptr, a, n, out); });
    double trw = time_ms([&] { hipLaunchKernelGGL((k_read_write<4>), dim3(wgs), dim3(256), 0, nullptr, a, n, w); });
    std::printf("workgroups %5d: read nt x4 %.0f GB/s, read nt x8 %.0f GB/s, read temporal x8 %.0f GB/s, read + 1/4 write %.0f GB/s (traffic)\n",
                wgs, bytes / t4 / 1e6, bytes / t8 / 1e6, bytes / t8t / 1e6, bytes * 1.25 / trw / 1e6);
  }
  return 0;
}
