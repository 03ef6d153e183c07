// Diagnostic (not part of the library): per-wave timeline of the TSMM kernel at the reference
// shape of demo/gpu_tsmm (100 000 x 125 . 125 x 125).  Prints, in microseconds relative to the
// earliest wave start: prologue end, and the end of each cell tile, as min / median / max over waves.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iwave_fenics_amd/csrc tools/tsmm_trace.hip -o examples/bin/tsmm_trace
#define WF_TSMM_TRACE 1
#include "../wave_fenics_amd/csrc/tsmm.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

namespace wf {
void set_error(const std::string& msg) { std::fprintf(stderr, "%s\n", msg.c_str()); }
}

int main(int argc, char** argv)
{
  const int64_t ncells = argc > 1 ? std::atoll(argv[1]) : 100000;
  const int nd = argc > 2 ? std::atoi(argv[2]) : 125;
  const int layout = argc > 3 ? std::atoi(argv[3]) : 0;
  double *in, *out, *phi;
  hipMalloc(&in, ncells * nd * 8);
  hipMalloc(&out, ncells * nd * 8);
  hipMalloc(&phi, nd * nd * 8);
  {   // random operands: all-zero data lets the MFMA pipe clock higher than real data does
    std::vector<double> h((size_t)ncells * nd);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0) - 0.5; };
    for (auto& v : h) v = rnd();
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    h.resize((size_t)nd * nd);
    for (auto& v : h) v = rnd();
    hipMemcpy(phi, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 8; ++rep) {
    hipEventRecord(e0, nullptr);
    wf_tsmm(layout, ncells, nd, nd, in, phi, out, nullptr);
    wf_tsmm(layout, ncells, nd, nd, out, phi, in, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("pair of products: %.4f ms\n", ms);
  }
  const int NWV = 256 * wf::kTsmmWaves;
  std::vector<unsigned long long> tr(NWV * 8);
  void* sym;
  hipGetSymbolAddress(&sym, HIP_SYMBOL(wf::g_tsmm_trace));
  hipMemset(sym, 0, tr.size() * 8);
  wf_tsmm(layout, ncells, nd, nd, in, phi, out, nullptr);
  hipDeviceSynchronize();
  hipMemcpy(tr.data(), sym, tr.size() * 8, hipMemcpyDeviceToHost);
  {   // shader clock during the kernel: clock64 ticks per wall_clock64 tick (100 MHz) on wave 0 of workgroup 100
    std::vector<unsigned long long> ck(NWV * 8);
    void* sym2;
    hipGetSymbolAddress(&sym2, HIP_SYMBOL(wf::g_tsmm_trace_clk));
    hipMemcpy(ck.data(), sym2, ck.size() * 8, hipMemcpyDeviceToHost);
    const int w = 100 * wf::kTsmmWaves;
    int last = 1;
    while (last + 1 < 8 && tr[w * 8 + last + 1]) ++last;
    std::printf("shader clock between prologue end and last tile end: %.0f MHz\n",
                100.0 * (double)(ck[w * 8 + last] - ck[w * 8 + 1]) / (double)(tr[w * 8 + last] - tr[w * 8 + 1]));
  }
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < NWV; ++w)
    if (tr[w * 8]) t0 = std::min(t0, tr[w * 8]);
  const char* names[8] = {"wave start", "prologue end", "tile 1 end", "tile 2 end", "tile 3 end", "tile 4 end", "tile 5 end", "tile 6 end"};
  for (int i = 0; i < 8; ++i) {
    std::vector<double> v;
    for (int w = 0; w < NWV; ++w)
      if (tr[w * 8 + i]) v.push_back((tr[w * 8 + i] - t0) * 0.01);
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    std::printf("%-13s waves %4zu  min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us\n", names[i], v.size(), v.front(),
                v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  }
  // per-tile durations of wave 0 of a few workgroups
  for (int b : {0, 13, 14, 100, 255}) {
    std::printf("wg %3d:", b);
    for (int w = 0; w < 8; w += 4) {
      std::printf("  wave %d:", w);
      for (int i = 0; i < 7; ++i)
        if (tr[(b * wf::kTsmmWaves + w) * 8 + i]) std::printf(" %.2f", (tr[(b * wf::kTsmmWaves + w) * 8 + i] - t0) * 0.01);
    }
    std::printf("\n");
  }
  return 0;
}
