// Range markers for profiles: the roctx counterpart of the reference's nvtxMarkA / cudaProfilerStart
// bracketing (demo/gpu_scatter_mpi/main.cpp:89,101-121).  Off by default; wf_markers_enable(1) binds
// librocprofiler-sdk-roctx at run time (no hard dependency) and from then on wf_op_apply*, the ghost
// exchange and the fused RK4 stage push / pop a named range on the calling host thread, visible in
// `rocprofv3 --marker-trace`.  The flag is one relaxed atomic load per call when off.
#include <dlfcn.h>

#include <atomic>

#include "common.h"

namespace wf {

namespace {
std::atomic<int> g_on{0};
int (*g_push)(const char*) = nullptr;
int (*g_pop)() = nullptr;
void (*g_mark)(const char*) = nullptr;
}  // namespace

bool markers_on() { return g_on.load(std::memory_order_relaxed) != 0; }
void marker_push(const char* name)
{
  if (markers_on()) (void)g_push(name);
}
void marker_pop()
{
  if (markers_on()) (void)g_pop();
}

}  // namespace wf

extern "C" {

int wf_markers_enable(int on)
{
  using namespace wf;
  if (!on) {
    g_on.store(0);
    return WF_OK;
  }
  if (!g_push) {
    void* h = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                             "libroctx64.so.4", "libroctx64.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      set_error("wf_markers_enable: no roctx library (librocprofiler-sdk-roctx / libroctx64) found");
      return WF_ERR_UNSUPPORTED;
    }
    g_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    g_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    g_mark = reinterpret_cast<void (*)(const char*)>(dlsym(h, "roctxMarkA"));
    if (!g_push || !g_pop || !g_mark) {
      g_push = nullptr;
      set_error("wf_markers_enable: roctx library lacks roctxRangePushA / roctxRangePop / roctxMarkA");
      return WF_ERR_UNSUPPORTED;
    }
  }
  g_on.store(1);
  return WF_OK;
}

int wf_marker_push(const char* name)
{
  WF_REQUIRE(name != nullptr, "wf_marker_push: null name");
  wf::marker_push(name);
  return WF_OK;
}

int wf_marker_pop(void)
{
  wf::marker_pop();
  return WF_OK;
}

int wf_marker_mark(const char* name)
{
  WF_REQUIRE(name != nullptr, "wf_marker_mark: null name");
  if (wf::markers_on()) wf::g_mark(name);
  return WF_OK;
}

}  // extern "C"
