// ABI version, error string and device probe.
#include "common.hpp"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>

namespace dx {
namespace {
using PushFn = int (*)(const char *);
using PopFn = int (*)();
PushFn g_push = nullptr;
PopFn g_pop = nullptr;
bool trace_enabled() {
  static int state = -1;  // -1 unknown, 0 off, 1 on
  if (state < 0) {
    state = 0;
    const char *e = getenv("DX_ROCTX");
    if (e && atoi(e) != 0) {
      void *h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
      if (h) {
        g_push = reinterpret_cast<PushFn>(dlsym(h, "roctxRangePushA"));
        g_pop = reinterpret_cast<PopFn>(dlsym(h, "roctxRangePop"));
        state = (g_push && g_pop) ? 1 : 0;
      }
    }
  }
  return state == 1;
}
}  // namespace
TraceRange::TraceRange(const char *name) : active(trace_enabled()) {
  if (active) g_push(name);
}
TraceRange::~TraceRange() {
  if (active) g_pop();
}

namespace {
std::atomic<long long> g_launches{0};
std::atomic<int> g_env_generation{0};
}
int lds_opt_in(DeviceFlags &flags, const void *kernel, int bytes) {
  int dev = 0;
  DX_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && flags.seen[dev]) return DX_OK;
  DX_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (dev >= 0 && dev < 64) flags.seen[dev] = true;
  return DX_OK;
}
int device_cus(int *cus_out) {
  static int table[64] = {};
  int dev = 0;
  DX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || table[dev] == 0) {
    int cus = 0;
    DX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev >= 0 && dev < 64) table[dev] = cus;
    *cus_out = cus;
    return DX_OK;
  }
  *cus_out = table[dev];
  return DX_OK;
}
int env_generation() { return g_env_generation.load(std::memory_order_relaxed); }
void env_read(EnvSlot &slot, const char *name, int dflt) {
  const int generation = env_generation();  // (read BEFORE the variable: a reload in between is seen at the next use)
  const char *e = getenv(name);
  slot.set.store(e != nullptr, std::memory_order_relaxed);
  slot.value.store(e ? atoi(e) : dflt, std::memory_order_relaxed);
  slot.generation.store(generation, std::memory_order_release);
}
void count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

char *error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}
}  // namespace dx

extern "C" int dx_abi_version(void) { return DX_ABI_VERSION; }

extern "C" const char *dx_last_error(void) { return dx::error_buffer(); }

extern "C" int dx_reload_env(void) {
  dx::g_env_generation.fetch_add(1, std::memory_order_relaxed);
  return DX_OK;
}

extern "C" long long dx_launch_count(void) { return dx::g_launches.load(std::memory_order_relaxed); }

extern "C" int dx_device_info(int device, char *name_host, int *cu_count, int *lds_bytes) {
  DX_TRACE("dx_device_info");
  DX_REQUIRE(name_host != nullptr, "dx_device_info: name_host is NULL");
  hipDeviceProp_t prop;
  DX_HIP(hipGetDeviceProperties(&prop, device));
  std::snprintf(name_host, 256, "%s (%s)", prop.name, prop.gcnArchName);
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = static_cast<int>(prop.sharedMemPerBlock);
  return DX_OK;
}
