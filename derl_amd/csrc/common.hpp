// Shared host-side helpers for the C-ABI translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/derl_amd.h"

// -DDX_DIAG=1 (libderl_amd_diag.so only, derl_amd/build.py): in-kernel stamps and the bisecting
// switches of the ring / weight-gradient kernels (DX_NTP_DIAG, DX_NT_DIAG, DX_WD_DIAG, DX_FC_DIAG,
// DX_NTP_NWG, DX_WD_NWG -- some of them compute WRONG results on purpose).  The product library is
// built without it: the branches are compiled out of its kernels and the variables are never read.
#ifndef DX_DIAG
#define DX_DIAG 0
#endif

namespace dx {

constexpr bool kDiag = DX_DIAG != 0;

char *error_buffer();  // thread-local, 512 bytes

inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define DX_REQUIRE(cond, ...) \
  do { if (!(cond)) return ::dx::fail(DX_EINVAL, __VA_ARGS__); } while (0)

#define DX_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return ::dx::fail(DX_EHIP, "%s failed: %s (%s:%d)", #expr,                  \
                        hipGetErrorString(e_), __FILE__, __LINE__);               \
  } while (0)

// after a kernel launch: catches bad launch configurations without synchronising (and counts the
// launch: dx_launch_count(), how bench.py reports launches per update)
void count_launch();
#define DX_LAUNCH_CHECK()          \
  do {                             \
    ::dx::count_launch();          \
    DX_HIP(hipGetLastError());     \
  } while (0)

// roctx range around a C-ABI call (tracing): DX_ROCTX=1 resolves
// roctxRangePushA / roctxRangePop from libroctx64.so at first use, so that a rocprofv3
// --marker-trace shows which entry point every kernel belongs to.  Off (the default) it is one
// predictable branch per call.
struct TraceRange {
  explicit TraceRange(const char *name);
  ~TraceRange();
  bool active;
};
#define DX_TRACE(name) ::dx::TraceRange dx_trace_range_(name)

// Environment switches (diagnostic switches): read through ONE cache.  A switch is parsed the first time a
// call needs it and again after dx_reload_env() (include/derl_amd.h) -- the values are process-wide on purpose (one
// process per GPU), but nothing else about them is static: a host or a test that changes a switch says so.
// (two host threads may meet in one slot: the value is published before the generation that vouches for it)
struct EnvSlot { std::atomic<int> generation{-1}, value{0}; std::atomic<bool> set{false}; };
int env_generation();
void env_read(EnvSlot &slot, const char *name, int dflt);  // getenv + atoi
inline const EnvSlot &env_cached(EnvSlot &slot, const char *name, int dflt) {
  if (slot.generation.load(std::memory_order_acquire) != env_generation()) env_read(slot, name, dflt);
  return slot;
}
#define DX_ENV(name, dflt) ([]() -> int { static ::dx::EnvSlot slot_; return ::dx::env_cached(slot_, name, dflt).value.load(std::memory_order_relaxed); }())
#define DX_ENV_SET(name) ([]() -> bool { static ::dx::EnvSlot slot_; return ::dx::env_cached(slot_, name, 0).set.load(std::memory_order_relaxed); }())

// Per-device one-time set-up of a call site (the dynamic-LDS opt-in belongs to a DEVICE's code object: a process that
// drives two devices must make it on each), and the CU count of the current device (devices may differ).
struct DeviceFlags { bool seen[64] = {}; };
int lds_opt_in(DeviceFlags &flags, const void *kernel, int bytes);
int device_cus(int *cus_out);
#define DX_LDS_OPT_IN(kernel, bytes)                                                                              \
  do {                                                                                                            \
    static ::dx::DeviceFlags flags_;                                                                              \
    if (int rc_ = ::dx::lds_opt_in(flags_, reinterpret_cast<const void *>(kernel), static_cast<int>(bytes))) return rc_; \
  } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline int cdiv(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace dx
