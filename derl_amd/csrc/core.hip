// ABI version, error string and device probe.
#include "common.hpp"
#include <cstring>

namespace dx {
char *error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}
}  // namespace dx

extern "C" int dx_abi_version(void) { return DX_ABI_VERSION; }

extern "C" const char *dx_last_error(void) { return dx::error_buffer(); }

extern "C" int dx_device_info(int device, char *name_host, int *cu_count, int *lds_bytes) {
  DX_REQUIRE(name_host != nullptr, "dx_device_info: name_host is NULL");
  hipDeviceProp_t prop;
  DX_HIP(hipGetDeviceProperties(&prop, device));
  std::snprintf(name_host, 256, "%s (%s)", prop.name, prop.gcnArchName);
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = static_cast<int>(prop.sharedMemPerBlock);
  return DX_OK;
}
