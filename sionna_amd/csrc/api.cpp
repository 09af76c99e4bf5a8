// Library-wide pieces of the C-ABI: error string, version, device count.
#include "common.h"

namespace samd {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace samd

extern "C" const char* samd_last_error(void) { return samd::g_last_error.c_str(); }
extern "C" int samd_version(void) { return 100; }
extern "C" int samd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return SAMD_ERR_HIP;
  return n;
}
