// Library-level entry points of libpfpp_hip.so: version, error text, device query.
#include <stdarg.h>
#include <string.h>

#include "pfpp_common.h"

namespace pfpp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return PFPP_EHIP;
  }
  return PFPP_OK;
}

}  // namespace pfpp

extern "C" int pfpp_version(void) { return 1; }

extern "C" const char* pfpp_last_error(void) { return pfpp::g_err; }

extern "C" int pfpp_device_cu_count(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  return prop.multiProcessorCount;
}
