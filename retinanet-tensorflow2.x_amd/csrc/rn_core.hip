// rn_core.hip — error reporting, ABI version and device probe for librnet_hip.so.
#include "rn_common.h"
#include <string.h>

static thread_local char g_rn_err[512] = "";

void rn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_rn_err, sizeof(g_rn_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* rn_last_error(void) { return g_rn_err; }

extern "C" int rn_abi_version(void) { return 2; }

extern "C" int rn_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 0;
  }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
