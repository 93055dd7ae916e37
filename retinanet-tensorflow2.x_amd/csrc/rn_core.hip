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

extern "C" int rn_abi_version(void) { return 3; }

// 0: bfloat16 storage (librnet_hip.so), 1: IEEE half (librnet_hip_f16.so, built with -DRN_F16)
extern "C" int rn_storage_dtype(void) {
#ifdef RN_F16
  return 1;
#else
  return 0;
#endif
}

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

// Compute units the persistent kernels leave free (data-parallel runs): RCCL's kernels need CUs of their own.  A
// persistent 256-row conv / wgrad workgroup owns its CU (130-160 KB of LDS, ~220 VGPRs per wave), so a collective that
// arrives while such a kernel covers every CU waits for a whole tile loop to end (hundreds of microseconds) — ~130
// latency-bound SyncBatchNorm all-reduces per step run beside the weight-gradient kernels of the second stream, and the
// gradient buckets beside the convolutions.  With n CUs kept free, the persistent grids are num_cu - n workgroups.
static int g_rn_reserved_cus = 0;
extern "C" int rn_set_reserved_cus(int n) {
  if (n < 0 || n > 128) {
    rn_set_error("rn_set_reserved_cus: %d out of range (0..128)", n);
    return RN_EINVAL;
  }
  g_rn_reserved_cus = n;
  return RN_OK;
}
int rn_persistent_grid(int work_items, int num_cu) {
  int g = num_cu - g_rn_reserved_cus;
  if (g < 1) g = 1;
  return work_items < g ? work_items : g;
}
int rn_reserved_cus() { return g_rn_reserved_cus; }
