// rn_core.hip — error reporting, ABI version and device probe for librnet_hip.so.
#include "rn_common.h"
#include <string.h>
#include <new>

static thread_local char g_rn_err[512] = "";

void rn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_rn_err, sizeof(g_rn_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* rn_last_error(void) { return g_rn_err; }

extern "C" int rn_abi_version(void) { return 8; }

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

// ---- per-device handle ---------------------------------------------------------------------
// No mutable process-wide state: launch options travel in the problem descriptors (rn_launch_opts); the handle keeps
// an engine's defaults, its device and the communicators it created.
struct rn_handle {
  int device;
  int num_cus;
  rn_launch_opts opts;
  void* comm[4];
};

static int rn_check_opts(const rn_launch_opts& o, const char* who) {
  RN_CHECK_ARG(o.conv_tile >= 0 && o.conv_tile <= 3, "%s: conv_tile=%d (0..3)", who, o.conv_tile);
  RN_CHECK_ARG(o.wgrad_kernel >= 0 && o.wgrad_kernel <= 3, "%s: wgrad_kernel=%d (0..3)", who, o.wgrad_kernel);
  RN_CHECK_ARG(o.reserved_cus >= 0 && o.reserved_cus <= 128, "%s: reserved_cus=%d out of range (0..128)", who,
               o.reserved_cus);
  RN_CHECK_ARG(o.max_workgroups >= 0 && o.conv_big_min_tiles >= 0 && o.wgrad_target_blocks >= 0 && o.splitk_target_blocks >= 0,
               "%s: negative max_workgroups / conv_big_min_tiles / wgrad_target_blocks / splitk_target_blocks", who);
  return RN_OK;
}
int rn_validate_launch_opts(const rn_launch_opts& o, const char* who) { return rn_check_opts(o, who); }

extern "C" int rn_create(int device_id, rn_handle** out) {
  RN_CHECK_ARG(out != nullptr, "rn_create: null out");
  *out = nullptr;
  int n = 0;
  RN_CHECK_HIP(hipGetDeviceCount(&n));
  RN_CHECK_ARG(device_id >= 0 && device_id < n, "rn_create: device %d of %d", device_id, n);
  hipDeviceProp_t p;
  RN_CHECK_HIP(hipGetDeviceProperties(&p, device_id));
  RN_CHECK_ARG(strncmp(p.gcnArchName, "gfx950", 6) == 0, "rn_create: device %d is %s, this library is gfx950 only",
               device_id, p.gcnArchName);
  rn_handle* h = new (std::nothrow) rn_handle();
  if (!h) {
    rn_set_error("rn_create: out of host memory");
    return RN_ENOMEM;
  }
  h->device = device_id;
  h->num_cus = p.multiProcessorCount;
  memset(&h->opts, 0, sizeof(h->opts));
  for (auto& c : h->comm) c = nullptr;
  *out = h;
  return RN_OK;
}

extern "C" int rn_destroy(rn_handle* h) {
  if (!h) return RN_OK;
  int rc = RN_OK;
  for (auto& c : h->comm)
    if (c) {
      const int r = rn_comm_destroy(c);
      if (r != RN_OK) rc = r;
      c = nullptr;
    }
  delete h;
  return rc;
}

extern "C" int rn_handle_device(const rn_handle* h) { return h ? h->device : RN_EINVAL; }
extern "C" int rn_handle_num_cus(const rn_handle* h) { return h ? h->num_cus : RN_EINVAL; }

extern "C" int rn_handle_set_launch_opts(rn_handle* h, const rn_launch_opts* opts) {
  RN_CHECK_ARG(h && opts, "rn_handle_set_launch_opts: null argument");
  const int rc = rn_check_opts(*opts, "rn_handle_set_launch_opts");
  if (rc != RN_OK) return rc;
  h->opts = *opts;
  return RN_OK;
}

extern "C" int rn_handle_get_launch_opts(const rn_handle* h, rn_launch_opts* opts) {
  RN_CHECK_ARG(h && opts, "rn_handle_get_launch_opts: null argument");
  *opts = h->opts;
  return RN_OK;
}

extern "C" int rn_handle_comm_init(rn_handle* h, int slot, const void* unique_id, int rank, int world) {
  RN_CHECK_ARG(h && slot >= 0 && slot < 4, "rn_handle_comm_init: bad handle / slot %d (0..3)", slot);
  RN_CHECK_ARG(h->comm[slot] == nullptr, "rn_handle_comm_init: slot %d already holds a communicator", slot);
  return rn_comm_init(unique_id, rank, world, &h->comm[slot]);
}

extern "C" int rn_handle_comm_destroy(rn_handle* h, int slot) {
  RN_CHECK_ARG(h && slot >= 0 && slot < 4, "rn_handle_comm_destroy: bad handle / slot %d (0..3)", slot);
  if (!h->comm[slot]) return RN_OK;
  void* c = h->comm[slot];
  h->comm[slot] = nullptr;
  return rn_comm_destroy(c);
}

extern "C" void* rn_handle_comm(const rn_handle* h, int slot) {
  return (h && slot >= 0 && slot < 4) ? h->comm[slot] : nullptr;
}

// Persistent grids: a persistent 256-row conv / wgrad workgroup owns its CU (130-160 KB of LDS, ~220 VGPRs per wave), so a
// collective that arrives while such a kernel covers every CU waits for a whole tile loop to end (hundreds of
// microseconds) — ~130 latency-bound SyncBatchNorm all-reduces per step run beside the weight-gradient kernels of the
// second stream, and the gradient buckets beside the convolutions.  With opts.reserved_cus CUs kept free the grids
// are num_cu - reserved workgroups.
int rn_persistent_grid(int work_items, int num_cu, const rn_launch_opts& o) {
  int g = num_cu - o.reserved_cus;
  if (o.max_workgroups > 0 && o.max_workgroups < g) g = o.max_workgroups;
  if (g < 1) g = 1;
  return work_items < g ? work_items : g;
}

int rn_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  return dev & 63;
}

int rn_num_cus() {   // of the current device; cached per device id
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cus[dev]) {
    hipDeviceProp_t prop;
    cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}
