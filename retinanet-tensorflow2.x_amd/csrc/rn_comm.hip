// rn_comm.hip — C1-C3 of the data-parallel step over RCCL / xGMI, behind the C ABI:
//   rn_comm_unique_id / rn_comm_init / rn_comm_destroy, rn_allreduce_bucket (C1 gradient buckets),
//   rn_allreduce_small (C2 loss normaliser, C3 SyncBatchNorm [sum | sum of squares] messages).
// The reference reaches its collectives through tf.distribute (replica_context.all_reduce, retinanet_loss.py:46-49;
// SyncBatchNormalization, model/utils.py:10-12; the optimizer's cross-replica sum, executor.py:436-437).  Here a
// collective is ONE ncclAllReduce enqueued on the CALLER'S stream: the ~130 latency-bound SyncBN messages of a step
// stay in program order on the compute stream — no hop to a communication stream and back (two event dependencies per
// message), no per-call Python dispatch through torch.distributed.  xGMI is point-to-point, RCCL picks its low-latency
// protocol for these few-KB messages by itself; the large gradient buckets use the same entry point on their own
// communicator (one communicator per stream that carries collectives, so a bucket never queues in front of a SyncBN
// message).
// librccl is loaded lazily with dlopen: a single-GPU process never touches it, and when PyTorch has already loaded its
// copy (same SONAME) this resolves to that one.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>

#include "rn_common.h"

namespace {
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::mutex g_mu;

bool rccl_load() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_rccl.ok) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    rn_set_error("rn_comm: cannot load librccl (%s)", dlerror());
    return false;
  }
#define RN_SYM(field_, name_)                                                      \
  g_rccl.field_ = (decltype(g_rccl.field_))dlsym(g_rccl.handle, name_);            \
  if (!g_rccl.field_) {                                                            \
    rn_set_error("rn_comm: librccl has no symbol %s", name_);                      \
    return false;                                                                  \
  }
  RN_SYM(GetUniqueId, "ncclGetUniqueId")
  RN_SYM(CommInitRank, "ncclCommInitRank")
  RN_SYM(CommDestroy, "ncclCommDestroy")
  RN_SYM(AllReduce, "ncclAllReduce")
  RN_SYM(GetErrorString, "ncclGetErrorString")
#undef RN_SYM
  g_rccl.ok = true;
  return true;
}

struct RnComm {
  ncclComm_t comm;
  int rank, world;
};
}  // namespace

#define RN_CHECK_NCCL(expr)                                                                          \
  do {                                                                                               \
    ncclResult_t r_ = (expr);                                                                        \
    if (r_ != ncclSuccess) {                                                                         \
      rn_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_));          \
      return RN_ECOMM;                                                                               \
    }                                                                                                \
  } while (0)

extern "C" int rn_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

// Local, non-collective: 1 when librccl and every entry point this file uses can be loaded in this process.  Callers
// agree on it (a MIN over the job) BEFORE any rank enters the collective rn_comm_init, so that a rank whose librccl is
// missing cannot leave the others waiting inside ncclCommInitRank.
extern "C" int rn_comm_available(void) { return rccl_load() ? 1 : 0; }

extern "C" int rn_comm_unique_id(void* out) {
  RN_CHECK_ARG(out != nullptr, "rn_comm_unique_id: null output");
  if (!rccl_load()) return RN_ECOMM;
  ncclUniqueId id;
  RN_CHECK_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return RN_OK;
}

// Collective: every rank of the job calls it with the id rank 0 generated (the caller's HIP device is the rank's GPU).
extern "C" int rn_comm_init(const void* unique_id, int rank, int world, void** comm_out) {
  RN_CHECK_ARG(unique_id && comm_out && world >= 1 && rank >= 0 && rank < world, "rn_comm_init: bad argument");
  if (!rccl_load()) return RN_ECOMM;
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  RnComm* c = new RnComm();
  c->rank = rank;
  c->world = world;
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    rn_set_error("rn_comm_init: ncclCommInitRank(rank %d of %d) -> %s", rank, world, g_rccl.GetErrorString(r));
    delete c;
    return RN_ECOMM;
  }
  *comm_out = c;
  return RN_OK;
}

extern "C" int rn_comm_destroy(void* comm) {
  if (!comm) return RN_OK;
  RnComm* c = (RnComm*)comm;
  if (g_rccl.ok) g_rccl.CommDestroy(c->comm);
  delete c;
  return RN_OK;
}

// In-place SUM over the ranks of `count` elements at `ptr` (device), enqueued on `stream`.
extern "C" int rn_allreduce_bucket(void* comm, void* ptr, int64_t count, int dtype, void* stream) {
  RN_CHECK_ARG(comm && ptr && count > 0, "rn_allreduce_bucket: bad argument");
  RN_CHECK_ARG(dtype == RN_DT_F32 || dtype == RN_DT_BF16, "rn_allreduce_bucket: dtype must be RN_DT_F32 or RN_DT_BF16");
  RnComm* c = (RnComm*)comm;
  RN_CHECK_NCCL(g_rccl.AllReduce(ptr, ptr, (size_t)count, dtype == RN_DT_F32 ? ncclFloat32 : ncclBfloat16, ncclSum, c->comm,
                                 (hipStream_t)stream));
  return RN_OK;
}

// The few-KB fp32 messages (SyncBatchNorm sums, the loss normaliser): same collective, named apart so that the
// latency path can change underneath (RCCL's LL protocol today) without touching the callers.
extern "C" int rn_allreduce_small(void* comm, float* ptr, int count, void* stream) {
  RN_CHECK_ARG(comm && ptr && count > 0, "rn_allreduce_small: bad argument");
  RnComm* c = (RnComm*)comm;
  RN_CHECK_NCCL(g_rccl.AllReduce(ptr, ptr, (size_t)count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
  return RN_OK;
}
