// The data-parallel gradient exchange behind the C-ABI: sum-all-reduce of contiguous ranges ("buckets") of the flat gradient
// buffer over RCCL (xGMI), on a side stream, overlapped with the rest of backward -- what replaces DistributedDataParallel's
// reducer (reference dlib/models/model_base.py:135-142) for a caller that is not PyTorch.  The Python host side keeps using
// torch.distributed (backend "nccl" = the same RCCL) for its rendezvous; these entry points are the same exchange for a C /
// C++ / ctypes caller that brings its own: one rank makes the 128-byte id, every rank gets it by the caller's means (a file, a
// socket, MPI), every rank opens its communicator on its current device.
//
// RCCL is NOT a link-time dependency of libsrhip.so: the five functions are resolved with dlopen at the first call -- the
// RCCL already mapped into the process if there is one (PyTorch bundles its own librccl.so: two copies of RCCL in one process
// would each claim the same GPUs), else librccl.so.1 of the ROCm installation.  A process that never calls
// srhip_allreduce_* never loads it.
#include <dlfcn.h>
#include <new>
#include <rccl/rccl.h>
#include "common.h"

namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  void* handle = nullptr;
  const char* error = nullptr;
};
RcclApi g_api;

void load_rccl() {
  static const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)                       // a copy already mapped into the process first (RTLD_NOLOAD)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) { g_api.error = "librccl.so not found (dlopen)"; return; }
  g_api.handle = h;
  g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
  g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_api.GetUniqueId || !g_api.CommInitRank || !g_api.AllReduce || !g_api.CommDestroy || !g_api.GetErrorString)
    g_api.error = "librccl.so lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy / ncclGetErrorString";
}

int need_rccl(const char* who) {
  static const bool once = (load_rccl(), true);       // thread-safe one-time initialisation (C++11 local static)
  (void)once;
  if (g_api.error) return sr_fail(-38, "%s: %s", who, g_api.error);
  return 0;
}

struct SrComm {
  ncclComm_t comm;
  hipEvent_t fork, join;       // compute -> comm stream (the bucket is ready), comm -> compute stream (the exchanges are done)
  int rank, world;
};

#define SR_NCCL(call_, who)                                                                    \
  do { const ncclResult_t r_ = (call_);                                                        \
       if (r_ != ncclSuccess) return sr_fail(-5, "%s: RCCL: %s", who, g_api.GetErrorString(r_)); } while (0)
#define SR_HIP(call_, who)                                                                     \
  do { const hipError_t e_ = (call_);                                                          \
       if (e_ != hipSuccess) return sr_fail(-5, "%s: %s", who, hipGetErrorString(e_)); } while (0)

int enqueue(SrComm* c, void* buf, size_t n, ncclDataType_t dt, ncclRedOp_t op, void* compute_stream, void* comm_stream,
            const char* who) {
  hipStream_t cs = (hipStream_t)compute_stream, xs = (hipStream_t)comm_stream;
  if (xs != cs) {              // the exchange starts when everything enqueued on the compute stream so far has run
    SR_HIP(hipEventRecord(c->fork, cs), who);
    SR_HIP(hipStreamWaitEvent(xs, c->fork, 0), who);
  }
  SR_NCCL(g_api.AllReduce(buf, buf, n, dt, op, c->comm, xs), who);
  return 0;
}

}  // namespace

int srhip_allreduce_unique_id(void* id128) {
  SR_REQUIRE(id128 != nullptr, "allreduce_unique_id: NULL");
  if (int rc = need_rccl("allreduce_unique_id")) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  SR_NCCL(g_api.GetUniqueId(&id), "allreduce_unique_id");
  memcpy(id128, &id, sizeof(id));
  return 0;
}

int srhip_allreduce_init(const void* id128, int rank, int world_size, void** comm) {
  SR_REQUIRE(id128 && comm, "allreduce_init: NULL argument");
  SR_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "allreduce_init: rank %d of %d", rank, world_size);
  if (int rc = need_rccl("allreduce_init")) return rc;
  SrComm* c = new (std::nothrow) SrComm();
  SR_REQUIRE(c != nullptr, "allreduce_init: out of memory");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = g_api.CommInitRank(&c->comm, world_size, id, rank);
  if (r != ncclSuccess) { delete c; return sr_fail(-5, "allreduce_init: RCCL: %s", g_api.GetErrorString(r)); }
  if (hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->join, hipEventDisableTiming) != hipSuccess) {
    g_api.CommDestroy(c->comm);
    delete c;
    return sr_fail(-5, "allreduce_init: cannot create events");
  }
  c->rank = rank; c->world = world_size;
  *comm = c;
  return 0;
}

int srhip_allreduce_bucket_async(void* comm, float* buf, long n, void* compute_stream, void* comm_stream) {
  SR_REQUIRE(comm && buf && n > 0, "allreduce_bucket_async: NULL / empty bucket");
  return enqueue((SrComm*)comm, buf, (size_t)n, ncclFloat32, ncclSum, compute_stream, comm_stream, "allreduce_bucket_async");
}

int srhip_allreduce_flag_async(void* comm, int* flag, void* compute_stream, void* comm_stream) {
  SR_REQUIRE(comm && flag, "allreduce_flag_async: NULL argument");
  return enqueue((SrComm*)comm, flag, 1, ncclInt32, ncclMax, compute_stream, comm_stream, "allreduce_flag_async");
}

int srhip_allreduce_wait(void* comm, void* comm_stream, void* compute_stream) {
  SR_REQUIRE(comm != nullptr, "allreduce_wait: NULL communicator");
  SrComm* c = (SrComm*)comm;
  hipStream_t cs = (hipStream_t)compute_stream, xs = (hipStream_t)comm_stream;
  if (xs == cs) return 0;
  SR_HIP(hipEventRecord(c->join, xs), "allreduce_wait");
  SR_HIP(hipStreamWaitEvent(cs, c->join, 0), "allreduce_wait");
  return 0;
}

int srhip_allreduce_destroy(void* comm) {
  if (!comm) return 0;
  SrComm* c = (SrComm*)comm;
  hipEventDestroy(c->fork);
  hipEventDestroy(c->join);
  const ncclResult_t r = g_api.CommDestroy(c->comm);
  delete c;
  if (r != ncclSuccess) return sr_fail(-5, "allreduce_destroy: RCCL: %s", g_api.GetErrorString(r));
  return 0;
}
