// RCCL entry points of the C ABI: thin wrappers over ncclComm_t that run stream-ordered on the
// HIP stream the caller names, so that the trainer can put them on a side stream next to its kernels.
//   srl_allreduce_stats_f64x3  <- the three all_reduce calls of masked_normalization (modules/utils.py:58-61) and of
//                                 RunningMeanStd.update (modules/utils.py:121-124): ONE message
//   srl_allreduce_grads        <- DistributedDataParallel's bucketed gradient all-reduce (api/policy.py:219-238)
//   srl_broadcast_params       <- the DDP constructor's parameter broadcast (same lines) and the parameter push / pull
//                                 between trainer and policy workers (system/parameter_db.py), as one flat buffer
// librccl is resolved at run time with dlopen: a process that has PyTorch loaded gets the very library torch
// itself uses (same SONAME), a plain C host gets the system one -- libsrlhip.so carries no link-time dependency.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "srl_common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.ok ? &r : nullptr;
  tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) return nullptr;
#define SRL_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
  SRL_SYM(GetUniqueId, "ncclGetUniqueId");
  SRL_SYM(CommInitRank, "ncclCommInitRank");
  SRL_SYM(CommDestroy, "ncclCommDestroy");
  SRL_SYM(CommCount, "ncclCommCount");
  SRL_SYM(AllReduce, "ncclAllReduce");
  SRL_SYM(Broadcast, "ncclBroadcast");
  SRL_SYM(GetErrorString, "ncclGetErrorString");
#undef SRL_SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.AllReduce && r.Broadcast && r.GetErrorString;
  return r.ok ? &r : nullptr;
}

#define SRL_RCCL_TRY(R, expr)                                                          \
  do {                                                                                 \
    ncclResult_t _e = (expr);                                                          \
    if (_e != ncclSuccess) {                                                           \
      srl_set_error("%s: %s failed: %s", __func__, #expr, (R)->GetErrorString(_e));    \
      return -EIO;                                                                     \
    }                                                                                  \
  } while (0)

#define SRL_NEED_RCCL(R)                                                               \
  Rccl* R = rccl();                                                                    \
  if (!R) {                                                                            \
    srl_set_error("%s: librccl could not be loaded: %s", __func__, dlerror());         \
    return -ENOSYS;                                                                    \
  }

}  // namespace

extern "C" int srl_comm_available(void) { return rccl() != nullptr ? 1 : 0; }

extern "C" int srl_comm_unique_id(void* id_out) {
  SRL_CHECK_ARG(id_out, "null id buffer");
  SRL_NEED_RCCL(R);
  static_assert(sizeof(ncclUniqueId) == SRL_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  SRL_RCCL_TRY(R, R->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return 0;
}

extern "C" int srl_comm_init(void** comm_out, const void* id, int rank, int world) {
  SRL_CHECK_ARG(comm_out && id && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
  SRL_NEED_RCCL(R);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  SRL_RCCL_TRY(R, R->CommInitRank(&c, world, uid, rank));  // binds to the calling thread's current HIP device
  *comm_out = c;
  return 0;
}

extern "C" int srl_comm_world(void* comm, int* world_out) {
  SRL_CHECK_ARG(comm && world_out, "null argument");
  SRL_NEED_RCCL(R);
  SRL_RCCL_TRY(R, R->CommCount((ncclComm_t)comm, world_out));
  return 0;
}

extern "C" int srl_comm_destroy(void* comm) {
  if (!comm) return 0;
  SRL_NEED_RCCL(R);
  SRL_RCCL_TRY(R, R->CommDestroy((ncclComm_t)comm));
  return 0;
}

extern "C" int srl_allreduce_stats_f64x3(void* stream, void* comm, double* stats, int count) {
  SRL_CHECK_ARG(comm && stats && count >= 0, "null argument");
  if (count == 0) return 0;
  SRL_NEED_RCCL(R);
  SRL_RCCL_TRY(R, R->AllReduce(stats, stats, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

extern "C" int srl_allreduce_grads(void* stream, void* comm, float* grad, int64_t n) {
  SRL_CHECK_ARG(comm && grad && n >= 0, "null argument");
  if (n == 0) return 0;
  SRL_NEED_RCCL(R);
  SRL_RCCL_TRY(R, R->AllReduce(grad, grad, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}

extern "C" int srl_broadcast_params(void* stream, void* comm, void* buf, int64_t nbytes, int root) {
  SRL_CHECK_ARG(comm && buf && nbytes >= 0 && root >= 0, "bad argument");
  if (nbytes == 0) return 0;
  SRL_NEED_RCCL(R);
  SRL_RCCL_TRY(R, R->Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream));
  return 0;
}
