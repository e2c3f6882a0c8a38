// Gradient exchange behind the C ABI (mv_comm_init / allreduce_async / wait / destroy): RCCL over xGMI for a host that is not
// torch.  Replaces what nn.DataParallel does for the reference (models/train_origin.py:53-55: per-step parameter broadcast, logit gather,
// gradient reduce to GPU 0) by the one exchange the math needs -- a sum all-reduce of the flat gradient's buckets (medvill_amd/dist.py drives
// the same exchange through torch.distributed's nccl backend, which IS RCCL on ROCm; a torch process keeps using that: one communicator).
// RCCL is bound at RUN time (dlopen of librccl.so.1, then librccl.so): the library has no link-time dependency on it, and a process that
// never calls mv_comm_* never loads it.  Host code only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "../../include/medvill.h"

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.so) break;
  }
  if (!r.so) return r;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
  r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
  return r;
}

struct Comm {
  ncclComm_t comm = nullptr;
  hipEvent_t done = nullptr;      // recorded behind the collective issued last
  int rank = 0, world = 1;
  bool pending = false;
};

inline int rc_of(ncclResult_t e) { return e == ncclSuccess ? MV_OK : MV_E_COMM_BASE + (int)e; }

}  // namespace

extern "C" int mv_comm_unique_id(void* id128) {
  if (!id128) return MV_E_ARG;
  Rccl& r = rccl();
  if (!r.ok) return MV_E_NO_RCCL;
  return rc_of(r.GetUniqueId((ncclUniqueId*)id128));
}

extern "C" int mv_comm_init(void** comm_out, int rank, int world, const void* id128) {
  if (!comm_out || !id128 || world <= 0 || rank < 0 || rank >= world) return MV_E_ARG;
  Rccl& r = rccl();
  if (!r.ok) return MV_E_NO_RCCL;
  Comm* c = new Comm();
  c->rank = rank; c->world = world;
  ncclUniqueId id;
  __builtin_memcpy(&id, id128, sizeof(id));
  const ncclResult_t e = r.CommInitRank(&c->comm, world, id, rank);           // on the calling thread's current device
  if (e != ncclSuccess) { delete c; return rc_of(e); }
  const hipError_t he = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
  if (he != hipSuccess) { r.CommDestroy(c->comm); delete c; return (int)he; }
  *comm_out = c;
  return MV_OK;
}

extern "C" int mv_comm_allreduce_async(void* comm, void* buf, size_t count, int dtype, void* stream_) {
  if (!comm || !buf || count == 0) return MV_E_ARG;
  Comm* c = (Comm*)comm;
  ncclDataType_t dt;
  if (dtype == MV_F32) dt = ncclFloat32;
  else if (dtype == MV_F16) dt = ncclFloat16;
  else if (dtype == MV_BF16) dt = ncclBfloat16;
  else return MV_E_DTYPE;
  hipStream_t stream = (hipStream_t)stream_;
  const ncclResult_t e = rccl().AllReduce(buf, buf, count, dt, ncclSum, c->comm, stream);      // in place; RCCL's kernels run on `stream`
  if (e != ncclSuccess) return rc_of(e);
  const hipError_t he = hipEventRecord(c->done, stream);
  if (he != hipSuccess) return (int)he;
  c->pending = true;
  return MV_OK;
}

extern "C" int mv_comm_wait(void* comm, void* stream_) {
  if (!comm) return MV_E_ARG;
  Comm* c = (Comm*)comm;
  if (!c->pending) return MV_OK;
  const hipError_t he = hipStreamWaitEvent((hipStream_t)stream_, c->done, 0);      // device-side: the host does not block
  return he == hipSuccess ? MV_OK : (int)he;
}

extern "C" int mv_comm_destroy(void* comm) {
  if (!comm) return MV_E_ARG;
  Comm* c = (Comm*)comm;
  int rc = MV_OK;
  if (c->done) (void)hipEventDestroy(c->done);
  if (c->comm) rc = rc_of(rccl().CommDestroy(c->comm));
  delete c;
  return rc;
}
