// Gradient exchange behind the C ABI (mv_comm_init / allreduce_async / wait / destroy): RCCL over xGMI for a host that is not
// torch.  Replaces what nn.DataParallel does for the reference (models/train_origin.py:53-55: per-step parameter broadcast, logit gather,
// gradient reduce to GPU 0) by the one exchange the math needs -- a sum all-reduce of the flat gradient's buckets (medvill_amd/dist.py drives
// the same exchange through torch.distributed's nccl backend, which IS RCCL on ROCm; a torch process keeps using that: one communicator).
// RCCL is bound at RUN time (dlopen of librccl.so.1, then librccl.so): the library has no link-time dependency on it, and a process that
// never calls mv_comm_* never loads it.  Host code only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include "../../include/medvill.h"
// RCCL's types: from its header where the ROCm install has one, else the handful this file needs, declared here with RCCL's published
// values (the entry points are bound by name at run time either way: a ROCm without the RCCL development files still builds this library)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9 } ncclDataType_t;
#endif

namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  bool ok = false;
};

Rccl load_rccl() {
  Rccl r;
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
// the function table is filled exactly once, by whichever thread gets here first (C++11 function-local static: concurrent first
// callers block until it is complete; it is immutable afterwards)
const Rccl& rccl() {
  static const Rccl r = load_rccl();
  return r;
}

// One event per DISTINCT stream a collective was issued on since the last wait: a host that issues buckets from two streams gets its
// compute stream ordered behind BOTH (round 4 kept one event, re-recorded on whichever stream issued last).
struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  std::mutex mu;
  struct Pending { hipStream_t stream; hipEvent_t done; bool live; };
  std::vector<Pending> pending;      // entries are reused (events are created once per distinct stream)
};

// RCCL failures: MV_E_COMM_BASE - ncclResult_t, i.e. -1001, -1002, ... -- negative like every MV_E_* code, and outside hipError_t's range
// (hipErrorRuntimeMemory = 1052 etc. are POSITIVE values above 1000, which the round-4 encoding 1000 + e collided with)
inline int rc_of(ncclResult_t e) { return e == ncclSuccess ? MV_OK : MV_E_COMM_BASE - (int)e; }

}  // namespace

extern "C" int mv_comm_unique_id(void* id128) {
  if (!id128) return MV_E_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return MV_E_NO_RCCL;
  return rc_of(r.GetUniqueId((ncclUniqueId*)id128));
}

extern "C" int mv_comm_init(void** comm_out, int rank, int world, const void* id128) {
  if (!comm_out || !id128 || world <= 0 || rank < 0 || rank >= world) return MV_E_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return MV_E_NO_RCCL;
  Comm* c = new Comm();
  c->rank = rank; c->world = world;
  ncclUniqueId id;
  __builtin_memcpy(&id, id128, sizeof(id));
  const ncclResult_t e = r.CommInitRank(&c->comm, world, id, rank);           // on the calling thread's current device
  if (e != ncclSuccess) { delete c; return rc_of(e); }
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
  std::lock_guard<std::mutex> lock(c->mu);        // RCCL wants the collectives of one communicator issued in one order on every rank
  const ncclResult_t e = rccl().AllReduce(buf, buf, count, dt, ncclSum, c->comm, stream);      // in place; RCCL's kernels run on `stream`
  if (e != ncclSuccess) return rc_of(e);
  Comm::Pending* slot = nullptr;
  for (auto& p : c->pending)
    if (p.stream == stream) { slot = &p; break; }
  if (!slot) {
    hipEvent_t ev = nullptr;
    const hipError_t hc = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (hc != hipSuccess) return (int)hc;
    c->pending.push_back({stream, ev, false});
    slot = &c->pending.back();
  }
  const hipError_t he = hipEventRecord(slot->done, stream);      // behind the collective issued last ON THIS STREAM
  if (he != hipSuccess) return (int)he;
  slot->live = true;
  return MV_OK;
}

extern "C" int mv_comm_wait(void* comm, void* stream_) {
  if (!comm) return MV_E_ARG;
  Comm* c = (Comm*)comm;
  std::lock_guard<std::mutex> lock(c->mu);
  for (auto& p : c->pending) {        // every stream that issued a collective since the last wait
    if (!p.live) continue;
    const hipError_t he = hipStreamWaitEvent((hipStream_t)stream_, p.done, 0);      // device-side: the host does not block
    if (he != hipSuccess) return (int)he;
    p.live = false;
  }
  return MV_OK;
}

extern "C" int mv_comm_destroy(void* comm) {
  if (!comm) return MV_E_ARG;
  Comm* c = (Comm*)comm;
  int rc = MV_OK;
  for (auto& p : c->pending) (void)hipEventDestroy(p.done);
  if (c->comm) rc = rc_of(rccl().CommDestroy(c->comm));
  delete c;
  return rc;
}
