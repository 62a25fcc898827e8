// RCCL behind the C ABI (segmif_comm_{init, allreduce, destroy}): the gradient exchange of the
// data-parallel training steps (one sum / average all-reduce of fp32 gradients per step over xGMI) for a host
// that is not Python.  The Python host of this repo uses torch.distributed (backend "nccl" = RCCL) - the same library, and
// the one whose communicator torch already owns -, so segmif_amd/parallel.py does not route through these entry points.
//
// librccl is NOT a link-time dependency of libsegmif_hip.so: it is looked up on first use, preferring the copy the process
// has already loaded (torch ships its own librccl.so; two RCCL instances in one process must not happen), so the kernels'
// library loads on boxes without RCCL and `import torch` is never influenced by it.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

// The few RCCL types and constants these entry points use.  With the RCCL development headers present they come from
// <rccl/rccl.h>; without them (a box that only runs the kernels) the same ABI-stable declarations are made here, so the kernel
// library still BUILDS there as the paragraph above promises - the functions are resolved with dlsym either way.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3, ncclAvg = 4 } ncclRedOp_t;
#endif

#include "segmif_hip.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)  // a copy the process already holds (torch's) first
      if (!x.lib) x.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!x.lib) x.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!x.lib) return x;
    x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.lib, "ncclGetUniqueId");
    x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.lib, "ncclCommInitRank");
    x.AllReduce = (decltype(x.AllReduce))dlsym(x.lib, "ncclAllReduce");
    x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.lib, "ncclCommDestroy");
    x.CommCount = (decltype(x.CommCount))dlsym(x.lib, "ncclCommCount");
    x.CommUserRank = (decltype(x.CommUserRank))dlsym(x.lib, "ncclCommUserRank");
    x.GetVersion = (decltype(x.GetVersion))dlsym(x.lib, "ncclGetVersion");
    x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy && x.CommCount && x.CommUserRank;
    return x;
  }();
  return r;
}

inline int rc(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }  // RCCL results are reported as 1000 + ncclResult_t

}  // namespace

extern "C" int segmif_comm_available(int* version) {
  Rccl& r = rccl();
  if (!r.ok) return SEGMIF_ENOSYS;
  if (version) {
    *version = 0;
    if (r.GetVersion) r.GetVersion(version);
  }
  return 0;
}

extern "C" int segmif_comm_unique_id(void* id, int64_t bytes) {
  if (!id || bytes < (int64_t)sizeof(ncclUniqueId)) return SEGMIF_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return SEGMIF_ENOSYS;
  ncclUniqueId u;
  const int e = rc(r.GetUniqueId(&u));
  if (e == 0) memcpy(id, &u, sizeof(u));
  return e;
}

extern "C" int segmif_comm_init(void** comm, int world, int rank, const void* id, int64_t bytes) {
  if (!comm || !id || world < 1 || rank < 0 || rank >= world || bytes < (int64_t)sizeof(ncclUniqueId)) return SEGMIF_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return SEGMIF_ENOSYS;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t c = nullptr;
  const int e = rc(r.CommInitRank(&c, world, u, rank));  // binds to the calling thread's current HIP device
  *comm = e == 0 ? (void*)c : nullptr;
  return e;
}

extern "C" int segmif_comm_world(void* comm, int* world, int* rank) {
  if (!comm || !world || !rank) return SEGMIF_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return SEGMIF_ENOSYS;
  int e = rc(r.CommCount((ncclComm_t)comm, world));
  if (e == 0) e = rc(r.CommUserRank((ncclComm_t)comm, rank));
  return e;
}

extern "C" int segmif_comm_allreduce_f32(void* comm, const float* send, float* recv, int64_t count, int average, void* stream) {
  if (!comm || !send || !recv || count <= 0) return SEGMIF_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return SEGMIF_ENOSYS;
  return rc(r.AllReduce(send, recv, (size_t)count, ncclFloat32, average ? ncclAvg : ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int segmif_comm_destroy(void* comm) {
  if (!comm) return SEGMIF_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return SEGMIF_ENOSYS;
  return rc(r.CommDestroy((ncclComm_t)comm));
}
