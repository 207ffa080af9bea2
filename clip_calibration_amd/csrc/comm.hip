// The one exchange step of the path: an RCCL all-gather over xGMI of the per-GPU L2-normalised image
// embeddings before the shared text-feature matmul.  The reference has no counterpart -- it is single-process and wraps
// the model in nn.DataParallel, which re-broadcasts the weights on every call (trainers/classification/coop.py:266-272,
// trainers/calibration/tempscaling.py:117-120); here one process drives one GPU, weights and text features stay resident,
// and the only bytes that cross xGMI per step are [B/G, E] fp16 embeddings per rank.
//
// librccl is opened at run time (dlopen) the first time a communicator is asked for: the rest of the library has no
// RCCL dependency, and a process that already loaded RCCL (torch.distributed "nccl") shares that copy by soname.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>

#include <mutex>
#include <new>

#include <rccl/rccl.h>

#include "common.h"

namespace clipmi {
namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "";
};

RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi& rccl() {
  std::call_once(g_rccl_once, [] {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      snprintf(g_rccl.why, sizeof(g_rccl.why), "cannot open librccl: %s", dlerror());
      return;
    }
    auto sym = [&](const char* name) {
      void* p = dlsym(h, name);
      if (!p && !g_rccl.why[0]) snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl lacks %s", name);
      return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(sym("ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(g_rccl.CommUserRank)>(sym("ncclCommUserRank"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.ok = g_rccl.why[0] == 0;
  });
  return g_rccl;
}

int rccl_fail(const char* what, ncclResult_t r) {
  const RcclApi& api = rccl();
  set_error("%s: %s", what, api.GetErrorString ? api.GetErrorString(r) : "RCCL error");
  return CLIPMI_ERR_HIP;
}

}  // namespace
}  // namespace clipmi

using namespace clipmi;

struct clipmi_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0;
};

extern "C" {

int clipmi_comm_unique_id(void* id_out) {
  CLIPMI_REQUIRE(id_out, CLIPMI_ERR_ARG, "comm_unique_id: null pointer");
  const RcclApi& api = rccl();
  CLIPMI_REQUIRE(api.ok, CLIPMI_ERR_HIP, "RCCL unavailable: %s", api.why);
  static_assert(sizeof(ncclUniqueId) == CLIPMI_COMM_ID_BYTES, "unique id size");
  const ncclResult_t r = api.GetUniqueId(static_cast<ncclUniqueId*>(id_out));
  return r == ncclSuccess ? CLIPMI_OK : rccl_fail("ncclGetUniqueId", r);
}

int clipmi_comm_create(const void* id, int world, int rank, clipmi_comm** out) {
  CLIPMI_REQUIRE(id && out, CLIPMI_ERR_ARG, "comm_create: null pointer");
  CLIPMI_REQUIRE(world >= 1 && rank >= 0 && rank < world, CLIPMI_ERR_ARG, "comm_create: rank %d outside world %d", rank, world);
  const RcclApi& api = rccl();
  CLIPMI_REQUIRE(api.ok, CLIPMI_ERR_HIP, "RCCL unavailable: %s", api.why);
  clipmi_comm* c = new (std::nothrow) clipmi_comm();
  CLIPMI_REQUIRE(c, CLIPMI_ERR_ARG, "comm_create: out of host memory");
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = api.CommInitRank(&c->comm, world, uid, rank);   // collective over the ranks; binds the CURRENT device
  if (r != ncclSuccess) {
    delete c;
    return rccl_fail("ncclCommInitRank", r);
  }
  // what the communicator itself says (bench.py prints it: proof that N ranks were really seen)
  if (api.CommCount(c->comm, &c->world) != ncclSuccess || api.CommUserRank(c->comm, &c->rank) != ncclSuccess) {
    c->world = world;
    c->rank = rank;
  }
  *out = c;
  return CLIPMI_OK;
}

int clipmi_comm_destroy(clipmi_comm* c) {
  if (!c) return CLIPMI_OK;
  ncclResult_t r = ncclSuccess;
  if (c->comm) r = rccl().CommDestroy(c->comm);
  delete c;
  return r == ncclSuccess ? CLIPMI_OK : rccl_fail("ncclCommDestroy", r);
}

int clipmi_comm_ranks(const clipmi_comm* c, int* world, int* rank) {
  CLIPMI_REQUIRE(c && world && rank, CLIPMI_ERR_ARG, "comm_ranks: null pointer");
  *world = c->world;
  *rank = c->rank;
  return CLIPMI_OK;
}

int clipmi_allgather(clipmi_comm* c, const void* in, void* out, size_t bytes_per_rank, clipmi_stream_t stream) {
  CLIPMI_REQUIRE(c && c->comm, CLIPMI_ERR_ARG, "allgather: null communicator");
  if (bytes_per_rank == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(in && out, CLIPMI_ERR_ARG, "allgather: null buffer");
  const ncclResult_t r = rccl().AllGather(in, out, bytes_per_rank, ncclUint8, c->comm, (hipStream_t)stream);
  return r == ncclSuccess ? CLIPMI_OK : rccl_fail("ncclAllGather", r);
}

}  // extern "C"
