// The one exchange step of the batch-data-parallel path behind the C-ABI (SURVEY.md 8b export list `rccl_allgather_outputs`, 8e): every rank's
// packed output rows go to every other rank over RCCL / xGMI.  One process per GPU, one communicator per process.  librccl is opened on first use
// (dlopen): the kernel library itself carries no link-time dependency on it, and a single-GPU user never loads it.
// The reference has no distributed inference path; unidepth/utils/distributed.py:153-176 (sync_tensor_across_gpus: size all-gather -> pad -> gather
// -> trim) is its pattern for variable-length gathers, which the host side (unidepth_amd/dist.py) keeps.
#include <dlfcn.h>
#include <stddef.h>
#include <string.h>
#include "ud_common.h"

namespace {

// the slice of rccl.h this file needs (ABI of ROCm 7.x librccl.so.1): opaque communicator, 128-byte unique id, int-valued enums
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                     // 0 = ncclSuccess
constexpr int kNcclInt8 = 0;

struct Api {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} api;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

bool load() {
  if (api.so) return true;
  void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!so) so = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!so) {
    ud_set_error("ud_rccl: librccl.so not found");
    return false;
  }
#define UD_SYM(field, name)                                        \
  *(void**)(&api.field) = dlsym(so, name);                         \
  if (!api.field) {                                                \
    ud_set_error("ud_rccl: symbol " name " missing in librccl");   \
    dlclose(so);                                                   \
    return false;                                                  \
  }
  UD_SYM(GetUniqueId, "ncclGetUniqueId")
  UD_SYM(CommInitRank, "ncclCommInitRank")
  UD_SYM(CommDestroy, "ncclCommDestroy")
  UD_SYM(AllGather, "ncclAllGather")
  UD_SYM(Send, "ncclSend")
  UD_SYM(Recv, "ncclRecv")
  UD_SYM(GroupStart, "ncclGroupStart")
  UD_SYM(GroupEnd, "ncclGroupEnd")
  UD_SYM(GetErrorString, "ncclGetErrorString")
#undef UD_SYM
  api.so = so;
  return true;
}

int fail(const char* what, ncclResult_t rc) {
  char msg[200];
  snprintf(msg, sizeof(msg), "%s: %s", what, api.GetErrorString ? api.GetErrorString(rc) : "rccl error");
  ud_set_error(msg);
  return UD_ERR_LAUNCH;
}

}  // namespace

extern "C" int ud_rccl_unique_id(void* id128) {
  if (!id128) {
    ud_set_error("ud_rccl_unique_id: NULL output");
    return UD_ERR_BAD_ARG;
  }
  if (!load()) return UD_ERR_LAUNCH;
  ncclUniqueId id;
  const ncclResult_t rc = api.GetUniqueId(&id);
  if (rc) return fail("ncclGetUniqueId", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return UD_OK;
}

extern "C" int ud_rccl_init(const void* id128, int world, int rank) {
  if (!id128 || world <= 0 || rank < 0 || rank >= world) {
    ud_set_error("ud_rccl_init: bad argument (id, 0 <= rank < world)");
    return UD_ERR_BAD_ARG;
  }
  if (g_comm) {
    ud_set_error("ud_rccl_init: already initialised in this process (ud_rccl_finalize first)");
    return UD_ERR_BAD_ARG;
  }
  if (!load()) return UD_ERR_LAUNCH;
  ncclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  const ncclResult_t rc = api.CommInitRank(&g_comm, world, id, rank);       // binds to the calling thread's current HIP device
  if (rc) {
    g_comm = nullptr;
    return fail("ncclCommInitRank", rc);
  }
  g_world = world;
  g_rank = rank;
  return UD_OK;
}

extern "C" int ud_rccl_allgather_outputs(const void* send, void* recv, size_t bytes_per_rank, int direct, void* stream) {
  if (!g_comm) {
    ud_set_error("ud_rccl_allgather_outputs: no communicator (ud_rccl_init)");
    return UD_ERR_BAD_ARG;
  }
  if (!send || !recv || bytes_per_rank == 0) {
    ud_set_error("ud_rccl_allgather_outputs: bad argument");
    return UD_ERR_BAD_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  ncclResult_t rc;
  if (!direct || g_world == 1) {
    rc = api.AllGather(send, recv, bytes_per_rank, kNcclInt8, g_comm, s);
    if (rc) return fail("ncclAllGather", rc);
    return UD_OK;
  }
  // all-pairs form for the xGMI full mesh (7 point-to-point links per GPU, no switch): every rank sends its block to each peer over that
  // peer's own link, all links busy at once -- a ring forwards every block over world - 1 hops, each bound by ONE link
  char* out = (char*)recv;
  if (hipMemcpyAsync(out + (size_t)g_rank * bytes_per_rank, send, bytes_per_rank, hipMemcpyDeviceToDevice, s) != hipSuccess) {
    ud_set_error("ud_rccl_allgather_outputs: local block copy failed");
    return UD_ERR_LAUNCH;
  }
  rc = api.GroupStart();
  if (rc) return fail("ncclGroupStart", rc);
  for (int k = 1; k < g_world; ++k) {                     // peer order rotated per rank: every step pairs distinct links
    const int dst = (g_rank + k) % g_world, src = (g_rank - k + g_world) % g_world;
    rc = api.Send(send, bytes_per_rank, kNcclInt8, dst, g_comm, s);
    if (!rc) rc = api.Recv(out + (size_t)src * bytes_per_rank, bytes_per_rank, kNcclInt8, src, g_comm, s);
    if (rc) {
      api.GroupEnd();
      return fail("ncclSend / ncclRecv", rc);
    }
  }
  rc = api.GroupEnd();
  if (rc) return fail("ncclGroupEnd", rc);
  return UD_OK;
}

extern "C" int ud_rccl_finalize(void) {
  if (g_comm) {
    const ncclResult_t rc = api.CommDestroy(g_comm);
    g_comm = nullptr;
    g_world = 0;
    g_rank = -1;
    if (rc) return fail("ncclCommDestroy", rc);
  }
  return UD_OK;
}
