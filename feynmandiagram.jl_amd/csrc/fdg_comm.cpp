// Multi-GPU side of the path: samples are sharded over one process per GPU, every rank
// accumulates its own R doubles, and ONE collective adds them up -- RCCL over xGMI, latency-bound
// (R <= a few dozen doubles).  RCCL is bound at run time (dlopen) so that libfdg.so shares whatever
// librccl.so.1 the process already has (PyTorch wheels bundle their own) and loads without it on a
// single GPU.  The reference has no counterpart (its examples are single-process).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "fdg_internal.h"

namespace {
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
std::string g_why;

void load_rccl() {
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    g_rccl.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.h) break;
  }
  if (!g_rccl.h) { g_why = std::string("RCCL not found: ") + dlerror(); return; }
  auto sym = [&](const char *n) { void *p = dlsym(g_rccl.h, n); if (!p) g_why = std::string("RCCL lacks ") + n; return p; };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
  g_rccl.Reduce = (decltype(g_rccl.Reduce))sym("ncclReduce");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
}
bool have_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_why.empty()) { fdg::set_error(g_why); return false; }
  return true;
}
int nccl_fail(const char *what, ncclResult_t r) {
  fdg::set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
  return FDG_E_NO_DEVICE;
}
}  // namespace

struct fdg_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int fdg_comm_unique_id(void *id, size_t bytes) {
  if (!id || bytes < FDG_COMM_ID_BYTES) { fdg::set_error("fdg_comm_unique_id: buffer of FDG_COMM_ID_BYTES needed"); return FDG_E_INVALID; }
  if (!have_rccl()) return FDG_E_NO_DEVICE;
  static_assert(sizeof(ncclUniqueId) == FDG_COMM_ID_BYTES, "id size");
  ncclUniqueId u;
  const ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  std::memcpy(id, &u, sizeof u);
  return FDG_OK;
}

int fdg_comm_create(const void *id, int rank, int world, fdg_comm **out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) { fdg::set_error("fdg_comm_create: bad argument"); return FDG_E_INVALID; }
  if (!have_rccl()) return FDG_E_NO_DEVICE;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  fdg_comm *c = new fdg_comm;
  c->rank = rank; c->world = world;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);   // binds to the calling thread's current device
  if (r != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", r); }
  *out = c;
  return FDG_OK;
}

int fdg_comm_destroy(fdg_comm *c) {
  if (!c) return FDG_OK;
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  delete c;
  return FDG_OK;
}

int fdg_reduce_device(fdg_comm *c, double *d_acc, uint32_t n, int root, void *stream) {
  if (!c || (!d_acc && n)) { fdg::set_error("fdg_reduce_device: null argument"); return FDG_E_INVALID; }
  if (root >= c->world) { fdg::set_error("fdg_reduce_device: root out of range"); return FDG_E_INVALID; }
  if (n == 0) return FDG_OK;
  const ncclResult_t r = root < 0
      ? g_rccl.AllReduce(d_acc, d_acc, n, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream)
      : g_rccl.Reduce(d_acc, d_acc, n, ncclFloat64, ncclSum, root, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return nccl_fail(root < 0 ? "ncclAllReduce" : "ncclReduce", r);
  return FDG_OK;
}

}  // extern "C"
