// Reassembly of the OH export across the GPUs of a node, through the C ABI (include/ohxgb.h part 4).
//
// Inside GEOS the OH field stays distributed: every MPI rank predicts its own (im, jm, km) block
// (OH_GridComp/OH_GridCompMod.F90:1199-1202, 1565) and nothing is exchanged.  The one exchange step of
// BASELINE.json's configs #4/#5 - every GPU ends with the whole field - is an all-gather of the float32 shards
// over xGMI.  A Fortran/MPI host has no torch.distributed: it gets the collective from here.  The host
// broadcasts the 128-byte id of OHXCommGetUniqueId with its own MPI (that is all MPI is needed for), every
// rank calls OHXCommInitRank, and OHXAllGatherOH enqueues the collective on the caller's stream.
//
// RCCL is loaded at the first call (dlopen "librccl.so"), not linked: the single-GPU product has no use for it
// and must load where it is absent.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <unordered_set>

#include "../../include/ohxgb.h"
#include "forest.hpp"

using ohx::OhxError;

namespace ohx {
void set_last_error(const std::string& m);   // capi.cpp: the per-thread text behind XGBGetLastError
}

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  static std::string why;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
      why = dlerror();
    }
    if (!r.so) return;
    auto sym = [&](const char* n) {
      void* p = dlsym(r.so, n);
      if (!p) why = std::string("librccl.so lacks ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  if (!r.so || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Broadcast || !r.GroupStart ||
      !r.GroupEnd || !r.GetErrorString)
    throw OhxError("RCCL is not usable here (the multi-GPU all-gather needs librccl.so): " + why);
  return r;
}

void nccl_check(ncclResult_t rc, const char* what) {
  if (rc != ncclSuccess) throw OhxError(std::string(what) + " failed: " + rccl().GetErrorString(rc));
}

struct CommObj {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = -1;
};

std::mutex g_mu;
std::unordered_set<const void*> g_live;

CommObj* as_comm(OHXCommHandle h) {
  std::lock_guard<std::mutex> g(g_mu);
  if (h == nullptr || !g_live.count(h)) throw OhxError("communicator handle is invalid or has been freed");
  return static_cast<CommObj*>(h);
}

#define COMM_API_BEGIN() try {
#define COMM_API_END()                        \
  }                                           \
  catch (const std::exception& e) {           \
    ohx::set_last_error(e.what());            \
    return -1;                                \
  }                                           \
  catch (...) {                               \
    ohx::set_last_error("unknown error");     \
    return -1;                                \
  }                                           \
  return 0;

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int OHXCommGetUniqueId(void* id) {
  COMM_API_BEGIN();
  static_assert(sizeof(ncclUniqueId) == OHX_UNIQUE_ID_BYTES, "include/ohxgb.h: OHX_UNIQUE_ID_BYTES");
  if (id == nullptr) throw OhxError("OHXCommGetUniqueId: id is NULL");
  nccl_check(rccl().GetUniqueId(static_cast<ncclUniqueId*>(id)), "ncclGetUniqueId");
  COMM_API_END();
}

int OHXCommInitRank(const void* id, int nranks, int rank, OHXCommHandle* out) {
  COMM_API_BEGIN();
  if (id == nullptr || out == nullptr) throw OhxError("OHXCommInitRank: NULL argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) throw OhxError("OHXCommInitRank: need 0 <= rank < nranks");
  auto c = new CommObj();
  c->nranks = nranks;
  c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    throw OhxError("OHXCommInitRank: no current HIP device");
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t rc = rccl().CommInitRank(&c->comm, nranks, uid, rank);
  if (rc != ncclSuccess) {
    delete c;
    nccl_check(rc, "ncclCommInitRank");
  }
  {
    std::lock_guard<std::mutex> g(g_mu);
    g_live.insert(c);
  }
  *out = c;
  COMM_API_END();
}

int OHXCommFree(OHXCommHandle handle) {
  COMM_API_BEGIN();
  CommObj* c = as_comm(handle);
  {
    std::lock_guard<std::mutex> g(g_mu);
    g_live.erase(handle);
  }
  ncclResult_t rc = rccl().CommDestroy(c->comm);
  delete c;
  nccl_check(rc, "ncclCommDestroy");
  COMM_API_END();
}

int OHXShardRows(bst_ulong nrows_total, int nranks, int rank, bst_ulong* row0, bst_ulong* nrows) {
  COMM_API_BEGIN();
  if (nranks < 1 || rank < 0 || rank >= nranks) throw OhxError("OHXShardRows: need 0 <= rank < nranks");
  const bst_ulong base = nrows_total / (bst_ulong)nranks, rem = nrows_total % (bst_ulong)nranks;
  if (row0) *row0 = (bst_ulong)rank * base + ((bst_ulong)rank < rem ? (bst_ulong)rank : rem);
  if (nrows) *nrows = base + ((bst_ulong)rank < rem ? 1 : 0);
  COMM_API_END();
}

int OHXAllGatherOH(OHXCommHandle handle, const float* d_shard, bst_ulong nrows_local, bst_ulong nrows_total,
                   float* d_full, void* stream) {
  COMM_API_BEGIN();
  CommObj* c = as_comm(handle);
  bst_ulong row0 = 0, mine = 0;
  (void)OHXShardRows(nrows_total, c->nranks, c->rank, &row0, &mine);
  if (mine != nrows_local)
    throw OhxError("OHXAllGatherOH: rank " + std::to_string(c->rank) + " of " + std::to_string(c->nranks) + " holds " +
                   std::to_string(nrows_local) + " rows but its shard of " + std::to_string(nrows_total) + " rows is " +
                   std::to_string(mine) + " (OHXShardRows)");
  if (nrows_total == 0) return 0;
  if (d_full == nullptr || (d_shard == nullptr && nrows_local != 0)) throw OhxError("OHXAllGatherOH: NULL buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Rccl& r = rccl();
  if (nrows_total % (bst_ulong)c->nranks == 0) {
    // equal shards: one all-gather, straight into place (in place when d_shard already is d_full + row0)
    nccl_check(r.AllGather(d_shard, d_full, (size_t)nrows_local, ncclFloat, c->comm, s), "ncclAllGather");
  } else {
    // ragged shards: every rank broadcasts its own piece to its rows, as one group
    nccl_check(r.GroupStart(), "ncclGroupStart");
    for (int q = 0; q < c->nranks; ++q) {
      bst_ulong r0 = 0, n = 0;
      (void)OHXShardRows(nrows_total, c->nranks, q, &r0, &n);
      if (n == 0) continue;
      nccl_check(r.Broadcast(q == c->rank ? (const void*)d_shard : (const void*)(d_full + r0), d_full + r0, (size_t)n,
                             ncclFloat, q, c->comm, s), "ncclBroadcast");
    }
    nccl_check(r.GroupEnd(), "ncclGroupEnd");
  }
  COMM_API_END();
}

}  // extern "C"
#pragma GCC visibility pop
