// Reassembly of the OH export across the GPUs of a node, through the C ABI (include/ohxgb.h part 4).
//
// Inside GEOS the OH field stays distributed: every MPI rank predicts its own (im, jm, km) block
// (OH_GridComp/OH_GridCompMod.F90:1199-1202, 1565) and nothing is exchanged.  The one exchange step of
// BASELINE.json's configs #4/#5 - every GPU ends with the whole field - is an all-gather of the float32 shards
// over xGMI.  A Fortran/MPI host has no torch.distributed: it gets the collective from here.  The host
// broadcasts the 128-byte id of OHXCommGetUniqueId with its own MPI (that is all MPI is needed for), every
// rank calls OHXCommInitRank, and OHXAllGatherOH enqueues the collective on the caller's stream.
//
// RCCL is loaded at the first call (dlopen "librccl.so"), not linked, and its header is not needed to build: the
// single-GPU product has no use for it and must build and load where it is absent.  The handful of types and
// prototypes used are declared below as RCCL's ABI has them (nccl.h: ncclUniqueId is 128 bytes, ncclFloat is 7).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdlib>
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

// RCCL's C ABI, as far as it is used here
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
typedef int ncclDataType_t;
constexpr ncclDataType_t ncclFloat = 7;

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
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
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  if (!r.so || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart ||
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
  // how equal shards travel: fixed when the communicator is made (OHX_ALLGATHER=pairs in THAT process's environment:
  // every rank must be started with the same setting - a rank that enqueues ncclAllGather while its peers enqueue
  // sends and receives hangs without a word; read once so that at least a setenv between two calls cannot split them)
  bool pairs = false;
  // held while a call enqueues on this communicator; OHXCommFree takes it before it destroys the communicator
  std::mutex use;
};

std::mutex g_mu;
std::unordered_set<const void*> g_live;

// A handle is checked against the table of live communicators, never by reading the object.  g_mu covers the table
// only: a call looks its communicator up under it, takes the communicator's own mutex, and lets g_mu go BEFORE it
// talks to RCCL - ncclGroupEnd may wait for a peer, and a process that drives several ranks from several threads must
// not hold every other communicator's calls up meanwhile.
CommObj* as_comm_locked(OHXCommHandle h) {
  if (h == nullptr || !g_live.count(h)) throw OhxError("communicator handle is invalid or has been freed");
  return static_cast<CommObj*>(h);
}

// closes an open RCCL group on every way out, the throwing ones included
struct GroupGuard {
  Rccl& r;
  bool open = false;
  explicit GroupGuard(Rccl& rr) : r(rr) {}
  void start() {
    nccl_check(r.GroupStart(), "ncclGroupStart");
    open = true;
  }
  void end() {
    open = false;
    nccl_check(r.GroupEnd(), "ncclGroupEnd");
  }
  ~GroupGuard() {
    if (open) (void)r.GroupEnd();
  }
};

// OHX_ALLGATHER=pairs asks for the direct exchange also where the shards are equal (default: ncclAllGather there)
bool pairs_wanted() {
  const char* e = getenv("OHX_ALLGATHER");
  return e != nullptr && strcmp(e, "pairs") == 0;
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
  c->pairs = pairs_wanted();
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
  CommObj* c = nullptr;
  {
    std::lock_guard<std::mutex> g(g_mu);            // look-up and erase in one step: of two frees one fails
    c = as_comm_locked(handle);
    g_live.erase(handle);
  }
  ncclResult_t rc;
  {
    std::lock_guard<std::mutex> u(c->use);          // a call that found it before the erase finishes its enqueue first
    rc = rccl().CommDestroy(c->comm);
  }
  delete c;
  nccl_check(rc, "ncclCommDestroy");
  COMM_API_END();
}

int OHXCommInfo(int* rccl_version) {
  COMM_API_BEGIN();
  int v = 0;
  nccl_check(rccl().GetVersion ? rccl().GetVersion(&v) : ncclSuccess, "ncclGetVersion");
  if (rccl_version) *rccl_version = v;
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
  std::unique_lock<std::mutex> table(g_mu);
  CommObj* c = as_comm_locked(handle);
  std::lock_guard<std::mutex> use(c->use);
  table.unlock();
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != c->device)
    throw OhxError("OHXAllGatherOH: the communicator was made on HIP device " + std::to_string(c->device) +
                   " but the current device is " + std::to_string(dev));
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
  if (nrows_total % (bst_ulong)c->nranks == 0 && !c->pairs) {
    // equal shards: one all-gather, straight into place (in place when d_shard already is d_full + row0)
    nccl_check(r.AllGather(d_shard, d_full, (size_t)nrows_local, ncclFloat, c->comm, s), "ncclAllGather");
  } else {
    // The direct exchange: every rank sends its shard to every peer and receives every peer's shard
    // at its rows, all of it one group - on the node's full xGMI mesh each pair has a link of its own, where a ring
    // is bound by one link per step.  Shards need not be equal.  The rank's own rows are a device copy.
    if (d_shard != d_full + row0 && nrows_local != 0) {
      hipError_t e = hipMemcpyAsync(d_full + row0, d_shard, (size_t)nrows_local * sizeof(float), hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) throw OhxError(std::string("OHXAllGatherOH: hipMemcpyAsync failed: ") + hipGetErrorString(e));
    }
    GroupGuard group(r);
    group.start();
    for (int q = 0; q < c->nranks; ++q) {
      if (q == c->rank) continue;
      bst_ulong r0 = 0, n = 0;
      (void)OHXShardRows(nrows_total, c->nranks, q, &r0, &n);
      if (nrows_local != 0) nccl_check(r.Send(d_shard, (size_t)nrows_local, ncclFloat, q, c->comm, s), "ncclSend");
      if (n != 0) nccl_check(r.Recv(d_full + r0, (size_t)n, ncclFloat, q, c->comm, s), "ncclRecv");
    }
    group.end();
  }
  COMM_API_END();
}

}  // extern "C"
#pragma GCC visibility pop
