// mock_rccl — a TEST DOUBLE for librccl.so.  Test infrastructure: never shipped, never linked by the product.
//
// quickchem_amd/csrc/comm.cpp loads RCCL by name at its first call (dlopen "librccl.so") and uses ten entry points.  A
// GPU box of this pool has ONE GPU and RCCL refuses two ranks on one device, so the branch of OHXAllGatherOH that a real
// multi-GPU job with unequal shards takes - a group of ncclSend / ncclRecv per peer, comm.cpp "the direct exchange" - had
// never executed.  With this library first on LD_LIBRARY_PATH the ranks of tests/test_comm_mock.py (processes sharing the
// one GPU) run that very code: the shard arithmetic, the offsets into d_full, the group, the device-to-device copy of a
// rank's own rows.  What it cannot show is anything about RCCL itself (its ordering on streams, xGMI, performance).
//
// How it moves data: a message is a file in a directory both ranks see (named in the unique id): the sender waits for its
// stream, copies device -> host and writes m_<src>_<dst>_<seq>; the receiver polls for that name, reads, copies host ->
// device.  Inside ncclGroupStart / ncclGroupEnd operations are queued and run at the end - all sends, then all
// receives - which is what makes a symmetric exchange between ranks that each send first free of deadlock.
#include <hip/hip_runtime_api.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
struct ncclComm {
  int nranks = 0, rank = 0;
  std::string dir;
  std::vector<unsigned> sent, received;   // per peer: messages so far
};
typedef ncclComm* ncclComm_t;

enum { kOk = 0, kUnhandled = 1, kSystem = 2, kInvalidArgument = 4 };

struct Op {
  bool send;
  void* buf;
  size_t count;
  int peer;
  ncclComm_t comm;
  hipStream_t stream;
};
static thread_local std::vector<Op> g_queue;
static thread_local int g_depth = 0;

static std::string name_of(const ncclComm& c, int src, int dst, unsigned seq) {
  return c.dir + "/m_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(seq);
}

static ncclResult_t run(const Op& op) {
  ncclComm& c = *op.comm;
  const size_t bytes = op.count * sizeof(float);
  std::vector<char> host(bytes);
  if (op.send) {
    if (hipStreamSynchronize(op.stream) != hipSuccess) return kUnhandled;            // what the stream wrote is there
    if (bytes && hipMemcpy(host.data(), op.buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandled;
    const std::string final_name = name_of(c, c.rank, op.peer, c.sent[(size_t)op.peer]++), tmp = final_name + ".part";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return kSystem;
    const bool ok = fwrite(host.data(), 1, bytes, f) == bytes;
    fclose(f);
    if (!ok || rename(tmp.c_str(), final_name.c_str()) != 0) return kSystem;
    return kOk;
  }
  const std::string want = name_of(c, op.peer, c.rank, c.received[(size_t)op.peer]++);
  const auto give_up = std::chrono::steady_clock::now() + std::chrono::seconds(120);
  struct stat st;
  while (stat(want.c_str(), &st) != 0) {
    if (std::chrono::steady_clock::now() > give_up) return kSystem;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  if ((size_t)st.st_size != bytes) return kInvalidArgument;          // the peer sent another count than this rank expects
  FILE* f = fopen(want.c_str(), "rb");
  if (!f) return kSystem;
  const bool ok = fread(host.data(), 1, bytes, f) == bytes;
  fclose(f);
  unlink(want.c_str());
  if (!ok) return kSystem;
  if (hipStreamSynchronize(op.stream) != hipSuccess) return kUnhandled;
  if (bytes && hipMemcpy(op.buf, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return kUnhandled;
  return kOk;
}

static ncclResult_t flush() {
  std::vector<Op> q;
  q.swap(g_queue);
  for (int pass = 0; pass < 2; ++pass)                                 // every send, then every receive
    for (const Op& op : q)
      if (op.send == (pass == 0)) {
        const ncclResult_t rc = run(op);
        if (rc != kOk) return rc;
      }
  return kOk;
}

static ncclResult_t post(const Op& op) {
  if (op.comm == nullptr || op.peer < 0 || op.peer >= op.comm->nranks || op.peer == op.comm->rank) return kInvalidArgument;
  g_queue.push_back(op);
  return g_depth > 0 ? kOk : flush();
}

__attribute__((visibility("default"))) ncclResult_t ncclGetVersion(int* v) {
  if (v) *v = 0;              // "no RCCL": a caller that prints the version prints 0
  return kOk;
}
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t rc) {
  switch (rc) {
    case kOk: return "no error (mock RCCL)";
    case kSystem: return "mock RCCL: file exchange failed or a peer never sent";
    case kInvalidArgument: return "mock RCCL: invalid argument or the peer's count differs";
    default: return "mock RCCL: HIP call failed";
  }
}
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return kInvalidArgument;
  const char* base = getenv("OHX_MOCK_RCCL_DIR");
  char path[128];
  snprintf(path, sizeof path, "%s/ohx_mock_rccl_%d_%lld", base ? base : "/tmp", (int)getpid(),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  memset(id->internal, 0, sizeof id->internal);
  memcpy(id->internal, path, strlen(path));
  return mkdir(path, 0700) == 0 ? kOk : kSystem;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return kInvalidArgument;
  ncclComm* c = new ncclComm();
  c->nranks = nranks;
  c->rank = rank;
  c->dir = std::string(id.internal, strnlen(id.internal, sizeof id.internal));
  c->sent.assign((size_t)nranks, 0u);
  c->received.assign((size_t)nranks, 0u);
  *out = c;
  return kOk;
}
__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t c) {
  delete c;
  return kOk;
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() {
  ++g_depth;
  return kOk;
}
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return kInvalidArgument;
  return --g_depth == 0 ? flush() : kOk;
}
__attribute__((visibility("default"))) ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer,
                                                            ncclComm_t comm, hipStream_t stream) {
  if (type != 7) return kInvalidArgument;         // ncclFloat: all this library's caller sends
  return post(Op{true, const_cast<void*>(buf), count, peer, comm, stream});
}
__attribute__((visibility("default"))) ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer,
                                                            ncclComm_t comm, hipStream_t stream) {
  if (type != 7) return kInvalidArgument;
  return post(Op{false, buf, count, peer, comm, stream});
}
__attribute__((visibility("default"))) ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type,
                                                                 ncclComm_t comm, hipStream_t stream) {
  if (type != 7 || comm == nullptr) return kInvalidArgument;
  float* full = static_cast<float*>(recv);
  float* mine = full + (size_t)comm->rank * count;
  if (hipStreamSynchronize(stream) != hipSuccess) return kUnhandled;
  if (send != mine && count && hipMemcpy(mine, send, count * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) return kUnhandled;
  ncclGroupStart();
  for (int q = 0; q < comm->nranks; ++q) {
    if (q == comm->rank) continue;
    ncclResult_t rc = ncclSend(mine, count, type, q, comm, stream);
    if (rc == kOk) rc = ncclRecv(full + (size_t)q * count, count, type, q, comm, stream);
    if (rc != kOk) {
      g_queue.clear();
      --g_depth;
      return rc;
    }
  }
  return ncclGroupEnd();
}

}  // extern "C"
