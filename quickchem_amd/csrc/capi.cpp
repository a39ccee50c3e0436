// C ABI of libohxgb.so (include/ohxgb.h): the XGBoost C-API symbols that
// QuickChem's Shared/xgb_fortran_api.F90 binds, served by the gfx950 kernels,
// plus the device-resident and fused entry points.
//
// Host logic only: handle bookkeeping, model load/save, lazy upload of the
// flattened booster to HBM, staging and launches.  No prediction arithmetic
// happens on the CPU anywhere in this library.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <atomic>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/ohxgb.h"
#include "flatten.hpp"
#include "forest.hpp"
#include "kernels.hpp"

using namespace ohx;

namespace {

thread_local std::string g_last_error;

void set_error(const std::string& m) { g_last_error = m; }

}  // namespace

namespace ohx {
void set_last_error(const std::string& m) { g_last_error = m; }   // for comm.cpp
}

namespace {

#define API_BEGIN() try {
#define API_END()                                   \
  }                                                 \
  catch (const std::exception& e) {                 \
    set_error(e.what());                            \
    return -1;                                      \
  }                                                 \
  catch (...) {                                     \
    set_error("unknown error");                     \
    return -1;                                      \
  }                                                 \
  return 0;

#define HIP_CHECK(expr)                                                                            \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) throw OhxError(std::string(#expr) + " failed: " + hipGetErrorString(e_)); \
  } while (0)

// Handles are checked against the set of live objects, never by reading the object: the reference frees its
// DMatrix every step (OH_GridCompMod.F90:377), so a stale handle points at memory the allocator has long
// handed to someone else.
struct HandleRegistry {
  std::mutex mu;
  std::unordered_set<const void*> live;
  void add(const void* p) {
    std::lock_guard<std::mutex> g(mu);
    live.insert(p);
  }
  bool has(const void* p) {
    std::lock_guard<std::mutex> g(mu);
    return live.count(p) != 0;
  }
  bool remove(const void* p) {
    std::lock_guard<std::mutex> g(mu);
    return live.erase(p) != 0;
  }
};
HandleRegistry g_dmats, g_boosters;

// ------------------------------------------------------------------ device

struct DeviceInfo {
  int ordinal = -1;
  int num_cus = 0;
};

int device_count_or_throw() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw OhxError(std::string("libohxgb needs a HIP device and has no CPU fallback: hipGetDeviceCount: ") +
                   (e != hipSuccess ? hipGetErrorString(e) : "0 devices"));
  return n;
}

DeviceInfo use_device(int ordinal) {
  const int n = device_count_or_throw();
  if (ordinal < 0) {
    int cur = 0;
    HIP_CHECK(hipGetDevice(&cur));
    ordinal = cur;
  }
  if (ordinal >= n) throw OhxError("HIP device ordinal " + std::to_string(ordinal) + " out of range");
  HIP_CHECK(hipSetDevice(ordinal));
  DeviceInfo d;
  d.ordinal = ordinal;
  int cus = 0;
  HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ordinal));
  d.num_cus = cus > 0 ? cus : 256;
  return d;
}

// The two streams everything of this library runs on (per device, made at first use, kept for the life of the process):
// `exec` for kernels and whatever must stay in order with them, `copy` for PCIe transfers that run beside them.  None of
// the library's own work goes to the null stream, and no object has streams of its own: every stream a process has ever
// used is a hardware queue the GPU's scheduler keeps mapped, and a GPU shared by several MPI ranks - how GEOS runs -
// time-slices queues once there are more of them than it has slots.  Six ranks with four queues each: a tick in a
// hundred took 14-37 ms instead of 1; with two each, none did (profiles/r05_sweeps.txt, "six ranks").
struct LibStreams {
  hipStream_t exec = nullptr, copy = nullptr;
  hipEvent_t caller = nullptr;      // order_behind_caller
};
LibStreams& lib_streams(int ordinal) {
  static std::mutex mu;
  static std::unordered_map<int, LibStreams> by_device;
  std::lock_guard<std::mutex> g(mu);
  LibStreams& ls = by_device[ordinal];
  if (ls.exec == nullptr) {
    int cur = 0;
    HIP_CHECK(hipGetDevice(&cur));
    if (cur != ordinal) HIP_CHECK(hipSetDevice(ordinal));
    HIP_CHECK(hipStreamCreateWithFlags(&ls.exec, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&ls.copy, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&ls.caller, hipEventDisableTiming));      // made here, under the lock: two threads may ask
    if (cur != ordinal) HIP_CHECK(hipSetDevice(cur));
  }
  return ls;
}
// A host-form call on a matrix that BORROWS the caller's HBM (OHXDMatrixCreateFromDevice): whatever the caller has
// enqueued on the legacy default stream - and so on every blocking stream, torch's default among them - to fill those
// rows must be there before the library's non-blocking stream reads them.  The null-stream launch these calls made
// until round 4 gave that order by itself.  An event on the default stream, waited for by `s`; nothing of
// the library runs on the default stream.  A producer on a NON-blocking stream of the caller's is the caller's to
// synchronise (include/ohxgb.h), or to name: the *Device forms take the stream.
void order_behind_caller(int ordinal, hipStream_t s) {
  // (one event per device, re-recorded by every such call: a wait takes the event's state as it is when the wait is
  // enqueued, so a later record does not disturb an earlier call's wait; two threads at once serialise here - the pair
  // record + wait must not interleave with another thread's)
  static std::mutex mu;
  LibStreams& ls = lib_streams(ordinal);
  std::lock_guard<std::mutex> g(mu);
  HIP_CHECK(hipEventRecord(ls.caller, nullptr));
  HIP_CHECK(hipStreamWaitEvent(s, ls.caller, 0));
}
// the current device's
hipStream_t exec_stream() {
  int cur = 0;
  HIP_CHECK(hipGetDevice(&cur));
  return lib_streams(cur).exec;
}

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void ensure(size_t count) {
    if (count <= n) return;
    release();
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T)));
    n = count;
  }
  void upload(const std::vector<T>& h) {
    ensure(h.size());
    if (h.empty()) return;
    hipStream_t s = exec_stream();
    HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));      // `h` is the caller's again
  }
};

template <class T>
struct PinnedBuf {
  T* p = nullptr;
  size_t n = 0;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void ensure(size_t count) {
    if (count <= n) return;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    n = 0;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault));
    n = count;
  }
};

// ------------------------------------------------------------------ objects

// Freed matrix buffers are parked here and handed to the next XGDMatrixCreateFromMat that fits: the
// reference creates and frees its DMatrix on every OH tick (OH_GridCompMod.F90:347,377), and a hipMalloc +
// hipFree of the whole batch per tick costs more than the prediction of a small sub-domain.  At most
// kSlots buffers are kept; OHXReleaseScratch() (or OHX_DMATRIX_POOL=0 in the environment) returns them.
struct RowPool {
  static constexpr size_t kSlots = 2;
  struct Slot {
    float* p;
    size_t n;
    int device;
  };
  std::mutex mu;
  std::vector<Slot> slots;
  bool enabled() const {
    static const bool on = [] {
      const char* e = getenv("OHX_DMATRIX_POOL");
      return !(e && e[0] == '0');
    }();
    return on;
  }
  // a buffer of at least `count` floats on `device`; *cap = what it really holds
  float* take(size_t count, int device, size_t* cap) {
    {
      std::lock_guard<std::mutex> g(mu);
      size_t best = slots.size();
      for (size_t i = 0; i < slots.size(); ++i)
        if (slots[i].device == device && slots[i].n >= count && slots[i].n <= 4 * count + (1u << 18) &&
            (best == slots.size() || slots[i].n < slots[best].n))
          best = i;
      if (best != slots.size()) {
        Slot sl = slots[best];
        slots.erase(slots.begin() + (long)best);
        *cap = sl.n;
        return sl.p;
      }
    }
    float* p = nullptr;
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(float)));
    *cap = std::max<size_t>(count, 1);
    return p;
  }
  void give(float* p, size_t cap, int device) {
    if (p == nullptr) return;
    if (enabled()) {
      std::lock_guard<std::mutex> g(mu);
      slots.push_back({p, cap, device});
      if (slots.size() <= kSlots) return;
      size_t drop = 0;                                  // keep the biggest
      for (size_t i = 1; i < slots.size(); ++i)
        if (slots[i].n < slots[drop].n) drop = i;
      p = slots[drop].p;
      device = slots[drop].device;
      slots.erase(slots.begin() + (long)drop);
    }
    (void)hipFree(p);
  }
  void release_all() {
    std::lock_guard<std::mutex> g(mu);
    for (Slot& sl : slots) (void)hipFree(sl.p);
    slots.clear();
  }
};
RowPool g_row_pool;

// The caller's host arrays, registered with the driver (ohx_register_host = 1; include/ohxgb.h part 2).
//
// The host forms of the fused calls copy some forty arrays per OH tick (OHXBoosterRun1: 37 inputs, up to 12 outputs).
// From pageable memory every one of those copies is a staged, synchronous transfer: on a GEOS-sized rank block
// (48 x 24 x 72) they cost more than the prediction (profiles/r04_ranks_per_gpu*.json).  A registered (pinned) array
// is read and written by the GPU's DMA engines directly and `hipMemcpyAsync` returns at once.  Registration costs
// about as much as ten ticks and is done once per array: (pointer, bytes) is remembered, and MAPL's import, export
// and internal pointers are the same from tick to tick (OH_GridCompMod.F90:1136,1195: MAPL_GetPointer on states that
// live as long as the run).  It is OPT-IN because it is a contract: an array that was passed while this is on must
// stay allocated until OHXUnregisterHost(array), OHXReleaseScratch() or the end of the process.  What happens to an
// array that is freed while registered is the driver's business, not this table's: ROCm pins registered ranges
// through KFD userptr objects that follow the process's page tables (a range that is unmapped is invalidated, a later
// GPU access to it faults; one that is mapped again is pinned again), so the failure is loud rather than stale - but
// it is a failure, hence the contract, and hence OHXUnregisterHost for a caller whose arrays do come and go.
// Nothing is unregistered while a copy may be in flight: the device is synchronised first.  What the driver
// refuses to register (it happens to ranges that share a page with another registration) is copied the pageable way
// as before.
struct HostRegistry {
  std::mutex mu;
  std::atomic<bool> on{false};
  struct Pin {
    size_t bytes;
    void* mapped;      // where the GPU sees the array (hipHostGetDevicePointer, asked once)
  };
  std::unordered_map<const void*, Pin> pinned;        // base -> what we registered
  std::unordered_map<const void*, size_t> refused;    // base -> bytes the driver would not take
  size_t bytes = 0;
  // true when [p, p + n) is registered afterwards; *mapped (optional) = the address the GPU reaches it at, or null
  bool want(const void* p, size_t n, void** mapped = nullptr) {
    if (mapped) *mapped = nullptr;
    if (!on.load(std::memory_order_relaxed) || p == nullptr || n == 0) return false;
    std::lock_guard<std::mutex> g(mu);
    auto it = pinned.find(p);
    if (it != pinned.end()) {
      if (it->second.bytes >= n) {
        if (mapped) *mapped = it->second.mapped;
        return true;
      }
      (void)hipDeviceSynchronize();                       // no copy of the old range may still be in flight
      (void)hipHostUnregister(const_cast<void*>(p));      // the same array, grown: again from the start
      bytes -= it->second.bytes;
      pinned.erase(it);
    }
    auto rf = refused.find(p);
    if (rf != refused.end() && rf->second == n) return false;
    if (hipHostRegister(const_cast<void*>(p), n, hipHostRegisterDefault) != hipSuccess) {
      (void)hipGetLastError();
      refused[p] = n;
      return false;
    }
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, const_cast<void*>(p), 0) != hipSuccess) {
      (void)hipGetLastError();
      dev = nullptr;
    }
    pinned[p] = Pin{n, dev};
    bytes += n;
    if (mapped) *mapped = dev;
    return true;
  }
  void release_all() {
    std::lock_guard<std::mutex> g(mu);
    if (getenv("OHX_REGISTRY_LOG") && (!pinned.empty() || !refused.empty()))
      fprintf(stderr, "[ohxgb] host registry: %zu arrays (%zu bytes) registered, %zu refused\n", pinned.size(), bytes,
              refused.size());
    if (!pinned.empty()) (void)hipDeviceSynchronize();     // DMA targets are not unpinned under a running copy
    for (auto& kv : pinned) (void)hipHostUnregister(const_cast<void*>(kv.first));
    pinned.clear();
    refused.clear();
    bytes = 0;
  }
  // one array leaves (OHXUnregisterHost): true if it was registered
  bool forget(const void* p) {
    std::lock_guard<std::mutex> g(mu);
    refused.erase(p);
    auto it = pinned.find(p);
    if (it == pinned.end()) return false;
    (void)hipDeviceSynchronize();
    (void)hipHostUnregister(const_cast<void*>(p));
    bytes -= it->second.bytes;
    pinned.erase(it);
    return true;
  }
};
HostRegistry g_host_registry;

// Host arrays <-> HBM for the host forms of the fused calls.  Arrays the registry holds are gathered into ONE launch of
// copy_arrays_kernel (the GPU reads or writes the caller's memory over PCIe itself) as long as the whole list is
// small - a rank-sized block, where forty separate copies cost forty fixed prices; big lists and arrays that are not
// registered go through hipMemcpyAsync.
// blocks of a launch of copy_arrays_kernel, over all its arrays ("ohx_copy_blocks"; 0 = as many as an array has 1 KiB
// pieces, up to 2 048 per array).  Sixty-four: a copy kernel that fills the chip with waves waiting on PCIe is slower at
// moving a rank's arrays (88 against 72 us per 3.7 MB list) and - with megabytes of reads queued on the link - keeps
// every other queue's kernels from STARTING until it is done (the command processor fetches their packets and
// arguments over the same link); a 48 x 24 x 72 tick 0.349 / 0.345 / 0.328 / 0.339 / 0.389 ms at 256 / 128 / 64 / 32 / 16
// blocks, 0.398 with a block per KiB (profiles/r05_sweeps.txt)
std::atomic<uint32_t> g_copy_blocks{64};
// "ohx_copy_engine" (process-wide): how REGISTERED host arrays cross.  kernel (default) = lists by copy_arrays_kernel (one
// launch per list; the one list that crosses under the walk by DMA, OHX_COPY_POST_BLOCKS); dma = every array by
// hipMemcpyAsync (the DMA engines: a fixed price per array, no wave on any CU); auto = OHXBoosterRun1's host form times
// its own ticks both ways and keeps the faster (CopyEngineTrial below), everything else as kernel.
// Round 5 supposed that DMA would win once ranks share the card (a copy kernel's waves, waiting on PCIe on every CU, hold
// up the other ranks' walks).  Measured (profiles/r06_ranks_per_gpu_block_48x24_engines.json, a 48 x 24 x 72 tick at 1 / 2 / 3 / 6
// ranks): copy kernels 0.290 / 0.583 / 0.678 / 1.348 ms, all by DMA 0.614 / 1.258 / 1.738 / 3.131 ms - forty fixed prices
// a tick lose at every count, auto picked "kernel" on every rank of every run, and its trial's eight DMA ticks are what
// its p99 is made of (3.8 against 1.9 ms at six ranks).  So kernel is the default and auto is there to be asked for.
enum CopyEngine : int { kCopyKernel = 0, kCopyDma = 1, kCopyAuto = 2 };
std::atomic<int> g_copy_engine{kCopyKernel};
struct HostMover {
  static constexpr size_t kKernelBytesMax = 64u << 20;
  CopyList list;
  size_t list_bytes = 0;
  hipStream_t stream;
  bool to_device;
  uint32_t blocks_override = 0;       // blocks of this mover's launches, when its owner knows better than the defaults
  bool by_dma;                        // registered arrays too go through hipMemcpyAsync (the DMA engines: no wave on a CU)
  explicit HostMover(hipStream_t s, bool h2d)
      : stream(s), to_device(h2d), by_dma(g_copy_engine.load(std::memory_order_relaxed) == kCopyDma) {}
  void add(float* dst, const float* src, size_t count) {
    if (count == 0) return;
    const void* host = to_device ? (const void*)src : (const void*)dst;
    void* mapped = nullptr;
    if (g_host_registry.want(host, count * sizeof(float), &mapped) && mapped != nullptr && !by_dma && list.count < kCopyListMax &&
        list_bytes + count * sizeof(float) <= kKernelBytesMax) {
      list.src[list.count] = to_device ? static_cast<const float*>(mapped) : src;
      list.dst[list.count] = to_device ? dst : static_cast<float*>(mapped);
      list.n[list.count] = count;
      ++list.count;
      list_bytes += count * sizeof(float);
      return;
    }
    (void)hipGetLastError();
    HIP_CHECK(hipMemcpyAsync(dst, src, count * sizeof(float), to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, stream));
  }
  // `count` floats at `offset` of a host array of `whole` floats that is registered as a whole (a k-slab of a MAPL field)
  void add_slice(float* dev, const float* host_base, size_t whole, size_t offset, size_t count) {
    if (count == 0) return;
    void* mapped = nullptr;
    if (g_host_registry.want(host_base, whole * sizeof(float), &mapped) && mapped != nullptr && !by_dma && list.count < kCopyListMax &&
        list_bytes + count * sizeof(float) <= kKernelBytesMax) {
      float* m = static_cast<float*>(mapped) + offset;
      list.src[list.count] = to_device ? m : dev;
      list.dst[list.count] = to_device ? dev : m;
      list.n[list.count] = count;
      ++list.count;
      list_bytes += count * sizeof(float);
      return;
    }
    (void)hipGetLastError();
    float* host = const_cast<float*>(host_base) + offset;
    HIP_CHECK(hipMemcpyAsync(to_device ? (void*)dev : (void*)host, to_device ? (const void*)host : (const void*)dev,
                             count * sizeof(float), to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, stream));
  }
  // a few words between HBM and pinned memory of the library's own (hipHostMalloc: the GPU reaches it at its host address),
  // riding on the list's launch; false when the list is full or empty - the caller copies them its own way then
  bool ride(void* dst, const void* src, size_t words) {
    if (list.count == 0 || list.count >= kCopyListMax) return false;
    list.src[list.count] = static_cast<const float*>(src);
    list.dst[list.count] = static_cast<float*>(dst);
    list.n[list.count] = words;
    ++list.count;
    return true;
  }
  void go() {
    // a launch that WRITES the caller's memory runs at a tick's end with nothing beside it to keep from starting, and posted
    // writes want more waves than reads do: 512 blocks (OHX_COPY_BACK_BLOCKS, read once; 0 = as many as ohx_copy_blocks).
    // A rank's tick at 64 / 128 / 256 / 512: 0.309 / 0.309 / 0.306 / 0.303 ms (profiles/r05_sweeps.txt)
    static const int back_blocks = [] { const char* e = getenv("OHX_COPY_BACK_BLOCKS"); return e ? atoi(e) : 512; }();
    uint32_t blocks = (!to_device && back_blocks > 0) ? (uint32_t)back_blocks : g_copy_blocks.load(std::memory_order_relaxed);
    if (blocks_override > 0) blocks = blocks_override;
    if (list.count == 0) return;
    HIP_CHECK(launch_copy_arrays(list, stream, blocks));
    list.count = 0;
    list_bytes = 0;
  }
};

// host -> device, through the registry when it is on (then asynchronous on `stream`), the pageable way otherwise
void upload_host_array(float* dst, const float* src, size_t count, hipStream_t stream) {
  if (count == 0) return;
  (void)g_host_registry.want(src, count * sizeof(float));
  HIP_CHECK(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyHostToDevice, stream));
}
// device -> host; the caller synchronises `stream` before it looks at dst
void download_host_array(float* dst, const float* src, size_t count, hipStream_t stream) {
  if (count == 0) return;
  (void)g_host_registry.want(dst, count * sizeof(float));
  HIP_CHECK(hipMemcpyAsync(dst, src, count * sizeof(float), hipMemcpyDeviceToHost, stream));
}

// Flag and verdict words of the matrix checks (inf scan, level-size search): one device buffer and one pinned
// host mirror per process, so that a check is launches + ONE read-back.
struct CheckScratch {
  std::mutex mu;
  int device = -1;
  DevBuf<uint32_t> d;
  PinnedBuf<uint32_t> h;
  void ensure(size_t words, int dev) {
    if (dev != device) {
      d.release();
      device = dev;
    }
    d.ensure(words);
    h.ensure(words);
  }
};
CheckScratch g_check;

struct DMatrixObj {
  ~DMatrixObj() {
    // XGBoosterPredict has waited for its kernels; OHXBoosterPredictDevice only enqueues on the caller's stream, and the
    // next XGDMatrixCreateFromMat copies into a parked buffer on the library's: not before the readers are done
    if (owned != nullptr && used_async) {
      (void)hipSetDevice(device);
      (void)hipDeviceSynchronize();
    }
    g_row_pool.give(owned, owned_cap, device);
  }
  bool used_async = false;
  uint64_t nrow = 0, ncol = 0;
  float missing = NAN;
  const float* d_data = nullptr;  // device
  float* owned = nullptr;         // set when the matrix owns its storage (XGDMatrixCreateFromMat)
  size_t owned_cap = 0;
  int device = -1;
  // OHXDMatrixSetGrid: the rows are grid rows grid_row0 .. of an (im,jm,*) grid; 0 = not said
  int grid_im = 0, grid_jm = 0;
  uint64_t grid_row0 = 0;
  bool grid_inferred = false;     // found by infer_level_size, not said by the caller
  bool grid_looked = false;       // the rows have been searched (or the caller has spoken): do not look again
  // rows in no known order: has the clustering pass been judged worth it for this matrix, and was it?
  bool cluster_decided = false, cluster_on = false;
};

struct BoosterObj {
  ~BoosterObj() {
    for (hipEvent_t e : {run1_fork, run1_slab, run1_join, run1_clear})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : run1_prep) (void)hipEventDestroy(e);
    for (hipEvent_t e : run1_walk) (void)hipEventDestroy(e);
    for (hipEvent_t e : run1_feed_events) (void)hipEventDestroy(e);
    if (train.side) (void)hipStreamDestroy(train.side);
    if (train.fork) (void)hipEventDestroy(train.fork);
    if (train.join) (void)hipEventDestroy(train.join);
    if (cluster_done) (void)hipEventDestroy(cluster_done);
    if (defer_seen) (void)hipEventDestroy(defer_seen);
  }
  TrainStreams train;              // second stream of the launch train (kernels.hpp): made when ohx_overlap_group asks for it
  Forest forest;
  bool loaded = false;
  float margin_base = 0.0f;        // Forest::margin_base() of the loaded model
  std::string margin_error;        // why it is unknown (objective this library cannot start a margin for)
  LayoutParams layout;
  std::string kernel_name = "auto";
  int device_pref = -1;
  LaunchTuning tune;
  // device state, built lazily at the first compute call
  bool uploaded = false;
  DeviceInfo dev;
  Placement placement;
  bool packed_ok = false;
  DevBuf<PackedNode> d_packed;
  DevBuf<WideNode> d_wide;
  DevBuf<SuperNode> d_super;
  DevBuf<SuperTreeHead> d_super_heads;
  bool super_ok = false;
  uint64_t super_slots = 0;
  uint64_t super_gathers[3] = {0, 0, 0};   // vector-memory instructions a wave issues to walk the whole forest once
                                        // without / with tree tops
  double super_mean_steps = 0.0;
  std::string symbol;                   // OHXBoosterKernelSymbol's answer
  DevBuf<uint32_t> d_roots;
  DevBuf<uint32_t> d_flags;
  uint32_t ring_reruns_seen = 0;        // of d_flags[1], as last read back; the first sighting is said on stderr
  DevBuf<float> d_leaves;               // small batches (kernels.hpp PredictArgs::leaf_buf): [tile][tree][lane] leaf values
  DevBuf<uint32_t> d_defer;             // deferred rows (kernels.hpp PredictArgs::defer_list): the count, then the list
  // how many rows of the last batch held missing values (read back behind the batch, looked at before the next one)
  PinnedBuf<uint32_t> h_defer_count;
  PinnedBuf<uint32_t> h_flags;          // d_flags[0..1] as the last copy-back kernel of a host-form tick left them
  hipEvent_t defer_seen = nullptr;
  bool defer_pending = false;
  uint64_t defer_last_nrow = 0;
  bool defer_too_many = false;
  DevBuf<float> d_pred;
  PinnedBuf<float> h_pred;
  // the library's two streams on the booster's device (lib_streams; not the booster's to destroy).  Host forms of the
  // fused calls: PCIe copies on s_copy beside the kernels on s_exec.  OH Run1's device form: the slab count, and the
  // streaming kernels of the pieces that are not being walked, on s_copy beside the caller's stream
  hipStream_t s_copy = nullptr, s_exec = nullptr;
  hipEvent_t run1_fork = nullptr, run1_slab = nullptr, run1_join = nullptr, run1_clear = nullptr;
  std::vector<hipEvent_t> run1_prep, run1_walk, run1_feed_events;
  // ohx_copy_engine = auto: the host form of Run1 on this booster decides by its own clock.  After two ticks of warm-up,
  // eight ticks of each engine in runs of four (K K K K D D D D K K K K D D D D); the lower median wins; every 512 ticks
  // the trial is run again (ranks come and go); a block of another size starts over.  Both engines give the same bits.
  struct CopyEngineTrial {
    size_t vol = 0;
    unsigned tick = 0;                 // since the trial (re)started
    int choice = -1;                   // -1 = trying
    unsigned trials = 0, picked_dma = 0;
    std::vector<double> seconds[2];
    static constexpr unsigned kWarm = 2, kRun = 4, kEach = 8, kAgain = 512;
    bool dma_now(size_t v) {
      if (v != vol) { vol = v; tick = 0; choice = -1; seconds[0].clear(); seconds[1].clear(); }
      if (choice >= 0 && tick >= kAgain) { tick = kWarm; choice = -1; seconds[0].clear(); seconds[1].clear(); }
      if (choice >= 0) return choice == 1;
      if (tick < kWarm) return false;
      return ((tick - kWarm) / kRun) % 2 == 1;
    }
    void report(bool dma, double s) {
      const unsigned t = tick++;
      if (choice >= 0 || t < kWarm) return;
      seconds[dma ? 1 : 0].push_back(s);
      if (seconds[0].size() >= kEach && seconds[1].size() >= kEach) {
        double med[2];
        for (int e = 0; e < 2; ++e) {
          std::sort(seconds[e].begin(), seconds[e].end());
          med[e] = seconds[e][seconds[e].size() / 2];
        }
        choice = med[1] < med[0] ? 1 : 0;
        ++trials;
        picked_dma += (unsigned)choice;
        if (getenv("OHX_DEBUG"))
          fprintf(stderr, "[libohxgb] copy engine trial %u on a block of %zu gridcells: median tick %.1f us by copy kernels, %.1f us by DMA -> %s\n",
                  trials, vol, med[0] * 1e6, med[1] * 1e6, choice ? "dma" : "kernel");
      }
    }
  } copy_trial;
  PinnedBuf<int32_t> h_slab;
  std::vector<DevBuf<float>> d_stage;  // fused host path: per-field staging
  DevBuf<float> d_stage_out, d_stage_margin;
  // Run1: engineered features, OH_ML and the slab result stay in HBM between the steps
  DevBuf<float> d_run1[10];
  DevBuf<int32_t> d_slab;
  bool slab_zeroed = false;            // d_slab's two words are zero whenever no slab count is enqueued (run1_device) ...
  hipStream_t slab_zeroed_on = nullptr; // ... by a memset on this stream, with run1_clear recorded behind it
  // clustering pass for rows in no known order (kernels.hip)
  DevBuf<uint32_t> d_cluster_keys[2], d_cluster_vals[2], d_cluster_small;
  DevBuf<uint8_t> d_cluster_temp;
  PinnedBuf<uint32_t> h_cluster_small;
  // the walk that reads the permutation out of the buffers above is asynchronous: recorded behind it, waited for by
  // the next clustering pass on this booster whatever stream that one runs on
  hipEvent_t cluster_done = nullptr;
  bool cluster_in_flight = false;
  std::vector<DevBuf<float>> d_run1_stage;
};

DMatrixObj* as_dmat(DMatrixHandle h) {
  if (h == nullptr || !g_dmats.has(h)) throw OhxError("DMatrix handle is invalid or has been freed");
  return static_cast<DMatrixObj*>(h);
}

BoosterObj* as_booster(BoosterHandle h) {
  if (h == nullptr || !g_boosters.has(h)) throw OhxError("Booster handle is invalid or has been freed");
  return static_cast<BoosterObj*>(h);
}

// Tree tops (and the ring kernels) pay where the texture addresser is the bound: deep trees.  Measured on the MI355X
// with the OH recipe's boosters (profiles/r02_sweeps.txt): 9 steps per tree (depth 18) 3.4 % faster with tops, 5 steps
// (depth 10) 10 % slower, 3 steps (depth 6) 27 % slower.
constexpr double kTreeTopsMinMeanSteps = 7.0;
constexpr double kRingMinMeanSteps = 5.0;

KernelKind pick_kernel(const BoosterObj& b) {
  const std::string& k = b.kernel_name;
  if (k == "wide") return KernelKind::Wide;
  if (b.super_ok) {
    if (k == "super1") return KernelKind::Super1;
    if (k == "super3") return KernelKind::Super3;
    if (k == "super4") return KernelKind::Super4;
    // forests of some depth walk fastest with their tree tops resident in LDS (kernels.hip, the ring kernels), against
    // the tile kernel: 9 steps per tree (the OH booster, depth 18) -20 %, 7 steps -16 %, 6 steps -6 %, 5 steps -3 %,
    // 4 steps +2 %, 3 steps +29 % (profiles/r04_sweeps.txt)
    if (k == "ring" || (k == "auto" && b.forest.num_feature == 27 && b.super_mean_steps >= kRingMinMeanSteps))
      return KernelKind::Ring;
    if (k == "super2" || k == "auto") return KernelKind::Super2;
  }
  if (!b.packed_ok) return KernelKind::Wide;
  if (k == "packed1") return KernelKind::Packed1;
  if (k == "packed4") return KernelKind::Packed4;
  return KernelKind::Packed2;
}

// Vector-memory instructions a wave issues to walk the whole forest once (walk_super).  With tree tops: per tree
// one coalesced load of its top (steps 1-3) and one gather for every step after the third, of at least four
// steps.  Without: a gather per step, less the first step of the trees whose start nodes sit in the kernels'
// LDS table.
uint64_t count_super_gathers(const SuperForest& sf, int mode) {   // 0 plain walk, 1 tree tops, 2 ring kernels
  uint64_t n = 0;
  for (size_t t = 0; t < sf.heads.size(); ++t) {
    const uint32_t steps = sf.heads[t].steps;
    if (mode == 2) n += (steps < 4u ? 4u : steps) - 4u;        // steps 1-4 from LDS
    else n += mode == 1 ? (steps < 4u ? 4u : steps) - 2u : steps - (t < kFirstStepTrees && steps ? 1u : 0u);
  }
  // ring: a group of four trees is staged with 11 wave loads, once per block of 16 waves
  if (mode == 2) n += (sf.heads.size() * 11 + 63) / 64;
  return n;
}

double mean_super_steps(const SuperForest& sf) {
  double n = 0;
  for (const SuperTreeHead& h : sf.heads) n += h.steps;
  return sf.heads.empty() ? 0.0 : n / (double)sf.heads.size();
}
bool use_tree_tops(const LaunchTuning& tune, double mean_steps) {
  return tune.tree_tops < 0 ? mean_steps >= kTreeTopsMinMeanSteps : tune.tree_tops != 0;
}

bool wants_super(const std::string& k) {
  return k == "auto" || k == "ring" || (k.size() == 6 && k.compare(0, 5, "super") == 0);
}

void invalidate_device_state(BoosterObj& b) {
  b.uploaded = false;
  b.d_packed.release();
  b.d_wide.release();
  b.d_super.release();
  b.d_super_heads.release();
  b.super_ok = false;
  b.d_roots.release();
}

void ensure_wide(BoosterObj& b) {
  if (b.d_wide.p) return;
  std::vector<WideNode> wide = emit_wide(b.forest, b.placement);
  b.d_wide.upload(wide);
}

void ensure_train_stream(BoosterObj& b);

void ensure_uploaded(BoosterObj& b) {
  if (!b.loaded) throw OhxError("the booster holds no model: call XGBoosterLoadModel first");
  if (b.uploaded) {
    HIP_CHECK(hipSetDevice(b.dev.ordinal));
    return;
  }
  b.dev = use_device(b.device_pref);
  b.placement = place_forest(b.forest, b.layout);
  b.packed_ok = packed_format_fits(b.forest, b.placement);
  b.d_roots.upload(b.placement.roots);
  if (wants_super(b.kernel_name)) {
    SuperForest sf;
    b.super_ok = emit_super(b.forest, &sf) && sf.nodes.size() * sizeof(SuperNode) < 0xFFFFFFF0ull;
    if (b.super_ok) {
      b.super_slots = sf.nodes.size();
      b.super_gathers[0] = count_super_gathers(sf, 0);
      b.super_gathers[1] = count_super_gathers(sf, 1);
      b.super_gathers[2] = count_super_gathers(sf, 2);
      b.super_mean_steps = mean_super_steps(sf);
      b.d_super.upload(sf.nodes);
      b.d_super_heads.upload(sf.heads);
    }
  }
  if (b.packed_ok && !(b.super_ok && wants_super(b.kernel_name))) {
    std::vector<PackedNode> packed = emit_packed(b.forest, b.placement, nullptr);
    b.d_packed.upload(packed);
  }
  if (pick_kernel(b) == KernelKind::Wide) ensure_wide(b);
  b.d_flags.ensure(3);      // [0] kFlag* bits, [1] ring re-runs counted on the device, [2] the train a block last gave up in
  {
    LibStreams& ls = lib_streams(b.dev.ordinal);
    b.s_exec = ls.exec;
    b.s_copy = ls.copy;
  }
  HIP_CHECK(hipMemsetAsync(b.d_flags.p, 0, 3 * sizeof(uint32_t), b.s_exec));
  HIP_CHECK(hipStreamSynchronize(b.s_exec));
  b.ring_reruns_seen = 0;
  ensure_train_stream(b);
  b.uploaded = true;
}

// the launch train's second stream exists only for a booster that was told to use it (ohx_overlap_group >= 2: an
// experiment knob): a stream is a hardware queue for the life of the process (LibStreams)
void ensure_train_stream(BoosterObj& b) {
  b.tune.train = nullptr;
  if (b.tune.overlap_group < 2) return;
  if (!b.train.side) {
    HIP_CHECK(hipStreamCreateWithFlags(&b.train.side, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&b.train.fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&b.train.join, hipEventDisableTiming));
  }
  b.tune.train = &b.train;
}

DeviceForest device_forest(const BoosterObj& b) {
  DeviceForest d;
  d.packed = b.d_packed.p;
  d.wide = b.d_wide.p;
  d.roots = b.d_roots.p;
  d.super = b.d_super.p;
  d.super_heads = b.d_super_heads.p;
  d.packed_bytes = (uint32_t)(b.d_packed.n * sizeof(PackedNode));
  d.super_bytes = (uint32_t)(b.d_super.n * sizeof(SuperNode));
  d.num_trees = (uint32_t)b.forest.trees.size();
  d.num_feature = b.forest.num_feature;
  d.base_score = b.margin_base;
  d.tree_tops = (b.super_ok && use_tree_tops(b.tune, b.super_mean_steps)) ? 1u : 0u;
  return d;
}

void tree_range(const BoosterObj& b, unsigned ntree_limit, uint32_t* t0, uint32_t* t1) {
  const uint32_t T = (uint32_t)b.forest.trees.size();
  *t0 = 0;
  *t1 = (ntree_limit == 0 || ntree_limit > T) ? T : ntree_limit;
}

void check_predict_options(const BoosterObj& b, int option_mask, bool* pred_leaf) {
  *pred_leaf = false;
  if (option_mask == 0 || option_mask == 1) {
    if (!b.margin_error.empty()) throw OhxError(b.margin_error);
    if (option_mask == 0 && !objective_is_identity(b.forest.objective))
      throw OhxError("objective '" + b.forest.objective +
                     "' needs a prediction transform this library does not implement; "
                     "use option_mask = 1 (margin) or a reg:squarederror model");
    return;
  }
  if (option_mask == 16) {
    *pred_leaf = true;
    return;
  }
  throw OhxError("XGBoosterPredict: option_mask " + std::to_string(option_mask) +
                 " is not supported (0 = value, 1 = margin, 16 = leaf index)");
}

void check_columns(const BoosterObj& b, uint64_t ncol) {
  if (ncol > b.forest.num_feature)
    throw OhxError("Number of columns does not match number of features in booster (" + std::to_string(ncol) +
                   " vs. " + std::to_string(b.forest.num_feature) + ")");
}

void judge_flags(BoosterObj& b, const uint32_t flags[2], hipStream_t stream);

void raise_flag_errors(BoosterObj& b, hipStream_t stream) {
  uint32_t flags[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(flags, b.d_flags.p, sizeof(flags), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  judge_flags(b, flags, stream);
}

// `flags` = the first two words of d_flags as read back behind the work on `stream`, which the caller has waited for
void judge_flags(BoosterObj& b, const uint32_t flags[2], hipStream_t stream) {
  // a ring block that timed out is not an error: its rows were walked again by the tile kernel on the stream, behind
  // the train (kernels.hip launch_rows_ring); worth a line on stderr the first time, and a counter from then on
  if (flags[1] != b.ring_reruns_seen) {
    if (b.ring_reruns_seen == 0)
      fprintf(stderr, "[libohxgb] warning: a block of the ring kernel gave up waiting for another; the batch was predicted "
                      "again by the tile kernel (same results, slower).  Counted in OHXBoosterRingReruns; "
                      "ohx_kernel=super2 avoids the ring kernels.\n");
    b.ring_reruns_seen = flags[1];
  }
  if (flags[0] != 0) {
    HIP_CHECK(hipMemsetAsync(b.d_flags.p, 0, sizeof(uint32_t), stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (flags[0] & kFlagInfInput) throw OhxError("Input data contains `inf` or `nan`");
    throw OhxError("the predict kernel reported error flags " + std::to_string(flags[0]));
  }
}

// A parsed model becomes the booster's: what the readers forgave goes to stderr (as xgboost's LOG(WARNING)
// does), the margin every prediction starts from is fixed here.
void adopt_model(BoosterObj& b, Forest&& f) {
  f.validate();
  for (const std::string& w : f.warnings) fprintf(stderr, "[libohxgb] warning: model file: %s\n", w.c_str());
  invalidate_device_state(b);
  b.forest = std::move(f);
  b.loaded = true;
  b.margin_error.clear();
  try {
    b.margin_base = b.forest.margin_base();
  } catch (const OhxError& e) {
    b.margin_base = b.forest.base_score;
    b.margin_error = e.what();       // leaf indices (option_mask 16) still work
  }
}

bool ends_with(const std::string& s, const std::string& suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

// A caller that does not say which grid its rows were gathered from (OHXDMatrixSetGrid) - the
// reference's own call sequence - still shows the size of a level: the gather stacks levels
// (OH_GridCompMod.F90:309-345) and its first column is a 2-D field (LAT, :313), so that column repeats
// bit for bit with the level size as period.  Knowing the level size alone the kernels take runs of
// 8 cells x 8 levels per wave, which measures within 1 % of the full 4x4x4 bricks (34.8 vs 34.6 ms per
// C360 step; 40.9 ms without).  Speed only: any period gives a valid tiling of the rows.
// The search runs once per matrix, at its first predict (or OHXDMatrixGetGrid / OHXDMatrixInferGrid), and
// not at all when the caller has named the grid by then (OHXDMatrixSetGrid).
constexpr uint64_t kMinLevel = 4096, kMaxLevels = 1024;

uint32_t level_candidates(const DMatrixObj& d) {
  if (d.ncol < 1 || d.nrow < 2 * kMinLevel) return 0;
  return (uint32_t)std::min<uint64_t>(kMaxLevels, d.nrow / kMinLevel);       // kmax: candidates k = kmax .. 2
}

// LAT is the first column of the OH gather; the last one (SZA) and the 2-D fields in between
// (GMISTRATO3, ALBUV, :334-335) serve a caller whose first column is something else
PeriodColumns level_columns(const DMatrixObj& d) { return PeriodColumns{{0u, (uint32_t)d.ncol - 1u, 21u, 22u}}; }

// verdict[c * (kmax - 1) + x]: 0 = column c repeats with period nrow / (kmax - x)
void adopt_level_size(DMatrixObj& d, const uint32_t* verdict, uint32_t kmax) {
  d.grid_looked = true;
  for (uint32_t c = 0; c < 4; ++c)
    for (uint32_t x = 0; x + 1 < kmax; ++x) {                    // ascending periods: the smallest that holds
      const uint64_t period = d.nrow / (kmax - x);
      if (verdict[(size_t)c * (kmax - 1) + x] == 0 && period <= 0x7FFFFFFFull) {
        d.grid_im = (int)period;
        d.grid_jm = 1;
        d.grid_row0 = 0;
        d.grid_inferred = true;
        return;
      }
    }
}

// What the search found for matrices of a shape seen before.  The reference creates and frees its DMatrix on every
// OH tick (OH_GridCompMod.F90:347,377) with the same N x 27 every time: the launches and the host wait of the search
// are paid at the first tick only.  Speed only - any period is a valid tiling of the rows - so a later matrix of the
// same shape from another grid is tiled with the remembered level size, not wrongly.  OHXDMatrixInferGrid always looks.
struct LevelSizeCache {
  std::mutex mu;
  struct Entry {
    uint64_t nrow, ncol;
    int period;      // 0 = looked and found none
    unsigned hits;   // since the last real search
  };
  std::vector<Entry> seen;
  // A verdict - either way - is only taken on trust fifteen times in a row; the sixteenth matrix of the shape is
  // searched again (one read-back), so that one odd matrix - shuffled rows, a constant first column - cannot decide
  // how every later matrix of its shape is tiled for the life of the process.
  bool find(uint64_t nrow, uint64_t ncol, int* period) {
    std::lock_guard<std::mutex> g(mu);
    for (Entry& e : seen)
      if (e.nrow == nrow && e.ncol == ncol) {
        if (++e.hits >= 16) return false;
        *period = e.period;
        return true;
      }
    return false;
  }
  void put(uint64_t nrow, uint64_t ncol, int period) {
    std::lock_guard<std::mutex> g(mu);
    for (Entry& e : seen)
      if (e.nrow == nrow && e.ncol == ncol) {
        e.period = period;
        e.hits = 0;
        return;
      }
    if (seen.size() >= 64) seen.erase(seen.begin());
    seen.push_back({nrow, ncol, period, 0u});
  }
};
LevelSizeCache g_level_sizes;

bool stream_capturing(hipStream_t stream);                 // defined with the deferred rows below
[[noreturn]] void refuse_in_capture(const char* what);

// Launches + one read-back on `stream`; waits for it (the rows must be there) - unless a matrix of this shape has
// been searched before and `use_cache` allows taking its verdict.
void infer_level_size(DMatrixObj& d, hipStream_t stream, bool use_cache = false) {
  d.grid_looked = true;
  const uint32_t kmax = level_candidates(d);
  if (kmax < 2) return;
  int period = 0;
  if (use_cache && g_level_sizes.find(d.nrow, d.ncol, &period)) {
    if (period > 0) {
      d.grid_im = period;
      d.grid_jm = 1;
      d.grid_row0 = 0;
      d.grid_inferred = true;
    }
    return;
  }
  {
    std::lock_guard<std::mutex> g(g_check.mu);
    const size_t words = (size_t)4 * (kmax - 1);
    g_check.ensure(words, d.device);
    HIP_CHECK(hipMemsetAsync(g_check.d.p, 0, words * sizeof(uint32_t), stream));
    HIP_CHECK(launch_detect_period(d.d_data, d.nrow, (uint32_t)d.ncol, level_columns(d), 4, kmax, g_check.d.p, stream));
    HIP_CHECK(hipMemcpyAsync(g_check.h.p, g_check.d.p, words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    adopt_level_size(d, g_check.h.p, kmax);
  }
  g_level_sizes.put(d.nrow, d.ncol, d.grid_inferred ? d.grid_im : 0);
}

// Rows nobody has described and in which no level size was found: group them by the decisions they take at
// the top of the booster's first trees (kernels.hip) and let every wave take 64 rows of one group.  Returns the
// permutation to walk through, or nullptr for "as they come".  Judged once per matrix (one wait on the stream,
// like the level-size search): rows that mostly agree with their predecessor are in some useful order already.
const uint32_t* cluster_rows(BoosterObj& b, DMatrixObj& d, const PredictArgs& a, bool pred_leaf, KernelKind kind,
                             hipStream_t stream) {
  constexpr uint64_t kMinRows = 1u << 18;
  if (b.tune.cluster == 0 || pred_leaf || d.grid_im != 0 || !b.super_ok || kind == KernelKind::Wide) return nullptr;
  if (d.nrow > 0xFFFFFFF0ull || d.ncol > b.forest.num_feature || b.forest.num_feature > 32) return nullptr;
  if (b.tune.cluster < 0 && (d.nrow < kMinRows || (d.cluster_decided && !d.cluster_on))) return nullptr;
  ClusterArgs c;
  c.rows = d.d_data;
  c.nrow = d.nrow;
  c.ncol = (uint32_t)d.ncol;
  c.missing = d.missing;
  c.ntrees = (uint32_t)std::min<size_t>((size_t)std::max(b.tune.cluster_trees, 1), b.forest.trees.size());
  c.nsteps = (uint32_t)std::max(b.tune.cluster_steps, 1);
  c.zorder = b.tune.cluster_zorder ? 1u : 0u;
  c.ntrees = std::min<uint32_t>(c.ntrees, 4u);
  while (cluster_key_bits(c) > 32u && c.nsteps > 1) --c.nsteps;
  while (cluster_key_bits(c) > 32u && c.ntrees > 1) --c.ntrees;
  const unsigned sort_bits = cluster_key_bits(c);
  // an earlier walk of this booster may still be reading its permutation out of these buffers (another stream,
  // another matrix): the keys of this pass go in behind it
  if (b.cluster_in_flight) HIP_CHECK(hipStreamWaitEvent(stream, b.cluster_done, 0));
  for (int q = 0; q < 2; ++q) {
    b.d_cluster_keys[q].ensure(d.nrow);
    b.d_cluster_vals[q].ensure(d.nrow);
  }
  b.d_cluster_small.ensure(9);
  b.h_cluster_small.ensure(9);
  c.keys = b.d_cluster_keys[0].p;
  c.vals = b.d_cluster_vals[0].p;
  c.agree = b.d_cluster_small.p;
  HIP_CHECK(hipMemsetAsync(c.agree, 0, 9 * sizeof(uint32_t), stream));
  HIP_CHECK(launch_cluster_keys(device_forest(b), c, b.dev.num_cus, stream));
  if (b.tune.cluster < 0 && !d.cluster_decided) {
    if (stream_capturing(stream)) refuse_in_capture("judge the order of the rows (it waits for the stream)");
    HIP_CHECK(hipMemcpyAsync(b.h_cluster_small.p, c.agree, 9 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    // observed agreement between neighbours in row order against what rows in random order would show
    // (sum of squared shares of the eight outcomes): 0 = as good as shuffled, 1 = every row like the one before
    const uint32_t* h = b.h_cluster_small.p;
    const double n = (double)d.nrow, observed = (double)h[0] / n;
    double by_chance = 0.0;
    for (int v = 1; v <= 8; ++v) by_chance += ((double)h[v] / n) * ((double)h[v] / n);
    const double order = by_chance < 0.999 ? (observed - by_chance) / (1.0 - by_chance) : 0.0;
    d.cluster_decided = true;
    d.cluster_on = order < 0.33;
    if (getenv("OHX_DEBUG"))
      fprintf(stderr, "[libohxgb] cluster verdict: %.1f %% of the rows take the first three decisions of the row before them, "
              "%.1f %% would by chance: order %.2f -> %s\n", 100.0 * observed, 100.0 * by_chance, order,
              d.cluster_on ? "cluster" : "leave");
    if (!d.cluster_on) return nullptr;
  }
  size_t temp_bytes = 0;
  HIP_CHECK(sort_pairs_u32(nullptr, &temp_bytes, c.keys, b.d_cluster_keys[1].p, c.vals, b.d_cluster_vals[1].p, d.nrow,
                           sort_bits, stream, nullptr));
  b.d_cluster_temp.ensure(temp_bytes);
  uint32_t* sorted = nullptr;
  HIP_CHECK(sort_pairs_u32(b.d_cluster_temp.p, &temp_bytes, c.keys, b.d_cluster_keys[1].p, c.vals, b.d_cluster_vals[1].p,
                           d.nrow, sort_bits, stream, &sorted));
  return sorted;
}

// Deferred rows (kernels.hpp PredictArgs::defer_list).  The second launch pays while few rows hold missing values
// (1e-4 of the entries: +2 % instead of +17 %); from about one row in fifty on it costs what it saves, and more beyond
// (profiles/r03_sweeps.txt).  The count of the last batch - read back behind it, never waited for - decides; the
// reference's ticks resemble each other.  Returns whether the launch will count (and maybe list) such rows.
// A small batch or slab: room for the leaves of its trees, so that they can be walked by several waves per tile
// (kernels.hpp PredictArgs::leaf_buf; only batches that can qualify: their tiles fill at most half of the chip's 20
// waves per CU).
// `im`, `jm`, `row0`: the grid the rows come from (0 = unknown): the buffer is sized for the tiles the launcher will
// really make of them - bricks over the rows' levels, or 64 consecutive rows where bricks would be mostly empty -
// not for a bound (a 48 x 24 x 72 rank block took 73 MB per booster; now 24 MB).
// `plan_only` (OHXBoosterKernelSymbolRows, a query): what a predict of this batch WOULD be handed - the buffer as it is or
// as it would be made - without making it: the launch planner only looks at the pointer being non-NULL and at the size.
float* const kPlannedBuffer = reinterpret_cast<float*>((uintptr_t)256);
void leaf_room(BoosterObj& b, uint64_t nrow, uint32_t ntree, LaunchTuning& tune, int im, int jm, uint64_t row0,
               bool plan_only = false) {
  if (tune.tree_split == 0 || !b.super_ok || nrow == 0 || nrow > (uint64_t)b.dev.num_cus * 10u * 64u + 4096u) return;
  size_t tiles = (size_t)((nrow + 63) / 64);
  if (im > 0 && jm > 0) {
    TileShape probe;
    if (tune.brick_li < 0) probe.set_grid_auto((uint32_t)im, (uint32_t)jm, row0, nrow);
    else if (tune.brick_li + tune.brick_lj + tune.brick_lk == 6)
      probe.set_grid((uint32_t)im, (uint32_t)jm, row0, nrow, (uint32_t)tune.brick_li, (uint32_t)tune.brick_lj, (uint32_t)tune.brick_lk);
    if (probe.im != 0 && probe.ntiles(nrow) > tiles) tiles = (size_t)probe.ntiles(nrow);
  }
  if (plan_only) {
    tune.leaf_buf = b.d_leaves.p != nullptr ? b.d_leaves.p : kPlannedBuffer;
    tune.leaf_words = std::max(b.d_leaves.n, tiles * 64 * (size_t)ntree);
    return;
  }
  b.d_leaves.ensure(tiles * 64 * (size_t)ntree);
  tune.leaf_buf = b.d_leaves.p;
  tune.leaf_words = b.d_leaves.n;
}

// What defer_prepare would say and hand out for this batch, touching nothing: not the pending read-back, not
// defer_too_many, no allocation, no memset on any stream (the query used to do all four, the memset on NULL).
bool defer_plan(const BoosterObj& b, uint64_t nrow, LaunchTuning& tune) {
  if (tune.defer_missing == 0 || !(tune.defer_missing > 0 || nrow >= (1u << 18))) return false;
  tune.defer_buf = b.d_defer.p != nullptr ? b.d_defer.p : reinterpret_cast<uint32_t*>(kPlannedBuffer);
  tune.defer_words = std::max(b.d_defer.n, (size_t)(nrow / 32 + 1024 + 1));
  tune.defer_count_only = (b.defer_too_many && tune.defer_missing < 0) ? 1 : 0;
  return true;
}

// Is `stream` being captured into a hipGraph?  The device forms (OHXBoosterPredictDevice, OHXBoosterPredictFieldsDevice)
// may be: what they enqueue is launches, memsets and copies on the caller's stream and nothing else, once the
// booster's buffers exist and the matrix has been looked at - i.e. after one plain call of the same shape.  What the
// host does BESIDE the launches to adapt the next call (the read-back of how many rows were left to the second launch,
// the event query that consumes it) is left out while capturing: a replay could not repeat it, and an event query is
// not allowed beside a capture.  The captured work is what the last plain call decided.
bool stream_capturing(hipStream_t stream) {
  if (stream == nullptr) return false;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st == hipStreamCaptureStatusActive;
}

[[noreturn]] void refuse_in_capture(const char* what) {
  throw OhxError(std::string("inside a stream capture the library cannot ") + what +
                 ": make one plain call of the same shape on this booster and matrix first, then capture");
}

bool defer_prepare(BoosterObj& b, uint64_t nrow, LaunchTuning& tune, hipStream_t stream) {
  if (tune.defer_missing == 0 || !(tune.defer_missing > 0 || nrow >= (1u << 18))) return false;
  const bool capturing = stream_capturing(stream);
  if (!capturing && b.defer_pending && hipEventQuery(b.defer_seen) == hipSuccess) {
    b.defer_pending = false;
    b.defer_too_many = (uint64_t)b.h_defer_count.p[0] * 50u > b.defer_last_nrow;        // more than 2 % of the rows
  }
  const uint32_t* before = b.d_defer.p;
  if (capturing && b.d_defer.n < (size_t)(nrow / 32 + 1024 + 1)) refuse_in_capture("allocate the list of rows left to the second launch");
  b.d_defer.ensure((size_t)(nrow / 32 + 1024 + 1));
  // the launchers zero the count only when they really defer (kernels.hip launch_predict: prefetch kernel, no split
  // ...); defer_look reads it back regardless, so a buffer fresh from hipMalloc must not hold garbage
  // on the stream the launches and defer_look's read-back run on: those streams are non-blocking, a NULL-stream memset
  // is not ordered before them
  if (b.d_defer.p != before) HIP_CHECK(hipMemsetAsync(b.d_defer.p, 0, sizeof(uint32_t), stream));
  tune.defer_buf = b.d_defer.p;
  tune.defer_words = b.d_defer.n;
  tune.defer_count_only = (b.defer_too_many && tune.defer_missing < 0) ? 1 : 0;
  return true;
}

void defer_look(BoosterObj& b, uint64_t nrow, hipStream_t stream) {
  if (b.defer_pending || stream_capturing(stream)) return;
  if (b.defer_seen == nullptr) HIP_CHECK(hipEventCreateWithFlags(&b.defer_seen, hipEventDisableTiming));
  b.h_defer_count.ensure(1);
  HIP_CHECK(hipMemcpyAsync(b.h_defer_count.p, b.d_defer.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipEventRecord(b.defer_seen, stream));
  b.defer_pending = true;
  b.defer_last_nrow = nrow;
}

void launch_predict_checked(BoosterObj& b, DMatrixObj& d, int option_mask, unsigned ntree_limit, float* d_out,
                            hipStream_t stream) {
  bool pred_leaf = false;
  check_predict_options(b, option_mask, &pred_leaf);
  check_columns(b, d.ncol);
  ensure_uploaded(b);
  if (d.device >= 0 && d.device != b.dev.ordinal)
    throw OhxError("the DMatrix lives on HIP device " + std::to_string(d.device) + " but the booster on device " +
                   std::to_string(b.dev.ordinal));
  KernelKind kind = pick_kernel(b);
  // nobody has said which grid the rows come from: look once (work on `stream` enqueued so far is waited for)
  // (a matrix the library copied itself - the reference's create / predict / free per tick - takes the verdict of
  // the last matrix of its shape instead of looking again)
  if (!d.grid_looked && d.grid_im == 0 && !pred_leaf && kind != KernelKind::Wide) {
    if (stream_capturing(stream)) refuse_in_capture("look for the rows' level size (it waits for the stream)");
    infer_level_size(d, stream, d.owned != nullptr);
  }
  if (pred_leaf || kind == KernelKind::Wide) ensure_wide(b);
  PredictArgs a;
  a.rows = d.d_data;
  a.nrow = d.nrow;
  a.ncol = (uint32_t)d.ncol;
  a.missing = d.missing;
  tree_range(b, ntree_limit, &a.tree_begin, &a.tree_end);
  a.out = d_out;
  a.pred_leaf = pred_leaf;
  a.flags = b.d_flags.p;
  LaunchTuning tune = b.tune;
  tune.grid_im = d.grid_im;
  tune.grid_jm = d.grid_jm;
  tune.grid_row0 = d.grid_row0;
  a.perm = cluster_rows(b, d, a, pred_leaf, kind, stream);
  if (!pred_leaf && a.perm == nullptr) leaf_room(b, d.nrow, a.tree_end - a.tree_begin, tune, tune.grid_im, tune.grid_jm, tune.grid_row0);
  // rows with missing values leave for a second, small launch instead of slowing their whole wave down (big batches)
  const bool deferring = a.perm == nullptr && !pred_leaf && d.ncol == 27 && defer_prepare(b, d.nrow, tune, stream);
  HIP_CHECK(launch_predict(kind, device_forest(b), a, b.dev.num_cus, stream, tune));
  if (deferring) defer_look(b, d.nrow, stream);
  if (a.perm != nullptr) {
    if (b.cluster_done == nullptr) HIP_CHECK(hipEventCreateWithFlags(&b.cluster_done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(b.cluster_done, stream));
    b.cluster_in_flight = true;
  }
}

}  // namespace

// =================================================================== C ABI

#pragma GCC visibility push(default)
extern "C" {

const char* XGBGetLastError(void) { return g_last_error.c_str(); }

int OHXDeviceCount(int* out) {
  API_BEGIN();
  if (out == nullptr) throw OhxError("OHXDeviceCount: out is NULL");
  *out = device_count_or_throw();
  API_END();
}

int XGDMatrixCreateFromMat(const float* data, bst_ulong nrow, bst_ulong ncol, float missing, DMatrixHandle* out) {
  API_BEGIN();
  if (out == nullptr) throw OhxError("XGDMatrixCreateFromMat: out is NULL");
  if (data == nullptr && nrow * ncol != 0) throw OhxError("XGDMatrixCreateFromMat: data is NULL");
  DeviceInfo dev = use_device(-1);
  auto d = std::make_unique<DMatrixObj>();
  d->nrow = nrow;
  d->ncol = ncol;
  d->missing = missing;
  d->device = dev.ordinal;
  const size_t count = (size_t)nrow * (size_t)ncol;
  d->owned = g_row_pool.take(count, dev.ordinal, &d->owned_cap);
  d->d_data = d->owned;
  if (count) {
    // two waits per call: the copy itself (pageable host memory) and one read-back of the inf flag
    hipStream_t s = lib_streams(dev.ordinal).exec;
    HIP_CHECK(hipMemcpyAsync(d->owned, data, count * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));          // `data` is the caller's again (:383 deallocates it)
    // xgboost 1.6.0 (SparsePage::Push): "Input data contains `inf` or `nan`"
    std::lock_guard<std::mutex> g(g_check.mu);
    g_check.ensure(1, dev.ordinal);
    HIP_CHECK(hipMemsetAsync(g_check.d.p, 0, sizeof(uint32_t), s));
    HIP_CHECK(launch_scan_dense(d->owned, count, missing, g_check.d.p, s));
    HIP_CHECK(hipMemcpyAsync(g_check.h.p, g_check.d.p, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (g_check.h.p[0] & kFlagInfInput) throw OhxError("Input data contains `inf` or `nan`");
  }
  g_dmats.add(d.get());
  *out = d.release();
  API_END();
}

int OHXDMatrixGetGrid(DMatrixHandle handle, int* im, int* jm, bst_ulong* row0, int* inferred) {
  API_BEGIN();
  DMatrixObj* d = as_dmat(handle);
  // a matrix the library copied itself can be searched whenever it is asked about
  if (d->owned != nullptr && !d->grid_looked && d->grid_im == 0) {
    HIP_CHECK(hipSetDevice(d->device));
    infer_level_size(*d, lib_streams(d->device).exec);
  }
  if (im) *im = d->grid_im;
  if (jm) *jm = d->grid_jm;
  if (row0) *row0 = d->grid_row0;
  if (inferred) *inferred = d->grid_inferred ? 1 : 0;
  API_END();
}

int OHXDMatrixInferGrid(DMatrixHandle handle, void* stream, int* found) {
  API_BEGIN();
  DMatrixObj* d = as_dmat(handle);
  HIP_CHECK(hipSetDevice(d->device));
  d->grid_im = d->grid_jm = 0;
  d->grid_row0 = 0;
  d->grid_inferred = false;
  infer_level_size(*d, static_cast<hipStream_t>(stream));   // waits for `stream`: the rows must be there
  if (found) *found = d->grid_inferred ? 1 : 0;
  API_END();
}

int OHXDMatrixCreateFromDevice(const float* d_data, bst_ulong nrow, bst_ulong ncol, float missing, DMatrixHandle* out) {
  API_BEGIN();
  if (out == nullptr) throw OhxError("OHXDMatrixCreateFromDevice: out is NULL");
  if (d_data == nullptr && nrow * ncol != 0) throw OhxError("OHXDMatrixCreateFromDevice: d_data is NULL");
  DeviceInfo dev = use_device(-1);
  auto d = std::make_unique<DMatrixObj>();
  d->nrow = nrow;
  d->ncol = ncol;
  d->missing = missing;
  d->d_data = d_data;
  d->device = dev.ordinal;
  g_dmats.add(d.get());
  *out = d.release();
  API_END();
}

int OHXDMatrixSetGrid(DMatrixHandle handle, int im, int jm, bst_ulong row0) {
  API_BEGIN();
  DMatrixObj* d = as_dmat(handle);
  if (im < 0 || jm < 0 || (im == 0) != (jm == 0)) throw OhxError("OHXDMatrixSetGrid: im and jm must both be positive (or both 0)");
  d->grid_im = im;
  d->grid_jm = jm;
  d->grid_row0 = im ? row0 : 0;
  d->grid_inferred = false;
  d->grid_looked = true;          // the caller has spoken (0, 0 = "no grid": 64 consecutive rows per wave)
  API_END();
}

int XGDMatrixFree(DMatrixHandle handle) {
  API_BEGIN();
  if (handle == nullptr || !g_dmats.remove(handle)) throw OhxError("DMatrix handle is invalid or has been freed");
  delete static_cast<DMatrixObj*>(handle);
  API_END();
}

int OHXReleaseScratch(void);   // defined at the end of the file: it knows every pool

int XGDMatrixNumRow(DMatrixHandle handle, bst_ulong* out) {
  API_BEGIN();
  if (out == nullptr) throw OhxError("XGDMatrixNumRow: out is NULL");
  *out = as_dmat(handle)->nrow;
  API_END();
}

int XGDMatrixNumCol(DMatrixHandle handle, bst_ulong* out) {
  API_BEGIN();
  if (out == nullptr) throw OhxError("XGDMatrixNumCol: out is NULL");
  *out = as_dmat(handle)->ncol;
  API_END();
}

int XGDMatrixSaveBinary(DMatrixHandle handle, const char* fname, int silent) {
  API_BEGIN();
  (void)silent;
  DMatrixObj* d = as_dmat(handle);
  if (fname == nullptr) throw OhxError("XGDMatrixSaveBinary: fname is NULL");
  const size_t count = (size_t)d->nrow * (size_t)d->ncol;
  std::vector<float> host(count);
  if (count) {
    HIP_CHECK(hipSetDevice(d->device));
    hipStream_t st = lib_streams(d->device).exec;
    if (d->owned == nullptr) order_behind_caller(d->device, st);
    HIP_CHECK(hipMemcpyAsync(host.data(), d->d_data, count * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
  }
  std::ofstream o(fname, std::ios::binary);
  if (!o) throw OhxError(std::string("cannot open '") + fname + "' for writing");
  o.write("OHXDMAT1", 8);
  o.write((const char*)&d->nrow, 8);
  o.write((const char*)&d->ncol, 8);
  o.write((const char*)&d->missing, 4);
  o.write((const char*)host.data(), (std::streamsize)(count * sizeof(float)));
  if (!o) throw OhxError(std::string("failed writing '") + fname + "'");
  API_END();
}

int XGDMatrixCreateFromFile(const char* fname, int silent, DMatrixHandle* out) {
  API_BEGIN();
  (void)silent;
  if (fname == nullptr || out == nullptr) throw OhxError("XGDMatrixCreateFromFile: NULL argument");
  std::string uri(fname), path(fname);
  bool csv = false;
  size_t q = uri.find('?');
  if (q != std::string::npos) {
    path = uri.substr(0, q);
    csv = uri.find("format=csv", q) != std::string::npos;
  }
  if (ends_with(path, ".csv")) csv = true;
  std::ifstream in(path, std::ios::binary);
  if (!in) throw OhxError("cannot open '" + path + "'");
  std::vector<float> host;
  uint64_t nrow = 0, ncol = 0;
  float missing = NAN;
  if (csv) {
    std::string line;
    while (std::getline(in, line)) {
      if (line.empty()) continue;
      uint64_t c = 0;
      std::stringstream ss(line);
      std::string cell;
      while (std::getline(ss, cell, ',')) {
        host.push_back(cell.empty() ? NAN : strtof(cell.c_str(), nullptr));
        ++c;
      }
      if (ncol == 0) ncol = c;
      if (c != ncol) throw OhxError("CSV '" + path + "': ragged row " + std::to_string(nrow));
      ++nrow;
    }
  } else {
    char magic[8];
    in.read(magic, 8);
    if (!in || memcmp(magic, "OHXDMAT1", 8) != 0)
      throw OhxError("'" + path + "' is not a dense matrix saved by XGDMatrixSaveBinary of this library "
                     "(xgboost's own binary DMatrix cache and libsvm text are not supported)");
    in.read((char*)&nrow, 8);
    in.read((char*)&ncol, 8);
    in.read((char*)&missing, 4);
    host.resize((size_t)nrow * (size_t)ncol);
    in.read((char*)host.data(), (std::streamsize)(host.size() * sizeof(float)));
    if (!in) throw OhxError("'" + path + "' is truncated");
  }
  int rc = XGDMatrixCreateFromMat(host.data(), nrow, ncol, missing, out);
  if (rc != 0) throw OhxError(g_last_error);
  API_END();
}

int XGBoosterCreate(const DMatrixHandle dmats[], bst_ulong len, BoosterHandle* out) {
  API_BEGIN();
  if (out == nullptr) throw OhxError("XGBoosterCreate: out is NULL");
  // The reference passes one handle BY VALUE with len == 0 (OH_GridCompMod.F90:255-256):
  // with len == 0 the pointer is not an array and must not be read.
  for (bst_ulong i = 0; i < len; ++i) (void)as_dmat(dmats[i]);
  auto* b = new BoosterObj();
  g_boosters.add(b);
  *out = b;
  API_END();
}

int XGBoosterFree(BoosterHandle handle) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (!g_boosters.remove(handle)) throw OhxError("Booster handle is invalid or has been freed");
  if (b->uploaded) (void)hipSetDevice(b->dev.ordinal);
  delete b;
  API_END();
}

int XGBoosterLoadModel(BoosterHandle handle, const char* fname) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (fname == nullptr) throw OhxError("XGBoosterLoadModel: fname is NULL");
  adopt_model(*b, load_model_file(fname));
  API_END();
}

int XGBoosterLoadModelFromBuffer(BoosterHandle handle, const void* buf, bst_ulong len) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  adopt_model(*b, load_model_buffer(buf, (size_t)len));
  API_END();
}

int XGBoosterSaveModel(BoosterHandle handle, const char* fname) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (fname == nullptr) throw OhxError("XGBoosterSaveModel: fname is NULL");
  if (!b->loaded) throw OhxError("XGBoosterSaveModel: the booster holds no model");
  save_model_file(b->forest, fname);
  API_END();
}

int XGBoosterSetParam(BoosterHandle handle, const char* name, const char* value) {
  API_BEGIN();
  if (name == nullptr || value == nullptr) throw OhxError("XGBoosterSetParam: NULL argument");
  const std::string n(name), v(value);
  if (n == "ohx_copy_engine") {                               // process-wide; the handle may be NULL
    if (v != "kernel" && v != "dma" && v != "auto") throw OhxError("ohx_copy_engine must be kernel, dma or auto");
    g_copy_engine.store(v == "kernel" ? kCopyKernel : (v == "dma" ? kCopyDma : kCopyAuto));
    return 0;
  }
  if (n == "ohx_register_host" && handle == nullptr) {       // process-wide: settable before any booster exists
    g_host_registry.on.store(atoi(value) != 0);
    if (atoi(value) == 0) g_host_registry.release_all();
    return 0;
  }
  BoosterObj* b = as_booster(handle);
  if (n == "ohx_kernel") {
    if (v != "auto" && v != "wide" && v != "packed1" && v != "packed2" && v != "packed4" && v != "super1" &&
        v != "super2" && v != "super3" && v != "super4" && v != "ring")
      throw OhxError("ohx_kernel must be one of auto, wide, packed1, packed2, packed4, super1 .. super4, ring");
    if (v != b->kernel_name) invalidate_device_state(*b);
    b->kernel_name = v;
  } else if (n == "ohx_top_levels") {
    int k = atoi(value);
    if (k < 1 || k > 30) throw OhxError("ohx_top_levels must be in 1..30");
    if (k != b->layout.top_levels) invalidate_device_state(*b);
    b->layout.top_levels = k;
  } else if (n == "ohx_line_slots") {
    int k = atoi(value);
    if (k < 0 || k > 1024 || (k & 1)) throw OhxError("ohx_line_slots must be even, 0..1024");
    if (k != b->layout.line_slots) invalidate_device_state(*b);
    b->layout.line_slots = k;
  } else if (n == "ohx_min_chunk") {
    int k = atoi(value);
    if (k < 2 || k > 1024) throw OhxError("ohx_min_chunk must be in 2..1024");
    if (k != b->layout.min_chunk) invalidate_device_state(*b);
    b->layout.min_chunk = k;
  } else if (n == "ohx_launches_per_residency") {
    int k = atoi(value);
    if (k < 0 || k > 1000000) throw OhxError("ohx_launches_per_residency must be >= 0");
    b->tune.launches_per_residency = k;
  } else if (n == "ohx_brick") {
    // "bi,bj,bk": gridcells a wave takes along i, j, k (powers of two, product 64); "auto" = chosen per
    // call for the fewest idle lanes; "0,0,0" = 64 consecutive rows
    int e[3] = {-1, -1, -1}, lg[3] = {0, 0, 0};
    if (v == "auto") {
      b->tune.brick_li = b->tune.brick_lj = b->tune.brick_lk = -1;
    } else if (sscanf(value, "%d,%d,%d", &e[0], &e[1], &e[2]) != 3) {
      throw OhxError("ohx_brick must be \"bi,bj,bk\" or \"auto\"");
    } else if (e[0] == 0 && e[1] == 0 && e[2] == 0) {
      b->tune.brick_li = b->tune.brick_lj = b->tune.brick_lk = 0;
    } else {
      for (int q = 0; q < 3; ++q) {
        if (e[q] < 1 || e[q] > 64 || (e[q] & (e[q] - 1))) throw OhxError("ohx_brick extents must be powers of two");
        while ((1 << lg[q]) < e[q]) ++lg[q];
      }
      if (lg[0] + lg[1] + lg[2] != 6) throw OhxError("ohx_brick extents must multiply to 64");
      b->tune.brick_li = lg[0];
      b->tune.brick_lj = lg[1];
      b->tune.brick_lk = lg[2];
    }
  } else if (n == "ohx_brick_k_fastest") {
    b->tune.brick_k_fastest = atoi(value) != 0;
  } else if (n == "ohx_cluster") {
    // rows in no known order: "auto" groups them by tree-top decisions unless they look ordered; "on" / "off"
    if (v == "auto") b->tune.cluster = -1;
    else if (v == "on" || v == "1") b->tune.cluster = 1;
    else if (v == "off" || v == "0") b->tune.cluster = 0;
    else throw OhxError("ohx_cluster must be auto, on or off");
  } else if (n == "ohx_cluster_trees") {
    b->tune.cluster_trees = std::max(1, atoi(value));
  } else if (n == "ohx_cluster_zorder") {
    b->tune.cluster_zorder = atoi(value) != 0;
  } else if (n == "ohx_cluster_steps") {
    b->tune.cluster_steps = std::max(1, atoi(value));
  } else if (n == "ohx_lds_pad") {
    b->tune.lds_pad = atoi(value);
  } else if (n == "ohx_register_host") {
    // process-wide (the host form of OHXOHPostProcess has no booster): see HostRegistry
    g_host_registry.on.store(atoi(value) != 0);
    if (atoi(value) == 0) g_host_registry.release_all();
  } else if (n == "ohx_reserve_cus") {
    const int k = atoi(value);
    if (k < 0 || k > 128) throw OhxError("ohx_reserve_cus must be 0..128");
    b->tune.reserve_cus = k;
  } else if (n == "ohx_ring_rounds") {
    const int k = atoi(value);
    if (k < 0) throw OhxError("ohx_ring_rounds must be >= 0");
    b->tune.ring_rounds = k;
  } else if (n == "ohx_copy_blocks") {
    g_copy_blocks.store((uint32_t)std::max(0, atoi(value)));
  } else if (n == "ohx_run1_pieces") {
    const int k = atoi(value);
    if (k < 0 || k > 64) throw OhxError("ohx_run1_pieces must be 0 .. 64 (0 and 1: one piece)");
    b->tune.run1_pieces = k;
  } else if (n == "ohx_tree_split") {
    if (v == "auto") b->tune.tree_split = -1;
    else if (v == "off") b->tune.tree_split = 0;
    else {
      const int k = atoi(value);
      if (k < 2 || k > 10) throw OhxError("ohx_tree_split must be auto, off or 2..10");
      b->tune.tree_split = k;
    }
  } else if (n == "ohx_defer_missing") {
    if (v != "auto" && v != "on" && v != "off") throw OhxError("ohx_defer_missing must be auto, on or off");
    b->tune.defer_missing = v == "auto" ? -1 : (v == "on" ? 1 : 0);
  } else if (n == "ohx_prefetch") {
    b->tune.prefetch = atoi(value) != 0;
  } else if (n == "ohx_coop_rows") {
    b->tune.coop_rows = atoi(value) != 0;
  } else if (n == "ohx_xcd_remap") {
    b->tune.xcd_remap = atoi(value) != 0;
  } else if (n == "ohx_overlap_group") {
    const int k = atoi(value);
    if (k < 0) throw OhxError("ohx_overlap_group must be >= 0");
    b->tune.overlap_group = k;
    if (b->uploaded) {
      HIP_CHECK(hipSetDevice(b->dev.ordinal));
      ensure_train_stream(*b);
    }
  } else if (n == "ohx_tree_tops") {
    const std::string v = value;
    if (v != "auto" && v != "on" && v != "off") throw OhxError("ohx_tree_tops must be auto, on or off");
    b->tune.tree_tops = v == "auto" ? -1 : (v == "on" ? 1 : 0);
  } else if (n == "ohx_device") {
    int k = atoi(value);
    if (k != b->device_pref) invalidate_device_state(*b);
    b->device_pref = k;
  }
  // any other name is an xgboost training/runtime parameter with no meaning here
  API_END();
}

int XGBoosterPredict(BoosterHandle handle, DMatrixHandle dmat, int option_mask, unsigned ntree_limit, int training,
                     bst_ulong* out_len, const float** out_result) {
  API_BEGIN();
  (void)training;
  BoosterObj* b = as_booster(handle);
  DMatrixObj* d = as_dmat(dmat);
  if (out_len == nullptr || out_result == nullptr) throw OhxError("XGBoosterPredict: NULL output argument");
  if (!b->loaded) throw OhxError("the booster holds no model: call XGBoosterLoadModel first");
  uint32_t t0, t1;
  tree_range(*b, ntree_limit, &t0, &t1);
  const size_t per_row = (option_mask == 16) ? (size_t)(t1 - t0) : 1;
  const size_t count = (size_t)d->nrow * per_row;
  ensure_uploaded(*b);
  b->d_pred.ensure(count);
  b->h_pred.ensure(count);
  if (d->owned == nullptr) order_behind_caller(b->dev.ordinal, b->s_exec);
  launch_predict_checked(*b, *d, option_mask, ntree_limit, b->d_pred.p, b->s_exec);
  if (count) HIP_CHECK(hipMemcpyAsync(b->h_pred.p, b->d_pred.p, count * sizeof(float), hipMemcpyDeviceToHost, b->s_exec));
  raise_flag_errors(*b, b->s_exec);              // waits for the stream: the predictions are in h_pred
  *out_len = count;
  *out_result = b->h_pred.p;
  API_END();
}

int OHXBoosterPredictDevice(BoosterHandle handle, DMatrixHandle dmat, int option_mask, unsigned ntree_limit,
                            float* d_out, void* stream) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  DMatrixObj* d = as_dmat(dmat);
  if (d_out == nullptr && d->nrow != 0) throw OhxError("OHXBoosterPredictDevice: d_out is NULL");
  d->used_async = true;
  launch_predict_checked(*b, *d, option_mask, ntree_limit, d_out, static_cast<hipStream_t>(stream));
  API_END();
}

int OHXBoosterCheck(BoosterHandle handle, void* stream) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  ensure_uploaded(*b);
  raise_flag_errors(*b, static_cast<hipStream_t>(stream));
  API_END();
}

static void fields_common(BoosterObj& b, FieldsArgs& a, const int32_t is2d[], int nfield, int pl_feature, int im,
                          int jm, int km, int k1, int k2, float missing, int apply_pow10, float ohscale) {
  if (nfield < 0 || nfield > 32) throw OhxError("predict_fields: nfield must be 0..32");
  if ((uint32_t)nfield > b.forest.num_feature)
    throw OhxError("Number of columns does not match number of features in booster (" + std::to_string(nfield) +
                   " vs. " + std::to_string(b.forest.num_feature) + ")");
  if (b.forest.num_feature > 32) throw OhxError("predict_fields supports boosters with at most 32 features");
  if (im <= 0 || jm <= 0 || km <= 0) throw OhxError("predict_fields: im, jm, km must be positive");
  if (k1 < 1 || k2 > km || k2 < k1 - 1) throw OhxError("predict_fields: need 1 <= k1, k2 <= km, k2 >= k1 - 1");
  if (!objective_is_identity(b.forest.objective))
    throw OhxError("objective '" + b.forest.objective + "' is not supported by predict_fields");
  a.is2d_mask = 0;
  for (int f = 0; f < nfield; ++f)
    if (is2d[f]) a.is2d_mask |= (1u << f);
  a.pl_feature = pl_feature < 0 ? 0xFFFFFFFFu : (uint32_t)pl_feature;
  a.nfield = (uint32_t)nfield;
  a.im = im;
  a.jm = jm;
  a.km = km;
  a.k1 = k1 - 1;
  a.k2 = k2 - 1;
  a.missing = missing;
  tree_range(b, 0, &a.tree_begin, &a.tree_end);
  a.apply_pow10 = apply_pow10;
  a.scale = ohscale;
  a.flags = b.d_flags.p;
}

int OHXBoosterPredictFieldsDevice(BoosterHandle handle, const float* const d_fields[], const int32_t is2d[],
                                  int nfield, int pl_feature, int im, int jm, int km, int k1, int k2, float missing,
                                  int apply_pow10, float ohscale, float* d_oh_ml, float* d_margin, void* stream) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (d_fields == nullptr || is2d == nullptr || d_oh_ml == nullptr) throw OhxError("predict_fields: NULL argument");
  ensure_uploaded(*b);
  FieldsArgs a{};
  fields_common(*b, a, is2d, nfield, pl_feature, im, jm, km, k1, k2, missing, apply_pow10, ohscale);
  for (int f = 0; f < nfield; ++f) {
    if (d_fields[f] == nullptr) throw OhxError("predict_fields: field " + std::to_string(f) + " is NULL");
    a.field[f] = d_fields[f];
  }
  a.out = d_oh_ml;
  a.margin_out = d_margin;
  if (pick_kernel(*b) == KernelKind::Wide) ensure_wide(*b);
  if (a.k2 >= a.k1) {
    const uint64_t nrow = (uint64_t)im * (uint64_t)jm * (uint64_t)(a.k2 - a.k1 + 1);
    LaunchTuning tune = b->tune;
    leaf_room(*b, nrow, a.tree_end - a.tree_begin, tune, im, jm, 0);
    const bool deferring = nfield == 27 && defer_prepare(*b, nrow, tune, static_cast<hipStream_t>(stream));
    HIP_CHECK(launch_predict_fields(pick_kernel(*b), device_forest(*b), a, b->dev.num_cus,
                                    static_cast<hipStream_t>(stream), tune));
    if (deferring) defer_look(*b, nrow, static_cast<hipStream_t>(stream));
  }
  API_END();
}

int OHXBoosterPredictFields(BoosterHandle handle, const float* const fields[], const int32_t is2d[], int nfield,
                            int pl_feature, int im, int jm, int km, int k1, int k2, float missing, int apply_pow10,
                            float ohscale, float* oh_ml, float* margin) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (fields == nullptr || is2d == nullptr || oh_ml == nullptr) throw OhxError("predict_fields: NULL argument");
  ensure_uploaded(*b);
  FieldsArgs a{};
  fields_common(*b, a, is2d, nfield, pl_feature, im, jm, km, k1, k2, missing, apply_pow10, ohscale);
  if (a.k2 < a.k1) return 0;
  const size_t plane = (size_t)im * (size_t)jm;
  const size_t nlev = (size_t)(a.k2 - a.k1 + 1);
  const size_t nrow = plane * nlev;
  if (b->d_stage.size() < (size_t)nfield) b->d_stage.resize((size_t)nfield);
  // Only the slab k1..k2 of each 3-D field crosses PCIe, and it crosses in pieces of whole
  // levels: while piece c is being walked on the compute stream, piece c+1 is already on its
  // way over PCIe on the copy stream (a step is PCIe-bound: 756 MB in, 28 MB out for a
  // C360/8 sub-domain, against ~6 ms of kernels).
  // pieces of at least two residencies of rows (the launch granule), at most 8 pieces
  const size_t min_rows = (size_t)64 * 256 * 20 * 2;
  size_t lev_per_piece = (min_rows + plane - 1) / plane;
  if (lev_per_piece * 8 < nlev) lev_per_piece = (nlev + 7) / 8;
  if (lev_per_piece < 1) lev_per_piece = 1;
  // A rank-sized block is ONE piece, and with the caller's arrays registered its 27 fields cross in one copy launch
  // (HostMover) instead of 27 copies at 27 fixed prices: a 48 x 24 x 72 block 0.55 -> 0.33 ms, 96 x 48 x 72 1.26 -> 1.03
  // (profiles/r04_sweeps.txt; the reference's five calls: 0.35 and 1.02)
  const bool registered = g_host_registry.on.load(std::memory_order_relaxed);
  const bool one_launch = registered && lev_per_piece >= nlev && nfield <= (int)kCopyListMax &&
                          ((size_t)nfield * nrow + nrow * 2) * sizeof(float) <= HostMover::kKernelBytesMax;
  HostMover in(b->s_exec, true);
  for (int f = 0; f < nfield; ++f) {
    if (fields[f] == nullptr) throw OhxError("predict_fields: field " + std::to_string(f) + " is NULL");
    b->d_stage[(size_t)f].ensure(is2d[f] ? plane : nrow);
    a.field[f] = b->d_stage[(size_t)f].p;
    if (one_launch) {
      if (is2d[f]) in.add(b->d_stage[(size_t)f].p, fields[f], plane);
      else in.add_slice(b->d_stage[(size_t)f].p, fields[f], plane * (size_t)km, plane * (size_t)a.k1, nrow);
    } else if (is2d[f]) {
      upload_host_array(b->d_stage[(size_t)f].p, fields[f], plane, b->s_copy);
    } else if (registered) {
      (void)g_host_registry.want(fields[f], plane * (size_t)km * sizeof(float));       // the whole field once, not per piece
    }
  }
  if (one_launch) in.go();
  a.src_k0 = a.k1;
  a.out_k0 = a.k1;
  b->d_stage_out.ensure(nrow);
  a.out = b->d_stage_out.p;
  if (margin) {
    b->d_stage_margin.ensure(nrow);
    a.margin_out = b->d_stage_margin.p;
  }
  if (pick_kernel(*b) == KernelKind::Wide) ensure_wide(*b);
  const float* margin_base = a.margin_out;
  std::vector<hipEvent_t> events;
  for (size_t l0 = 0; l0 < nlev; l0 += lev_per_piece) {
    const size_t l1 = std::min(nlev, l0 + lev_per_piece);
    for (int f = 0; f < nfield; ++f) {
      if (is2d[f] || one_launch) continue;
      const float* src = fields[f] + plane * ((size_t)a.k1 + l0);
      HIP_CHECK(hipMemcpyAsync(b->d_stage[(size_t)f].p + plane * l0, src, plane * (l1 - l0) * sizeof(float),
                               hipMemcpyHostToDevice, b->s_copy));
    }
    if (!one_launch) {           // (one launch: the list went to the kernels' own stream, in front of them)
      hipEvent_t ev;
      HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      events.push_back(ev);
      HIP_CHECK(hipEventRecord(ev, b->s_copy));
      HIP_CHECK(hipStreamWaitEvent(b->s_exec, ev, 0));
    }
    FieldsArgs piece = a;
    piece.k1 = a.k1 + (int)l0;
    piece.k2 = a.k1 + (int)l1 - 1;
    // rows of this piece start at plane * l0 of the slab-ordered margin buffer
    piece.margin_out = margin_base ? const_cast<float*>(margin_base) + plane * l0 : nullptr;
    LaunchTuning tune = b->tune;
    const uint64_t piece_rows = plane * (l1 - l0);
    leaf_room(*b, piece_rows, piece.tree_end - piece.tree_begin, tune, im, jm, 0);
    const bool deferring = nfield == 27 && defer_prepare(*b, piece_rows, tune, b->s_exec);
    HIP_CHECK(launch_predict_fields(pick_kernel(*b), device_forest(*b), piece, b->dev.num_cus, b->s_exec, tune));
    if (deferring) defer_look(*b, piece_rows, b->s_exec);
  }
  if (one_launch) {
    HostMover back(b->s_exec, false);
    back.add_slice(b->d_stage_out.p, oh_ml, plane * (size_t)km, plane * (size_t)a.k1, nrow);
    if (margin) back.add(margin, b->d_stage_margin.p, nrow);
    back.go();
  } else {
    if (registered) {
      (void)g_host_registry.want(oh_ml, plane * (size_t)km * sizeof(float));
      if (margin) (void)g_host_registry.want(margin, nrow * sizeof(float));
    }
    HIP_CHECK(hipMemcpyAsync(oh_ml + plane * (size_t)a.k1, b->d_stage_out.p, nrow * sizeof(float), hipMemcpyDeviceToHost,
                             b->s_exec));
    if (margin)
      HIP_CHECK(hipMemcpyAsync(margin, b->d_stage_margin.p, nrow * sizeof(float), hipMemcpyDeviceToHost, b->s_exec));
  }
  HIP_CHECK(hipStreamSynchronize(b->s_exec));
  for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
  raise_flag_errors(*b, b->s_exec);
  API_END();
}

// The host form's side of a tick (OHXBoosterRun1): its arrays cross PCIe in three lists on a copy stream, and
// run1_device says when - what the slab count needs first, what the feature engineering needs behind it, and the
// rest once the slab is known (of the sixteen fields only the walk reads, the slab's levels alone).
// Where the HOST spends a host-form tick (OHX_RUN1_TRACE=1 in the environment; measurement aid): microseconds since the
// call began at a few marks, printed for ticks 50 .. 52 of the process.
namespace {
struct TickTrace {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  std::vector<std::pair<const char*, double>> marks;
  void mark(const char* what) {
    marks.emplace_back(what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
  void print(unsigned tick) const {
    std::string line = "[libohxgb] run1 tick " + std::to_string(tick) + " (host, us):";
    char buf[96];
    for (const auto& m : marks) {
      snprintf(buf, sizeof buf, " %s %.1f", m.first, m.second);
      line += buf;
    }
    fprintf(stderr, "%s\n", line.c_str());
  }
};
static thread_local TickTrace* g_tick_trace = nullptr;
#define TICK_MARK(what) do { if (g_tick_trace) g_tick_trace->mark(what); } while (0)

struct Run1Feed {
  virtual ~Run1Feed() = default;
  // each enqueues copies on the feed's own stream and returns an event recorded behind them
  virtual hipEvent_t slab_inputs() = 0;                              // PLE and TROPP of the model
  virtual hipEvent_t prep_inputs() = 0;                              // what the feature engineering reads
  virtual hipEvent_t walk_inputs(int k1, int k2) = 0;                // 1-based slab levels
  virtual hipEvent_t post_inputs() = 0;                              // what the tick's last kernel reads; asked for once the walk is enqueued
  // true when the slab count may read PLE and TROPP where the caller has them (registered arrays: *ple, *tropp = the
  // addresses the GPU reaches them at): it then waits for nothing, and their device copies - the tick's last kernel
  // reads them too - cross with prep_inputs(); slab_inputs() is not called
  virtual bool slab_in_place(const float** ple, const float** tropp) { (void)ple; (void)tropp; return false; }
  // true: a kernel that needs a list is launched when the HOST has seen the list's event, not enqueued behind a wait
  // on it (run1_device, below)
  virtual bool host_gated() const { return false; }
};
}  // namespace

static void run1_device(BoosterObj& b, const OHXRun1Args& r, hipStream_t stream, Run1Feed* feed = nullptr) {
  if (r.im <= 0 || r.jm <= 0 || r.km <= 0) throw OhxError("OHXBoosterRun1: im, jm, km must be positive");
  if (b.forest.num_feature != 27) throw OhxError("OHXBoosterRun1 needs the 27-feature OH booster");
  const void* need[] = {r.ple_mod, r.t_mod, r.q_mod, r.tropp_mod, r.ple_bst, r.zle_bst, r.tauclw, r.taucli,
                        r.gmito3, r.gmitto3, r.lat_deg, r.t_bst, r.no2, r.o3, r.ch4, r.co, r.isop, r.acet,
                        r.c2h6, r.c3h8, r.prpe, r.alk4, r.mp, r.h2o2, r.cloud, r.qv, r.albuv, r.ch2o, r.sza,
                        r.default_oh, r.oh};
  for (const void* p : need)
    if (p == nullptr) throw OhxError("OHXBoosterRun1: a required field pointer is NULL");
  for (int i = 0; i < 7; ++i)
    if (r.scacoef[i] == nullptr) throw OhxError("OHXBoosterRun1: a scattering-coefficient pointer is NULL");
  const size_t plane = (size_t)r.im * (size_t)r.jm, vol = plane * (size_t)r.km;
  for (int i = 0; i < 8; ++i) b.d_run1[i].ensure(i == 7 ? plane : vol);
  b.d_run1[8].ensure(vol);
  b.d_run1[9].ensure(vol);
  b.d_slab.ensure(2);
  // a DIAG buffer the caller gave IS the place the feature is built in; scratch otherwise
  auto or_scratch = [&](float* wanted, int slot) { return wanted ? wanted : b.d_run1[slot].p; };
  float* pl_bst = or_scratch(r.diag_pl_bst, 0);
  float *tauclwdn = or_scratch(r.diag_tauclwdn, 1), *tauclidn = or_scratch(r.diag_tauclidn, 2);
  float *taucliup = or_scratch(r.diag_taucliup, 3), *tauclwup = or_scratch(r.diag_tauclwup, 4);
  float *aodup = or_scratch(r.diag_aodup, 5), *aoddn = or_scratch(r.diag_aoddn, 6), *strato3 = or_scratch(r.diag_strato3, 7);
  float* oh_ml = or_scratch(r.oh_boost, 8);
  float* aod = or_scratch(r.diag_aod, 9);

  // ---- How a tick is laid out on the GPU (r5).  Two streams: the caller's, and a side stream of the booster's.
  //   side:   slab count (needs PLE and TROPP only) -> OH_ML := 0 -> [features of piece 1] -> after the walk of piece 0:
  //           post-processing of piece 0, [features of piece 2] -> ...
  //   main:   features of piece 0 -> (the host reads the slab count) -> walk of piece 0 -> walk of piece 1 -> ...
  // A PIECE is a range of j of the grid, all levels: the column sums need whole columns, the walk needs every feature
  // of the cells it walks, the mask and the unit conversion need the walk's OH_ML - and none of them needs another
  // column.  So of the streaming kernels (pointwise features, column sums, slab count, post-processing: 1.3 ms of a
  // C360 tick's 20.2 when they ran one after the other around the walk) only the first piece's features and the last
  // piece's post-processing are not hidden behind a walk, which leaves the CUs' spare registers and the whole of HBM's
  // bandwidth to them.  That was the plan; measured, pieces lose (below), so a tick is ONE piece unless ohx_run1_pieces > 1.
  if (!b.run1_fork) {
    HIP_CHECK(hipEventCreateWithFlags(&b.run1_fork, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&b.run1_slab, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&b.run1_join, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&b.run1_clear, hipEventDisableTiming));
  }
  b.h_slab.ensure(2);
  // The side stream.  Device form: the library's copy stream, idle there (the slab count runs beside the first
  // features: 0.19 ms of a C360 tick).  Host form: the copy stream carries the copies, and the side work goes to the
  // caller's stream in order - a third stream made a rank's tick slower alone (0.36-0.41 against 0.32 ms) and gave six
  // ranks on one GPU more queues than its scheduler maps at once (LibStreams; profiles/r05_sweeps.txt)
  hipStream_t side = (feed != nullptr || stream == b.s_copy) ? stream : b.s_copy;
  if (side != stream) {
    HIP_CHECK(hipEventRecord(b.run1_fork, stream));          // the side stream starts behind what the caller has enqueued
    HIP_CHECK(hipStreamWaitEvent(side, b.run1_fork, 0));
  }
  // host form: the slab count is the first thing of the tick.  Registered arrays: it reads PLE and TROPP over PCIe where
  // they are (0.34 MB of a rank's block; every edge once) and waits for no copy.  Else they cross first and it follows
  // them.  Either way its answer is back on the host while the features' inputs are still crossing, so the walk's own
  // inputs follow those over PCIe without a gap.
  SlabArgs sa;
  sa.im = r.im; sa.jm = r.jm; sa.km = r.km;
  sa.dynamic_k_range = r.dynamic_k_range; sa.tropp_min = r.tropp_min;
  sa.ple_mod = r.ple_mod; sa.tropp = r.tropp_mod; sa.result = b.d_slab.p;
  hipEvent_t prep_inputs = nullptr;
  const bool gated = feed != nullptr && feed->host_gated() && b.tune.run1_pieces <= 1;
  if (feed != nullptr) {
    if (!feed->slab_in_place(&sa.ple_mod, &sa.tropp)) {
      hipEvent_t ev = feed->slab_inputs();
      if (gated) HIP_CHECK(hipEventSynchronize(ev)); else HIP_CHECK(hipStreamWaitEvent(side, ev, 0));
    }
  }
  // (the result words are zero: set so when they were made, and again behind every read-back - a memset in front of the
  // count was a 12 us launch on the tick's critical path)
  // That trailing memset went to the side stream of ITS tick, and the device form returns with it still pending.  A tick
  // whose side stream is another one (device form, then host form; OHX_RUN1_STREAMS=1) must not count into words the old
  // memset has yet to clear - slab = 0, k1 = km + 1, nothing predicted, silently: it waits for run1_clear,
  // recorded behind that memset.
  if (!b.slab_zeroed) {
    HIP_CHECK(hipMemsetAsync(b.d_slab.p, 0, 2 * sizeof(int32_t), side));
    b.slab_zeroed = true;
  } else if (b.slab_zeroed_on != side) {
    HIP_CHECK(hipStreamWaitEvent(side, b.run1_clear, 0));
  }
  b.slab_zeroed_on = side;
  HIP_CHECK(launch_k_slab(sa, side));
  // ... and the next list is handed to the copy stream before this thread turns to anything else: the calls that
  // enqueue the rest of the slab count took it 50 us, during which the link stood idle (profiles/r05_sweeps.txt)
  if (feed != nullptr) prep_inputs = feed->prep_inputs();
  HIP_CHECK(hipMemcpyAsync(b.h_slab.p, b.d_slab.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, side));
  HIP_CHECK(hipEventRecord(b.run1_slab, side));
  HIP_CHECK(hipMemsetAsync(b.d_slab.p, 0, 2 * sizeof(int32_t), side));
  HIP_CHECK(hipEventRecord(b.run1_clear, side));
  HIP_CHECK(hipMemsetAsync(oh_ml, 0, vol * sizeof(float), side));            // self%OH_ML(:,:,:) = 0.0 (:1559)
  // A kernel that needs a list of the feed waits for it in one of two ways.  Enqueued behind a wait on the list's event
  // (hipStreamWaitEvent: a barrier packet on this queue that the command processor re-examines at its leisure - the
  // kernel behind it started 21-38 us after the copy had ended, profiles/r05_run1_timeline_block_48x24.txt), or GATED:
  // this thread waits for the event and launches then (a few us).  The feed says which.  Measured on a rank's tick: no
  // difference - the barriers' delay lies under the link's time (OHXBoosterRun1 below) - so nothing is gated by default.
  if (prep_inputs != nullptr && !gated) HIP_CHECK(hipStreamWaitEvent(stream, prep_inputs, 0));

  // pieces: ONE by default (ohx_run1_pieces 0 or 1).  n > 1 is an experiment knob: ranges of whole rows of bricks (four
  // j) walked one after the other, the pieces' extents independent of the slab count, which is not known yet.  Measured
  // (profiles/r05_sweeps.txt): more than one piece is slower - predict_fields_ring_kernel takes all of a CU's vector
  // registers (4 x 128 of 512 per SIMD), so a streaming kernel enqueued beside it waits for its end, and the pieces' own
  // launch tails cost 0.3 ms per C360 tick.
  const KernelKind kind = pick_kernel(b);
  int npieces = 1;
  if (kind == KernelKind::Ring && b.tune.run1_pieces > 1) npieces = std::max(1, std::min(b.tune.run1_pieces, r.jm / 8));
  std::vector<int> j_lo((size_t)npieces + 1, 0);
  for (int q = 1; q <= npieces; ++q) j_lo[(size_t)q] = q == npieces ? r.jm : (int)((int64_t)r.jm * q / npieces / 4 * 4);
  while (b.run1_prep.size() < (size_t)npieces) {
    hipEvent_t e1, e2;
    HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    b.run1_prep.push_back(e1);
    b.run1_walk.push_back(e2);
  }

  PrepArgs pa;
  pa.im = r.im; pa.jm = r.jm; pa.km = r.km;
  pa.ple_bst = r.ple_bst; pa.zle_bst = r.zle_bst; pa.tauclw = r.tauclw; pa.taucli = r.taucli;
  for (int i = 0; i < 7; ++i) pa.sca[i] = r.scacoef[i];
  pa.gmito3 = r.gmito3; pa.gmitto3 = r.gmitto3;
  pa.pl_bst = pl_bst; pa.tauclwdn = tauclwdn; pa.tauclidn = tauclidn; pa.taucliup = taucliup; pa.tauclwup = tauclwup;
  pa.aodup = aodup; pa.aoddn = aoddn; pa.strato3 = strato3;
  auto prep_piece = [&](int q, hipStream_t s) {
    PrepArgs p = pa;
    p.beside_a_walk = q >= 2 ? 1 : 0;        // pieces 0 and 1 are prepared before the first walk starts
    if (npieces > 1) {
      p.col0 = (uint64_t)j_lo[(size_t)q] * (uint64_t)r.im;
      p.ncols = (uint64_t)(j_lo[(size_t)q + 1] - j_lo[(size_t)q]) * (uint64_t)r.im;
    }
    HIP_CHECK(launch_feature_prep(p, aod, s));
  };
  PostArgs po;
  po.im = r.im; po.jm = r.jm; po.km = r.km;
  po.avogad = r.avogad; po.runiv = r.runiv; po.epsilon = r.epsilon;
  po.ple_mod = r.ple_mod; po.t_mod = r.t_mod; po.q_mod = r.q_mod; po.tropp = r.tropp_mod;
  po.default_oh = r.default_oh; po.oh_ml = oh_ml; po.oh = r.oh; po.ndwet = r.ndwet;
  auto post_piece = [&](int q, hipStream_t s) {
    PostArgs p = po;
    if (npieces > 1) {
      p.col0 = (uint64_t)j_lo[(size_t)q] * (uint64_t)r.im;
      p.ncols = (uint64_t)(j_lo[(size_t)q + 1] - j_lo[(size_t)q]) * (uint64_t)r.im;
    }
    HIP_CHECK(launch_post_process(p, s));
  };

  if (!gated) prep_piece(0, stream);
  if (npieces > 1) {
    if (prep_inputs != nullptr && side != stream) HIP_CHECK(hipStreamWaitEvent(side, prep_inputs, 0));      // later pieces' features run here
    prep_piece(1, side);
    HIP_CHECK(hipEventRecord(b.run1_prep[1], side));
  }
  // the one wait in the middle of a tick: the slab count sizes the walk's launches (and the reference asserts on it,
  // :287-288); it was enqueued before everything else of this tick and the features are being computed meanwhile
  TICK_MARK("features-enqueued");
  HIP_CHECK(hipEventSynchronize(b.run1_slab));
  TICK_MARK("slab-known");
  const int32_t slab[2] = {b.h_slab.p[0], b.h_slab.p[1]};
  if (!r.dynamic_k_range && slab[1] != 0) {
    (void)hipStreamSynchronize(side);
    (void)hipStreamSynchronize(stream);
    throw OhxError("OH Prediction: Minimum tropopause pressure is not low enough!");
  }
  const int k1 = r.km - slab[0] + 1, k2 = r.km;   // 1-based (:300-301)
  if (r.k1) *r.k1 = k1;
  if (r.k2) *r.k2 = k2;
  hipEvent_t post_inputs = nullptr, walk_inputs = nullptr;
  if (feed != nullptr) walk_inputs = feed->walk_inputs(k1, k2);
  TICK_MARK("walk-inputs-enqueued");
  if (gated) {
    // (the walk's list is on its way before this thread waits for the features': the link is not idle meanwhile)
    HIP_CHECK(hipEventSynchronize(prep_inputs));
    prep_piece(0, stream);
    TICK_MARK("features-launched");
  } else if (walk_inputs != nullptr) {
    HIP_CHECK(hipStreamWaitEvent(stream, walk_inputs, 0));
  }

  const float* fields[27] = {r.lat_deg, pl_bst, r.t_bst, r.no2, r.o3, r.ch4, r.co, r.isop, r.acet, r.c2h6, r.c3h8,
                             r.prpe, r.alk4, r.mp, r.h2o2, tauclwdn, tauclidn, taucliup, tauclwup, r.cloud, r.qv,
                             strato3, r.albuv, aodup, aoddn, r.ch2o, r.sza};      // order of :313-339
  static const int32_t is2d[27] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 1};
  if (kind == KernelKind::Wide) ensure_wide(b);
  // OH_ML := 0 (side stream) lies before the first store of a walk
  if (side != stream) {
    HIP_CHECK(hipEventRecord(b.run1_join, side));
    HIP_CHECK(hipStreamWaitEvent(stream, b.run1_join, 0));
  }
  for (int q = 0; q < npieces; ++q) {
    const int jq = j_lo[(size_t)q + 1] - j_lo[(size_t)q];
    const uint64_t first = (uint64_t)j_lo[(size_t)q] * (uint64_t)r.im;        // the piece's first column
    if (q >= 1) HIP_CHECK(hipStreamWaitEvent(stream, b.run1_prep[(size_t)q], 0));      // its features are there
    if (k2 >= k1 && jq > 0) {
      FieldsArgs fa{};
      fa.is2d_mask = 0;
      for (int f = 0; f < 27; ++f) {
        fa.field[f] = fields[f] + first;
        if (is2d[f]) fa.is2d_mask |= (1u << f);
      }
      fa.pl_feature = 1;
      fa.nfield = 27;
      fa.im = r.im; fa.jm = jq; fa.km = r.km;
      fa.level_stride = npieces > 1 ? plane : 0;
      fa.k1 = k1 - 1; fa.k2 = k2 - 1;
      fa.missing = r.missing;
      tree_range(b, 0, &fa.tree_begin, &fa.tree_end);
      fa.apply_pow10 = 1;
      fa.scale = r.ohscale;
      fa.out = oh_ml + first;
      fa.flags = b.d_flags.p;
      const uint64_t slab_rows = (uint64_t)r.im * (uint64_t)jq * (uint64_t)(k2 - k1 + 1);
      LaunchTuning tune = b.tune;
      leaf_room(b, slab_rows, fa.tree_end - fa.tree_begin, tune, r.im, jq, 0);
      const bool deferring = defer_prepare(b, slab_rows, tune, stream);
      if (gated && q == 0) {
        HIP_CHECK(hipEventSynchronize(walk_inputs));
        TICK_MARK("walk-inputs-there");
      }
      HIP_CHECK(launch_predict_fields(kind, device_forest(b), fa, b.dev.num_cus, stream, tune));
      if (deferring && q + 1 == npieces) defer_look(b, slab_rows, stream);
    }
    TICK_MARK("walk-enqueued");
    if (feed != nullptr && post_inputs == nullptr) post_inputs = feed->post_inputs();      // behind the (first) walk's launch
    if (npieces == 1) {
      if (post_inputs != nullptr) HIP_CHECK(hipStreamWaitEvent(stream, post_inputs, 0));
      post_piece(0, stream);
      break;
    }
    // the piece is walked: its mask and unit conversion, then the features of the piece after next, beside the next walk
    HIP_CHECK(hipEventRecord(b.run1_walk[(size_t)q], stream));
    HIP_CHECK(hipStreamWaitEvent(side, b.run1_walk[(size_t)q], 0));
    if (post_inputs != nullptr) HIP_CHECK(hipStreamWaitEvent(side, post_inputs, 0));
    post_piece(q, side);
    if (q + 2 < npieces) {
      prep_piece(q + 2, side);
      HIP_CHECK(hipEventRecord(b.run1_prep[(size_t)q + 2], side));
    }
  }
  if (npieces > 1 && side != stream) {       // the caller's stream ends behind everything the side stream did
    HIP_CHECK(hipEventRecord(b.run1_join, side));
    HIP_CHECK(hipStreamWaitEvent(stream, b.run1_join, 0));
  }
}

int OHXBoosterRun1Device(BoosterHandle handle, const OHXRun1Args* args, void* stream) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (args == nullptr) throw OhxError("OHXBoosterRun1Device: args is NULL");
  if (!objective_is_identity(b->forest.objective))
    throw OhxError("objective '" + b->forest.objective + "' is not supported by OHXBoosterRun1");
  ensure_uploaded(*b);
  run1_device(*b, *args, static_cast<hipStream_t>(stream));
  API_END();
}

static PostArgs post_args(int im, int jm, int km, float avogad, float runiv, float epsilon, const float* ple_mod,
                          const float* t_mod, const float* q_mod, const float* tropp_mod, const float* default_oh,
                          const float* oh_ml, float* oh, float* ndwet) {
  if (im <= 0 || jm <= 0 || km <= 0) throw OhxError("OHXOHPostProcess: im, jm, km must be positive");
  if (!ple_mod || !t_mod || !q_mod || !tropp_mod || !default_oh || !oh_ml || !oh)
    throw OhxError("OHXOHPostProcess: a required field pointer is NULL");
  PostArgs po;
  po.im = im; po.jm = jm; po.km = km;
  po.avogad = avogad; po.runiv = runiv; po.epsilon = epsilon;
  po.ple_mod = ple_mod; po.t_mod = t_mod; po.q_mod = q_mod; po.tropp = tropp_mod;
  po.default_oh = default_oh; po.oh_ml = oh_ml; po.oh = oh; po.ndwet = ndwet;
  return po;
}

int OHXOHPostProcessDevice(int im, int jm, int km, float avogad, float runiv, float epsilon, const float* d_ple_mod,
                           const float* d_t_mod, const float* d_q_mod, const float* d_tropp_mod,
                           const float* d_default_oh, const float* d_oh_ml, float* d_oh, float* d_ndwet, void* stream) {
  API_BEGIN();
  (void)use_device(-1);
  HIP_CHECK(launch_post_process(post_args(im, jm, km, avogad, runiv, epsilon, d_ple_mod, d_t_mod, d_q_mod, d_tropp_mod,
                                          d_default_oh, d_oh_ml, d_oh, d_ndwet), static_cast<hipStream_t>(stream)));
  API_END();
}

namespace {
// staging of the host form: one set of device buffers per process, reused from tick to tick
struct PostScratch {
  std::mutex mu;
  int device = -1;
  DevBuf<float> buf[8];
};
static PostScratch g_post;   // static: inside extern "C" an unnamed namespace alone does not keep the name out of the dynamic table
}  // namespace

int OHXOHPostProcess(int im, int jm, int km, float avogad, float runiv, float epsilon, const float* ple_mod,
                     const float* t_mod, const float* q_mod, const float* tropp_mod, const float* default_oh,
                     const float* oh_ml, float* oh, float* ndwet) {
  API_BEGIN();
  (void)post_args(im, jm, km, avogad, runiv, epsilon, ple_mod, t_mod, q_mod, tropp_mod, default_oh, oh_ml, oh, ndwet);
  const DeviceInfo dev = use_device(-1);
  const size_t plane = (size_t)im * (size_t)jm, vol = plane * (size_t)km, edge = plane * (size_t)(km + 1);
  std::lock_guard<std::mutex> g(g_post.mu);
  if (g_post.device != dev.ordinal) {
    for (auto& bf : g_post.buf) bf.release();
    g_post.device = dev.ordinal;
  }
  const float* src[6] = {ple_mod, t_mod, q_mod, tropp_mod, default_oh, oh_ml};
  const size_t n[6] = {edge, vol, vol, plane, vol, vol};
  hipStream_t st = lib_streams(dev.ordinal).exec;
  HostMover in(st, true);
  for (int i = 0; i < 6; ++i) {
    g_post.buf[i].ensure(n[i]);
    in.add(g_post.buf[i].p, src[i], n[i]);
  }
  in.go();
  g_post.buf[6].ensure(vol);
  if (ndwet) g_post.buf[7].ensure(vol);
  HIP_CHECK(launch_post_process(post_args(im, jm, km, avogad, runiv, epsilon, g_post.buf[0].p, g_post.buf[1].p,
                                          g_post.buf[2].p, g_post.buf[3].p, g_post.buf[4].p, g_post.buf[5].p,
                                          g_post.buf[6].p, ndwet ? g_post.buf[7].p : nullptr), st));
  HostMover back(st, false);
  back.add(oh, g_post.buf[6].p, vol);
  if (ndwet) back.add(ndwet, g_post.buf[7].p, vol);
  back.go();
  HIP_CHECK(hipStreamSynchronize(st));
  API_END();
}

int OHXJulianDay(int nymd, int* jday) {
  API_BEGIN();
  if (jday == nullptr) throw OhxError("OHXJulianDay: jday is NULL");
  static const int days[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
  const int ny = nymd / 10000, mm = (nymd % 10000) / 100, dd = nymd % 100;
  // leap_year (:1940-1970): no leap years before year 0
  const bool leap = ny >= 0 && ((ny % 100 == 0 && ny % 400 == 0) || (ny % 4 == 0 && ny % 100 != 0));
  int ds = dd;
  for (int m = 1; m < mm && m <= 12; ++m) ds += (m == 2 && leap) ? 29 : days[m - 1];
  *jday = ds;
  API_END();
}

int OHXSolarGeometryDevice(int jday, const float* d_lats, const float* d_lons, int im, int jm, float deg2rad,
                           float rad2deg, float* d_lat_deg, float* d_sza_noon, void* stream) {
  API_BEGIN();
  if (im < 0 || jm < 0) throw OhxError("OHXSolarGeometry: im and jm must not be negative");
  if ((size_t)im * (size_t)jm != 0 && (d_lats == nullptr || (d_sza_noon != nullptr && d_lons == nullptr)))
    throw OhxError("OHXSolarGeometry: LATS (and LONS, for the zenith angle) must not be NULL");
  use_device(-1);
  SolarArgs a;
  a.im = im; a.jm = jm; a.jday = jday;
  a.deg2rad = deg2rad; a.rad2deg = rad2deg;
  a.lats = d_lats; a.lons = d_lons;
  a.lat_deg = d_lat_deg; a.sza_noon = d_sza_noon;
  HIP_CHECK(launch_solar_geometry(a, static_cast<hipStream_t>(stream)));
  API_END();
}

int OHXSolarGeometry(int jday, const float* lats, const float* lons, int im, int jm, float deg2rad, float rad2deg,
                     float* lat_deg, float* sza_noon) {
  API_BEGIN();
  if (im < 0 || jm < 0) throw OhxError("OHXSolarGeometry: im and jm must not be negative");
  const size_t plane = (size_t)im * (size_t)jm;
  if (plane == 0) return 0;
  if (lats == nullptr || (sza_noon != nullptr && lons == nullptr))
    throw OhxError("OHXSolarGeometry: LATS (and LONS, for the zenith angle) must not be NULL");
  const DeviceInfo dev = use_device(-1);
  hipStream_t st = lib_streams(dev.ordinal).exec;
  DevBuf<float> d_lats, d_lons, d_lat, d_sza;
  d_lats.ensure(plane);
  HIP_CHECK(hipMemcpyAsync(d_lats.p, lats, plane * sizeof(float), hipMemcpyHostToDevice, st));
  if (sza_noon) {
    d_lons.ensure(plane);
    HIP_CHECK(hipMemcpyAsync(d_lons.p, lons, plane * sizeof(float), hipMemcpyHostToDevice, st));
    d_sza.ensure(plane);
  }
  if (lat_deg) d_lat.ensure(plane);
  SolarArgs a;
  a.im = im; a.jm = jm; a.jday = jday;
  a.deg2rad = deg2rad; a.rad2deg = rad2deg;
  a.lats = d_lats.p; a.lons = d_lons.p;
  a.lat_deg = lat_deg ? d_lat.p : nullptr;
  a.sza_noon = sza_noon ? d_sza.p : nullptr;
  HIP_CHECK(launch_solar_geometry(a, st));
  if (lat_deg) HIP_CHECK(hipMemcpyAsync(lat_deg, d_lat.p, plane * sizeof(float), hipMemcpyDeviceToHost, st));
  if (sza_noon) HIP_CHECK(hipMemcpyAsync(sza_noon, d_sza.p, plane * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));        // before the device buffers go
  API_END();
}

int OHXBoosterRun1(BoosterHandle handle, const OHXRun1Args* args) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (args == nullptr) throw OhxError("OHXBoosterRun1: args is NULL");
  if (!objective_is_identity(b->forest.objective))
    throw OhxError("objective '" + b->forest.objective + "' is not supported by OHXBoosterRun1");
  ensure_uploaded(*b);
  const OHXRun1Args& h = *args;
  if (h.im <= 0 || h.jm <= 0 || h.km <= 0) throw OhxError("OHXBoosterRun1: im, jm, km must be positive");
  const size_t plane = (size_t)h.im * (size_t)h.jm, vol = plane * (size_t)h.km, edge = plane * (size_t)(h.km + 1);
  OHXRun1Args d = h;
  // Every input goes to HBM once, in the order the tick needs it (Run1Feed): stage 0 = what the slab count reads,
  // 1 = what the feature engineering reads, 3 = what only the walk reads - sixteen 3-D fields, of which the slab's
  // levels cross, once the slab is known, and three 2-D ones - 2 = what the last kernel of the tick reads (the model's
  // own T, Q and the default OH, on every level), which crosses last, under the walk (a 48 x 24 x 72 block: 51 of 72 levels, 1.5 of 10.3 MB less over PCIe).  An array that
  // is passed twice (ONLINE_INST: T is both the model's and Boost's, OH_GridCompMod.F90:1326-1349) crosses once.
  struct In { const float* host; const float** dev; size_t n; int stage; };
  In ins[] = {
      {h.ple_mod, &d.ple_mod, edge, 0}, {h.tropp_mod, &d.tropp_mod, plane, 0},
      {h.ple_bst, &d.ple_bst, edge, 1}, {h.zle_bst, &d.zle_bst, edge, 1}, {h.tauclw, &d.tauclw, vol, 1}, {h.taucli, &d.taucli, vol, 1},
      {h.scacoef[0], &d.scacoef[0], vol, 1}, {h.scacoef[1], &d.scacoef[1], vol, 1}, {h.scacoef[2], &d.scacoef[2], vol, 1},
      {h.scacoef[3], &d.scacoef[3], vol, 1}, {h.scacoef[4], &d.scacoef[4], vol, 1}, {h.scacoef[5], &d.scacoef[5], vol, 1},
      {h.scacoef[6], &d.scacoef[6], vol, 1}, {h.gmito3, &d.gmito3, plane, 1}, {h.gmitto3, &d.gmitto3, plane, 1},
      {h.t_mod, &d.t_mod, vol, 2}, {h.q_mod, &d.q_mod, vol, 2}, {h.default_oh, &d.default_oh, vol, 2},
      {h.lat_deg, &d.lat_deg, plane, 3}, {h.albuv, &d.albuv, plane, 3}, {h.sza, &d.sza, plane, 3},
      {h.t_bst, &d.t_bst, vol, 3}, {h.no2, &d.no2, vol, 3}, {h.o3, &d.o3, vol, 3}, {h.ch4, &d.ch4, vol, 3}, {h.co, &d.co, vol, 3},
      {h.isop, &d.isop, vol, 3}, {h.acet, &d.acet, vol, 3}, {h.c2h6, &d.c2h6, vol, 3}, {h.c3h8, &d.c3h8, vol, 3},
      {h.prpe, &d.prpe, vol, 3}, {h.alk4, &d.alk4, vol, 3}, {h.mp, &d.mp, vol, 3}, {h.h2o2, &d.h2o2, vol, 3},
      {h.cloud, &d.cloud, vol, 3}, {h.qv, &d.qv, vol, 3}, {h.ch2o, &d.ch2o, vol, 3}};
  constexpr size_t nin = sizeof(ins) / sizeof(ins[0]);
  if (b->d_run1_stage.size() < nin + 3 + 9) b->d_run1_stage.resize(nin + 3 + 9);
  // an array handed over twice is staged once: the later entry takes the earlier one's device copy - and its stage,
  // if that one crosses in full (a stage-3 entry shares only with a full copy, never the other way round)
  int same_as[nin];
  bool whole[nin];                       // crosses in full even at stage 3
  for (size_t i = 0; i < nin; ++i) whole[i] = ins[i].stage != 3 || ins[i].n == plane;
  for (size_t i = 0; i < nin; ++i) {
    if (ins[i].host == nullptr) throw OhxError("OHXBoosterRun1: a required field pointer is NULL");
    same_as[i] = -1;
    for (size_t q = 0; q < i && same_as[i] < 0; ++q)
      if (ins[q].host == ins[i].host && ins[q].n == ins[i].n && same_as[q] < 0 && ins[q].stage != 3) same_as[i] = (int)q;
    if (same_as[i] >= 0) {
      *ins[i].dev = b->d_run1_stage[(size_t)same_as[i]].p;
      // the walk reads it too: it crosses with the walk's inputs, in full (ins[q].n == vol keeps it whole below)
      if (ins[i].stage == 3 && ins[(size_t)same_as[i]].stage == 2) {
        ins[(size_t)same_as[i]].stage = 3;
        whole[(size_t)same_as[i]] = true;
      }
      continue;
    }
    b->d_run1_stage[i].ensure(ins[i].n);
    *ins[i].dev = b->d_run1_stage[i].p;
  }
  struct Feed : Run1Feed {
    BoosterObj* b;
    In* ins;
    const int* same_as;
    const bool* whole;
    size_t plane, km;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    uint32_t post_blocks = 0;
    bool post_by_dma = false, all_by_dma = false;
    void stage(int which, int k1, int k2) {
      HostMover in(b->s_copy, true);
      in.by_dma = all_by_dma;
      if (which == 2) {
        in.blocks_override = post_blocks;
        in.by_dma = post_by_dma || all_by_dma;
      }
      for (size_t i = 0; i < nin; ++i) {
        if (ins[i].stage != which || same_as[i] >= 0) continue;
        float* dev = b->d_run1_stage[i].p;
        if (!whole[i]) {
          if (k2 >= k1) in.add_slice(dev + plane * (size_t)(k1 - 1), ins[i].host, plane * km, plane * (size_t)(k1 - 1), plane * (size_t)(k2 - k1 + 1));
        } else {
          in.add(dev, ins[i].host, ins[i].n);
        }
      }
      in.go();
    }
    hipEvent_t slab_inputs() override {
      stage(0, 0, 0);
      HIP_CHECK(hipEventRecord(ev[0], b->s_copy));
      return ev[0];
    }
    hipEvent_t prep_inputs() override {
      stage(1, 0, 0);
      HIP_CHECK(hipEventRecord(ev[1], b->s_copy));
      return ev[1];
    }
    hipEvent_t walk_inputs(int k1, int k2) override {
      stage(3, k1, k2);
      HIP_CHECK(hipEventRecord(ev[2], b->s_copy));
      return ev[2];
    }
    hipEvent_t post_inputs() override {          // the model's T, Q and the default OH cross under the walk
      stage(2, 0, 0);
      HIP_CHECK(hipEventRecord(ev[3], b->s_copy));
      return ev[3];
    }
    bool in_place = true, gate = false;
    bool slab_in_place(const float** ple, const float** tropp) override {
      if (!in_place) return false;
      void *mp = nullptr, *mt = nullptr;           // ins[0], ins[1]: the model's PLE and TROPP
      if (!g_host_registry.want(ins[0].host, ins[0].n * sizeof(float), &mp) || mp == nullptr) return false;
      if (!g_host_registry.want(ins[1].host, ins[1].n * sizeof(float), &mt) || mt == nullptr) return false;
      *ple = static_cast<const float*>(mp);
      *tropp = static_cast<const float*>(mt);
      ins[0].stage = ins[1].stage = 1;             // their device copies (the last kernel's) cross with the next list
      return true;
    }
    bool host_gated() const override { return gate; }
  } feed;
  feed.b = b;
  feed.ins = ins;
  feed.same_as = same_as;
  feed.whole = whole;
  feed.plane = plane;
  feed.km = (size_t)h.km;
  while (b->run1_feed_events.size() < 4) {
    hipEvent_t e;
    // with timing: such a record is a packet with a completion signal of its own.  Recorded without, the waits on these
    // events were released by the NEXT copy's end (the feature kernels started when list B had landed, 80 us late:
    // profiles/r05_sweeps.txt)
    HIP_CHECK(hipEventCreate(&e));
    b->run1_feed_events.push_back(e);
  }
  for (int i = 0; i < 4; ++i) feed.ev[i] = b->run1_feed_events[(size_t)i];
  if (h.oh == nullptr) throw OhxError("OHXBoosterRun1: oh is NULL");
  b->d_run1_stage[nin].ensure(vol);
  d.oh = b->d_run1_stage[nin].p;
  d.ndwet = nullptr;
  if (h.ndwet) {
    b->d_run1_stage[nin + 1].ensure(vol);
    d.ndwet = b->d_run1_stage[nin + 1].p;
  }
  d.oh_boost = nullptr;
  if (h.oh_boost) {
    b->d_run1_stage[nin + 2].ensure(vol);
    d.oh_boost = b->d_run1_stage[nin + 2].p;
  }
  // DIAG dumps: device copies only for the ones asked for
  struct Out { float* host; float** dev; size_t n; };
  Out outs[] = {{h.diag_pl_bst, &d.diag_pl_bst, vol}, {h.diag_tauclwdn, &d.diag_tauclwdn, vol},
                {h.diag_tauclidn, &d.diag_tauclidn, vol}, {h.diag_taucliup, &d.diag_taucliup, vol},
                {h.diag_tauclwup, &d.diag_tauclwup, vol}, {h.diag_aodup, &d.diag_aodup, vol},
                {h.diag_aoddn, &d.diag_aoddn, vol}, {h.diag_aod, &d.diag_aod, vol}, {h.diag_strato3, &d.diag_strato3, plane}};
  const size_t nout = sizeof(outs) / sizeof(outs[0]);
  if (b->d_run1_stage.size() < nin + 3 + nout) b->d_run1_stage.resize(nin + 3 + nout);
  for (size_t i = 0; i < nout; ++i) {
    *outs[i].dev = nullptr;
    if (outs[i].host == nullptr) continue;
    b->d_run1_stage[nin + 3 + i].ensure(outs[i].n);
    *outs[i].dev = b->d_run1_stage[nin + 3 + i].p;
  }
  // experiment knobs, read once (profiles/r05_sweeps.txt has what each is worth): OHX_RUN1_STREAMS=1 puts the copies on
  // the kernels' stream; OHX_RUN1_GATE=1 launches every kernel when this thread has seen its list's event instead of
  // enqueueing it behind a wait on the event; OHX_RUN1_SLAB_IN_PLACE=1 lets the slab count read PLE and TROPP over PCIe
  static const int nstreams = [] { const char* e = getenv("OHX_RUN1_STREAMS"); return e ? atoi(e) : 2; }();
  // (both measured worth nothing on a rank's block - 0.312 ms off / off, 0.314 in place, 0.319 gated and in place: the front
  // of a tick is bound by the link, 9.4 MB in 178 us, whatever the kernels wait for - and so both are off)
  static const bool knob_gate = [] { const char* e = getenv("OHX_RUN1_GATE"); return e && e[0] == '1'; }();
  static const bool knob_in_place = [] { const char* e = getenv("OHX_RUN1_SLAB_IN_PLACE"); return e && e[0] == '1'; }();
  feed.gate = knob_gate && nstreams != 1 && g_host_registry.on.load(std::memory_order_relaxed);
  feed.in_place = knob_in_place;
  // The last list of a tick (what only the mask and the conversion read) crosses UNDER the walk, and a copy kernel is bad
  // company for a walk: with the usual 64 blocks the walk "started" when the 20 us copy ended (its packet and arguments are
  // fetched over the link, behind the copy's queued reads), with 8 blocks it started at once and took 116 us instead of 66
  // (waves that share a CU with copy waves waiting on PCIe wait with them: a CU's vector memory path returns in order)
  // (profiles/r05_run1_timeline_block_48x24.txt, r05_sweeps.txt: "copy kernels on CUs of their own" tells the two apart).  So
  // this one list goes through the DMA engines (three hipMemcpyAsync of registered arrays, 11 us each, no wave on any
  // CU): a rank's tick 0.301 -> 0.293-0.297 ms.  OHX_COPY_POST_BLOCKS (read once): -1 = DMA (default), 0 = a copy kernel
  // like the other lists, n = one of n blocks.
  static const int knob_post_blocks = [] { const char* e = getenv("OHX_COPY_POST_BLOCKS"); return e ? atoi(e) : -1; }();
  feed.post_blocks = knob_post_blocks > 0 ? (uint32_t)knob_post_blocks : 0u;
  feed.post_by_dma = knob_post_blocks < 0;
  // ohx_copy_engine: dma, or auto's verdict (or its trial's turn) for a tick on registered arrays
  const int engine = g_copy_engine.load(std::memory_order_relaxed);
  const bool trying = engine == kCopyAuto && g_host_registry.on.load(std::memory_order_relaxed);
  feed.all_by_dma = engine == kCopyDma || (trying && b->copy_trial.dma_now(vol));
  const auto tick_began = std::chrono::steady_clock::now();
  hipStream_t main = nstreams == 1 ? b->s_copy : b->s_exec;
  // OHX_RUN1_TRACE=<n> (1 = 50): the host's time marks of ticks n .. n + 2 of this process on stderr
  static const int trace_from = [] { const char* e = getenv("OHX_RUN1_TRACE"); const int n = e ? atoi(e) : -1; return n == 1 ? 50 : n; }();
  const bool tracing = trace_from >= 0;
  static std::atomic<unsigned> tick_no{0};
  TickTrace trace;
  // however this call ends - judge_flags / raise_flag_errors throw after the read-back too - the thread's pointer to
  // `trace` does not outlive it
  struct TraceReset { ~TraceReset() { g_tick_trace = nullptr; } } trace_reset;
  const unsigned tick = tick_no.fetch_add(1);
  g_tick_trace = (tracing && tick >= (unsigned)trace_from && tick <= (unsigned)trace_from + 2u) ? &trace : nullptr;
  TICK_MARK("set-up");
  try {
    run1_device(*b, d, main, &feed);
  } catch (...) {
    // nothing of this call may still be reading the caller's arrays when it returns, error or not
    (void)hipStreamSynchronize(b->s_copy);
    (void)hipStreamSynchronize(main);
    g_tick_trace = nullptr;
    throw;
  }
  HostMover back(main, false);
  back.by_dma = feed.all_by_dma;
  back.add(h.oh, d.oh, vol);
  if (h.ndwet) back.add(h.ndwet, d.ndwet, vol);
  if (h.oh_boost) back.add(h.oh_boost, d.oh_boost, vol);
  for (size_t i = 0; i < nout; ++i)
    if (outs[i].host) back.add(outs[i].host, *outs[i].dev, outs[i].n);
  // the kernels' flag words come back on the same launch (a read-back of their own was one more launch at a tick's end)
  b->h_flags.ensure(2);
  void* flags_there = nullptr;          // where the GPU reaches the pinned words
  if (hipHostGetDevicePointer(&flags_there, b->h_flags.p, 0) != hipSuccess) {
    (void)hipGetLastError();
    flags_there = nullptr;
  }
  const bool flags_ride = flags_there != nullptr && back.ride(flags_there, b->d_flags.p, 2);
  back.go();
  TICK_MARK("all-enqueued");
  if (flags_ride) {
    HIP_CHECK(hipStreamSynchronize(main));          // the outputs are in the caller's arrays
    const uint32_t flags[2] = {b->h_flags.p[0], b->h_flags.p[1]};
    judge_flags(*b, flags, main);
  } else {
    raise_flag_errors(*b, main);        // waits for the stream
  }
  TICK_MARK("done");
  if (trying) b->copy_trial.report(feed.all_by_dma, std::chrono::duration<double>(std::chrono::steady_clock::now() - tick_began).count());
  if (g_tick_trace) g_tick_trace->print(tick);
  g_tick_trace = nullptr;
  API_END();
}

int OHXBoosterGetInfo(BoosterHandle handle, bst_ulong info[8]) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (info == nullptr) throw OhxError("OHXBoosterGetInfo: info is NULL");
  if (!b->loaded) throw OhxError("the booster holds no model");
  // host-only: placement can be computed without a device
  Placement local;
  const Placement* p = &b->placement;
  bool packed_ok = b->packed_ok;
  if (!b->uploaded) {
    local = place_forest(b->forest, b->layout);
    p = &local;
    packed_ok = packed_format_fits(b->forest, local);
  }
  const bool packed_used = packed_ok && b->kernel_name != "wide";
  bool super_used = false;
  uint64_t super_slots = b->super_slots;
  // (the ring kernels where the booster's big batches go through them: pick_kernel)
  auto walk_mode = [&](double mean_steps) {
    const bool ring = b->kernel_name == "ring" ||
                      (b->kernel_name == "auto" && b->forest.num_feature == 27 && mean_steps >= kRingMinMeanSteps);
    return ring ? 2 : (use_tree_tops(b->tune, mean_steps) ? 1 : 0);
  };
  uint64_t super_gathers = b->super_gathers[walk_mode(b->super_mean_steps)];
  if (wants_super(b->kernel_name)) {
    if (b->uploaded) {
      super_used = b->super_ok;
    } else {
      SuperForest sf;
      super_used = emit_super(b->forest, &sf);
      super_slots = sf.nodes.size();
      super_gathers = count_super_gathers(sf, walk_mode(mean_super_steps(sf)));
    }
  }
  info[0] = b->forest.trees.size();
  info[1] = p->real_nodes;
  info[2] = super_used ? super_slots : p->num_slots;
  info[3] = super_used ? super_slots * sizeof(SuperNode)
                       : p->num_slots * (packed_used ? sizeof(PackedNode) : sizeof(WideNode));
  info[4] = (bst_ulong)p->max_depth;
  info[5] = b->forest.num_feature;
  info[6] = super_used ? 2 : (packed_used ? 1 : 0);
  info[7] = super_used ? super_gathers : 0;
  API_END();
}

int OHXBoosterKernelSymbol(BoosterHandle handle, bst_ulong ncol, const char** out) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (out == nullptr) throw OhxError("OHXBoosterKernelSymbol: out is NULL");
  ensure_uploaded(*b);
  b->symbol = predict_kernel_symbol(pick_kernel(*b), device_forest(*b), (uint32_t)ncol, b->tune);
  *out = b->symbol.c_str();
  API_END();
}

int OHXBoosterKernelSymbolRows(BoosterHandle handle, DMatrixHandle dmat, const char** out) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  DMatrixObj* d = as_dmat(dmat);
  if (out == nullptr) throw OhxError("OHXBoosterKernelSymbolRows: out is NULL");
  check_columns(*b, d->ncol);
  ensure_uploaded(*b);
  PredictArgs a;
  a.rows = d->d_data;
  a.nrow = d->nrow;
  a.ncol = (uint32_t)d->ncol;
  a.missing = d->missing;
  tree_range(*b, 0, &a.tree_begin, &a.tree_end);
  LaunchTuning tune = b->tune;
  tune.grid_im = d->grid_im;
  tune.grid_jm = d->grid_jm;
  tune.grid_row0 = d->grid_row0;
  leaf_room(*b, d->nrow, a.tree_end - a.tree_begin, tune, tune.grid_im, tune.grid_jm, tune.grid_row0, /*plan_only=*/true);
  if (d->ncol == 27) (void)defer_plan(*b, d->nrow, tune);
  b->symbol = predict_kernel_symbols_rows(pick_kernel(*b), device_forest(*b), a, b->dev.num_cus, tune);
  *out = b->symbol.c_str();
  API_END();
}

int OHXBoosterRingReruns(BoosterHandle handle, void* stream, bst_ulong* out) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  if (out == nullptr) throw OhxError("OHXBoosterRingReruns: out is NULL");
  *out = 0;
  if (b->uploaded) {
    uint32_t n = 0;
    HIP_CHECK(hipMemcpyAsync(&n, b->d_flags.p + 1, sizeof(n), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    *out = n;
  }
  API_END();
}

int OHXBoosterCopyEngineChoice(BoosterHandle handle, int* choice, unsigned* trials, unsigned* picked_dma) {
  API_BEGIN();
  BoosterObj* b = as_booster(handle);
  const bool in_charge = g_copy_engine.load(std::memory_order_relaxed) == kCopyAuto && g_host_registry.on.load(std::memory_order_relaxed);
  if (choice) *choice = in_charge ? b->copy_trial.choice : -1;
  if (trials) *trials = b->copy_trial.trials;
  if (picked_dma) *picked_dma = b->copy_trial.picked_dma;
  API_END();
}

int OHXUnregisterHost(const void* array) {
  API_BEGIN();
  if (array != nullptr) (void)g_host_registry.forget(array);
  API_END();
}

int OHXReleaseScratch(void) {
  API_BEGIN();
  g_row_pool.release_all();
  g_host_registry.release_all();
  {
    std::lock_guard<std::mutex> g(g_post.mu);
    for (auto& bf : g_post.buf) bf.release();
  }
  API_END();
}

}  // extern "C"
#pragma GCC visibility pop
