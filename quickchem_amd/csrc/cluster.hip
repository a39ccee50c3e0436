// Rows nobody can describe: a clustering pass in front of the walk.
//
// What a tree walk costs on gfx950 is set by how many different node lines the lanes of a wave - and above
// all the four lanes of a quad - ask for.  Rows that come in grid order are tiled into bricks of
// neighbouring gridcells; rows in no particular order (a caller that shuffled, filtered or concatenated its
// gather: the DMatrix contract allows any order, OH_GridCompMod.F90:275-345 is just one caller) have no
// neighbours to offer, and 64 arbitrary rows per wave run at a third of the speed.  This pass finds them
// neighbours: every row gets a key made of the decisions it takes at the top of the first few trees of the
// booster itself, rows are grouped by key (one counting sort over the whole key), and the walk then takes 64
// rows of one group per wave through a row permutation.  Predictions cannot change - rows are independent and
// every row still walks every tree in order; only which rows share a wave does.
//
// All of it is plain integer work next to the walk: one read of the rows (the walk reads them again), two
// atomics and eight bytes per row.  Kernels: keys + histogram, a three-phase exclusive scan of the histogram,
// scatter of row numbers.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace ohx {

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr uint32_t kScanTile = 4096;     // counters one block scans (16 per thread)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool left_or_default(float x, float thr, bool default_left) {
  return (x != x) ? default_left : (x < thr);
}

// One lane = one row, read straight from the row-major matrix (27 floats; the few features the top of a few
// trees asks for come out of L1 after the first touch).  Key = for each of `ntrees` trees the root decision (trees
// whose super-nodes start below the root) and two decisions per super-node step, most significant first; a
// row that reaches a leaf early keeps walking on fixed decisions.  Also counts how many rows agree with the
// row before them on the first tree's part of the key: rows in grid order mostly do, shuffled rows mostly do not.
__global__ __launch_bounds__(kBlock) void cluster_keys_kernel(DeviceForest fr, ClusterArgs a) {
  const u32x4* __restrict__ nodes = reinterpret_cast<const u32x4*>(fr.super);
  const bool missing_is_nan = a.missing != a.missing;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t agree = 0;
  for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < a.nrow + (kWave - 1); row += stride) {
    const bool valid = row < a.nrow;
    uint32_t key = 0;
    uint32_t first_tree_part = 0;
    if (valid) {
      const float* x = a.rows + row * (uint64_t)a.ncol;
      auto feature = [&](uint32_t f) {
        float v = f < a.ncol ? x[f] : __builtin_nanf("");
        if (!missing_is_nan && v == a.missing) v = __builtin_nanf("");
        return v;
      };
      for (uint32_t t = 0; t < a.ntrees; ++t) {
        const SuperTreeHead h = fr.super_heads[t];
        uint32_t rel = 4u;
        if (h.root_meta & 0x100u) {           // phase 1: the root is evaluated from the head
          const bool l = left_or_default(feature(h.root_meta & 31u), h.root_thr, (h.root_meta & 32u) != 0u);
          key = (key << 1) | (l ? 0u : 1u);
          rel = 4u + (l ? 0u : 1u);
        }
        for (uint32_t s = 0; s < a.nsteps; ++s) {
          uint32_t bits = 0;
          if (s < h.steps) {
            const u32x4 nd = nodes[h.base + rel];
            const uint32_t w = nd.w;
            const uint32_t f0 = (w >> 8) & 31u;
            if (f0 != kSuperLeaf) {
              const bool l0 = left_or_default(feature(f0), __uint_as_float(nd.x), (w & 32u) != 0u);
              const uint32_t f1 = (w >> (l0 ? 0u : 13u)) & 31u;
              bool l1 = true;
              if (f1 != kSuperLeaf)
                l1 = left_or_default(feature(f1), __uint_as_float(l0 ? nd.y : nd.z), (w & (l0 ? 64u : 128u)) != 0u);
              bits = (l0 ? 0u : 2u) | (l1 ? 0u : 1u);
              rel = ((w >> 18) << 2) + bits;
            } else {
              rel = (w >> 18) << 2;        // fillers lead to fillers
            }
          }
          key = (key << 2) | bits;
        }
        if (t == 0) first_tree_part = key;
      }
      a.keys[row] = key;
      atomicAdd(&a.counters[key], 1u);
    }
    // neighbours in row order: lane l against lane l - 1 (the first lane of a wave sits out)
    const uint32_t prev = __shfl_up(first_tree_part, 1);
    const bool same = valid && (threadIdx.x & (kWave - 1)) != 0 && prev == first_tree_part;
    agree += __popcll(__ballot(same));
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && agree) atomicAdd(a.agree, agree);
}

// Exclusive scan of `n` counters (n a multiple of kScanTile), in place: block sums, scan of the sums, local scans.
__global__ __launch_bounds__(kBlock) void scan_block_sums_kernel(const uint32_t* __restrict__ counters,
                                                                 uint32_t* __restrict__ sums) {
  __shared__ uint32_t part[kBlock / kWave];
  const uint32_t* p = counters + (size_t)blockIdx.x * kScanTile + threadIdx.x * 16u;
  uint32_t s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32x4 v = reinterpret_cast<const u32x4*>(p)[q];
    s += v.x + v.y + v.z + v.w;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// one block: exclusive scan of up to 4096 block sums (16 per thread), in place
__global__ __launch_bounds__(kBlock) void scan_sums_kernel(uint32_t* __restrict__ sums, uint32_t n) {
  __shared__ uint32_t warp_tot[kBlock / kWave];
  uint32_t v[16];
  uint32_t s = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const uint32_t i = threadIdx.x * 16u + q;
    v[q] = i < n ? sums[i] : 0u;
    s += v[q];
  }
  // inclusive scan of the per-thread totals across the block
  uint32_t inc = s;
  const int lane = threadIdx.x & (kWave - 1);
  for (int off = 1; off < kWave; off <<= 1) {
    const uint32_t up = __shfl_up(inc, off);
    if (lane >= off) inc += up;
  }
  if (lane == kWave - 1) warp_tot[threadIdx.x / kWave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < (int)(threadIdx.x / kWave); ++w) base += warp_tot[w];
  uint32_t run = base + inc - s;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const uint32_t i = threadIdx.x * 16u + q;
    if (i < n) sums[i] = run;
    run += v[q];
  }
}

__global__ __launch_bounds__(kBlock) void scan_local_kernel(uint32_t* __restrict__ counters,
                                                            const uint32_t* __restrict__ sums) {
  __shared__ uint32_t warp_tot[kBlock / kWave];
  uint32_t* p = counters + (size_t)blockIdx.x * kScanTile + threadIdx.x * 16u;
  uint32_t v[16];
  uint32_t s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32x4 t = reinterpret_cast<const u32x4*>(p)[q];
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    s += t.x + t.y + t.z + t.w;
  }
  uint32_t inc = s;
  const int lane = threadIdx.x & (kWave - 1);
  for (int off = 1; off < kWave; off <<= 1) {
    const uint32_t up = __shfl_up(inc, off);
    if (lane >= off) inc += up;
  }
  if (lane == kWave - 1) warp_tot[threadIdx.x / kWave] = inc;
  __syncthreads();
  uint32_t run = sums[blockIdx.x] + inc - s;
  for (int w = 0; w < (int)(threadIdx.x / kWave); ++w) run += warp_tot[w];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const uint32_t c = v[q];
    p[q] = run;
    run += c;
  }
}

// perm[cursor[key]++] = row.  Which row of a group lands where does not matter.
__global__ __launch_bounds__(kBlock) void cluster_scatter_kernel(const uint32_t* __restrict__ keys, uint64_t nrow,
                                                                 uint32_t* __restrict__ cursor,
                                                                 uint32_t* __restrict__ perm) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrow; row += stride)
    perm[atomicAdd(&cursor[keys[row]], 1u)] = (uint32_t)row;
}

int blocks_for(uint64_t items, int num_cus, int per_cu) {
  uint64_t b = (items + kBlock - 1) / kBlock;
  const uint64_t cap = (uint64_t)num_cus * (uint64_t)per_cu;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

uint32_t cluster_key_bits(const ClusterArgs& a) { return a.ntrees * (1u + 2u * a.nsteps); }

hipError_t launch_cluster_keys(const DeviceForest& fr, const ClusterArgs& a, int num_cus, hipStream_t stream) {
  if (a.nrow == 0) return hipSuccess;
  if (fr.super == nullptr || cluster_key_bits(a) > 24u || a.ntrees == 0 || a.ntrees > fr.num_trees)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(cluster_keys_kernel, dim3(blocks_for(a.nrow, num_cus, 16)), dim3(kBlock), 0, stream, fr, a);
  return hipGetLastError();
}

hipError_t launch_cluster_sort(const ClusterArgs& a, uint32_t* d_block_sums, int num_cus, hipStream_t stream) {
  if (a.nrow == 0) return hipSuccess;
  uint64_t ncounters = 1ull << cluster_key_bits(a);
  if (ncounters < kScanTile) ncounters = kScanTile;           // the buffer is allocated to at least one tile
  const uint32_t nblocks = (uint32_t)(ncounters / kScanTile);
  if (nblocks > kScanTile) return hipErrorInvalidValue;       // one block scans the sums: at most 2**24 counters
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblocks), dim3(kBlock), 0, stream, (const uint32_t*)a.counters,
                     d_block_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kBlock), 0, stream, d_block_sums, nblocks);
  hipLaunchKernelGGL(scan_local_kernel, dim3(nblocks), dim3(kBlock), 0, stream, a.counters, (const uint32_t*)d_block_sums);
  hipLaunchKernelGGL(cluster_scatter_kernel, dim3(blocks_for(a.nrow, num_cus, 16)), dim3(kBlock), 0, stream,
                     (const uint32_t*)a.keys, a.nrow, a.counters, a.perm);
  return hipGetLastError();
}

}  // namespace ohx
