// exec_mask_microbench — does a wave64 gather cost the texture addresser less when fewer lanes are active?
// The idea under test (profiles/r02_sweeps.txt): the eight candidate super-nodes of a tree's second step could be
// fetched by 8 lanes and handed to the others through LDS, if an 8-lane load is cheaper than a 64-lane one.
// 16-byte loads of consecutive elements from a table that sits in L1 (16 KiB) or L2 (1 MiB); lanes switched off
// by a branch.  Standalone: hipcc --offload-arch=gfx950 -O3 exec_mask_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                              \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                \
    }                                                         \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// ACTIVE 0: all 64 lanes; 1: lanes 0-31; 2: lanes 0-7; 3: lane 0; 4: one lane per quad (16 lanes); 5: lanes 0-15
template <int ACTIVE, bool VIA_LDS>
__global__ __launch_bounds__(256) void kernel(const uint4* __restrict__ table, uint32_t mask, int iters,
                                              uint32_t* __restrict__ sink) {
  __shared__ uint4 stage[4][8];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_in_block = threadIdx.x >> 6;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  bool on = true;
  if (ACTIVE == 1) on = lane < 32;
  if (ACTIVE == 2) on = lane < 8;
  if (ACTIVE == 3) on = lane == 0;
  if (ACTIVE == 4) on = (lane & 3u) == 0;
  if (ACTIVE == 5) on = lane < 16;
  uint32_t acc = 0;
  uint32_t s = mix(wave * 0x9E3779B1u + 12345u);
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t base = mix(s) & ~63u;
    if (VIA_LDS) {
      // 8 lanes fetch 8 consecutive elements, everybody reads "its" one (lane & 7) back from LDS
      if (lane < 8) stage[wave_in_block][lane] = table[(base + lane) & mask];
      const uint4 v = stage[wave_in_block][(lane * 5u + it) & 7u];
      acc ^= v.x;
    } else if (on) {
      const uint4 v = table[(base + lane) & mask];
      acc ^= v.x;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int ACTIVE, bool VIA_LDS>
double run(const uint4* table, size_t bytes, int blocks, int iters) {
  const uint32_t mask = (uint32_t)(bytes / 16) - 1u;
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((kernel<ACTIVE, VIA_LDS>), dim3(blocks), dim3(256), 0, 0, table, mask, iters / 8, sink);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((kernel<ACTIVE, VIA_LDS>), dim3(blocks), dim3(256), 0, 0, table, mask, iters, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipFree(sink));
  return ms * 1e-3;
}

int main() {
  int cus = 0, clock_khz = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
  const size_t max_bytes = 1u << 20;
  uint4* table;
  CHECK(hipMalloc(&table, max_bytes));
  std::vector<uint32_t> h(max_bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) | 1u;
  CHECK(hipMemcpy(table, h.data(), max_bytes, hipMemcpyHostToDevice));
  const int waves_per_cu = 16, blocks = cus * waves_per_cu / 4, iters = 4096;
  const double per_cu = (double)waves_per_cu * iters;
  printf("# gfx950: cycles (nominal %d MHz) per wave64 global_load_dwordx4 per CU, by active lanes; %d waves/CU\n",
         clock_khz / 1000, waves_per_cu);
  const size_t sizes[] = {16u << 10, 1u << 20};
  const char* names[] = {"16KiB(L1)", "1MiB(L2)"};
  for (int si = 0; si < 2; ++si) {
    double t[7];
    t[0] = run<0, false>(table, sizes[si], blocks, iters);
    t[1] = run<1, false>(table, sizes[si], blocks, iters);
    t[2] = run<5, false>(table, sizes[si], blocks, iters);
    t[3] = run<2, false>(table, sizes[si], blocks, iters);
    t[4] = run<3, false>(table, sizes[si], blocks, iters);
    t[5] = run<4, false>(table, sizes[si], blocks, iters);
    t[6] = run<0, true>(table, sizes[si], blocks, iters);
    const char* what[] = {"64 lanes", "32 lanes", "16 lanes", "8 lanes", "1 lane", "16 lanes, one per quad",
                          "8 lanes + ds_write_b128 + ds_read_b128 by all"};
    for (int q = 0; q < 7; ++q)
      printf("%-10s %-46s %8.3f ms %8.1f cyc\n", names[si], what[q], t[q] * 1e3, t[q] * (clock_khz * 1e3) / per_cu);
  }
  return 0;
}
