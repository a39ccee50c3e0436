// gather_microbench — what one wave64 gather instruction costs on gfx950, by access
// width, by how the 64 lanes' addresses are spread, and by where the table lives
// (L1 / L2 / Infinity Cache / HBM).  The tree walk is nothing but such gathers, so
// this table is what the node layout and the kernel structure are designed against
// (DESIGN.md §4).  Standalone: hipcc --offload-arch=gfx950 -O3 gather_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                    \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// PATTERN 0: every lane the same element; 1: 64 consecutive elements (coalesced);
//         2: every lane its own random element; 3: random 128-B line per group of 4 lanes
//         4: random element inside ONE random 1-KiB window per wave (8 lines)
template <typename T, int PATTERN>
__global__ __launch_bounds__(256) void gather_kernel(const T* __restrict__ table, uint32_t mask, int iters,
                                                     uint32_t* __restrict__ sink) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  uint32_t s = mix(wave * 0x9E3779B1u + 12345u);
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s = s * 1664525u + 1013904223u;      // wave-uniform stream
      uint32_t idx;
      if (PATTERN == 0) idx = mix(s);
      else if (PATTERN == 1) idx = (mix(s) & ~63u) + lane;
      else if (PATTERN == 2) idx = mix(s ^ (lane * 0x85EBCA77u));
      else if (PATTERN == 3) idx = (mix(s ^ ((lane >> 2) * 0x85EBCA77u)) & ~3u) + (lane & 3u);
      else idx = (mix(s) & ~(uint32_t)(1024 / sizeof(T) - 1)) + (mix(s ^ (lane * 0x85EBCA77u)) & (uint32_t)(1024 / sizeof(T) - 1));
      const T v = table[idx & mask];
      acc ^= reinterpret_cast<const uint32_t*>(&v)[0];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;   // keep the loads alive
}

template <typename T, int PATTERN>
double run(const void* table, size_t table_bytes, int blocks, int iters) {
  const uint32_t mask = (uint32_t)(table_bytes / sizeof(T)) - 1u;
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL((gather_kernel<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, (const T*)table, mask, iters / 8, sink);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL((gather_kernel<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, (const T*)table, mask, iters, sink);
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
  const size_t max_bytes = 1ull << 30;
  uint32_t* table;
  CHECK(hipMalloc(&table, max_bytes));
  std::vector<uint32_t> h(max_bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) | 1u;
  CHECK(hipMemcpy(table, h.data(), max_bytes, hipMemcpyHostToDevice));
  const int waves_per_cu = 16;
  const int blocks = cus * waves_per_cu / 4;
  const int iters = 4096;
  const double wave_instr_per_cu = (double)waves_per_cu * iters;
  printf("# gfx950 gather microbenchmark: %d CUs, %d waves/CU, %d gathers per wave, clock %.0f MHz (nominal)\n", cus,
         waves_per_cu, iters, clock_khz / 1000.0);
  printf("# cyc = nominal-clock cycles per wave64 gather instruction per CU;  Glane/s = lane-requests per second, chip\n");
  printf("%-10s %-8s %-10s %10s %10s %12s\n", "table", "width", "pattern", "ms", "cyc/instr", "Glane/s");
  const size_t sizes[] = {16u << 10, 1u << 20, 16u << 20, 48u << 20, 1u << 30};
  const char* size_names[] = {"16KiB", "1MiB", "16MiB", "48MiB", "1GiB"};
  const char* pat_names[] = {"same", "coalesced", "random", "rand-quad", "rand-1KiB"};
  for (int si = 0; si < 5; ++si) {
    for (int w = 0; w < 3; ++w) {
      for (int p = 0; p < 5; ++p) {
        double s = 0;
#define RUN(T, P) s = run<T, P>(table, sizes[si], blocks, iters)
        if (w == 0) { if (p == 0) RUN(uint32_t, 0); else if (p == 1) RUN(uint32_t, 1); else if (p == 2) RUN(uint32_t, 2); else if (p == 3) RUN(uint32_t, 3); else RUN(uint32_t, 4); }
        if (w == 1) { if (p == 0) RUN(uint2, 0); else if (p == 1) RUN(uint2, 1); else if (p == 2) RUN(uint2, 2); else if (p == 3) RUN(uint2, 3); else RUN(uint2, 4); }
        if (w == 2) { if (p == 0) RUN(uint4, 0); else if (p == 1) RUN(uint4, 1); else if (p == 2) RUN(uint4, 2); else if (p == 3) RUN(uint4, 3); else RUN(uint4, 4); }
        const double cyc = s * (clock_khz * 1e3) / wave_instr_per_cu;
        const double glane = (double)cus * wave_instr_per_cu * 64.0 / s / 1e9;
        printf("%-10s %-8s %-10s %10.3f %10.1f %12.1f\n", size_names[si], w == 0 ? "4B" : (w == 1 ? "8B" : "16B"),
               pat_names[p], s * 1e3, cyc, glane);
      }
    }
  }
  return 0;
}
