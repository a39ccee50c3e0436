// l1_tag_microbench — which addresses of ONE wave64 gather collide in the vector L1's tag look-up on gfx950
// (TCP_READ_TAGCONFLICT_STALL_CYCLES is 21 % of the CU-cycles under predict_rows_ring_kernel).
//
// The walk's deep steps are 16-byte gathers in which the four lanes of a quad (four k-adjacent gridcells) stand on 1-4
// distinct 64-byte blocks (a block = the four child super-nodes of one parent; 2.4 distinct per quad on average,
// tests/analysis/quad_lookups.py).  Here every quad reads D distinct blocks laid out base + j * stride, base random per
// quad and gather, for stride = 64 B ... 64 KiB; plus the two ends: one block per quad, four random blocks per quad.
// If the tag RAM is banked / set-indexed by address bits the stride sweeps over, the cycles per gather (and the
// counter, when run under rocprofv3 --pmc TCP_READ_TAGCONFLICT_STALL_CYCLES: one dispatch per printed row, in order)
// step up where the D blocks of a quad - or the blocks of neighbouring quads - fall into the same bank.
// A second sweep keeps the quads' blocks apart by a fixed stride BETWEEN quads (all sixteen quads one block each).
// Standalone: hipcc --offload-arch=gfx950 -O3 tools/l1_tag_microbench.hip -o tools/bin/l1_tag_microbench
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

// mode 0: quad-internal: lane j of a quad reads block  base(quad) + (j % distinct) * stride      (base random, aligned to 4 * stride)
// mode 1: wave-internal: quad q reads block            base(wave) + q * stride                   (one block per quad)
// mode 2: four random blocks per quad;  mode 3: one random block per quad
// mode 4: one random block per quad inside one random window per wave;  mode 5: mode 1 with strides that are no powers of two
// all units: 16-byte elements (uint4); a block is 4 elements; the slot inside the block is the lane's own
__global__ __launch_bounds__(256) void tag_kernel(const uint4* __restrict__ table, uint32_t mask, int iters, int mode,
                                                  uint32_t stride_el, uint32_t distinct, uint32_t* __restrict__ sink) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t quad = lane >> 2, j = lane & 3u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  uint32_t s = mix(wave * 0x9E3779B1u + 12345u);
  const uint32_t slot = mix(lane * 0x27D4EB2Fu) & 3u;
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s = s * 1664525u + 1013904223u;      // wave-uniform stream
      uint32_t idx;
      if (mode == 0) {
        const uint32_t base = mix(s ^ (quad * 0x85EBCA77u)) & ~(4u * stride_el - 1u);
        idx = base + (j % distinct) * stride_el;
      } else if (mode == 1) {
        const uint32_t base = mix(s) & ~(16u * stride_el - 1u);
        idx = base + quad * stride_el;
      } else if (mode == 4) {
        // every quad a random block inside ONE random window of the wave, stride_el elements wide (aligned)
        const uint32_t base = mix(s) & ~(stride_el - 1u);
        idx = base + (mix(s ^ (quad * 0x85EBCA77u)) & (stride_el - 1u) & ~3u);
      } else if (mode == 5) {
        // as mode 1 with a stride that is no power of two: quad q reads block base + q * stride, base random, 64-byte aligned
        idx = (mix(s) & ~3u) + quad * stride_el;
      } else if (mode == 2) {
        idx = mix(s ^ (lane * 0x85EBCA77u)) & ~3u;
      } else {
        idx = mix(s ^ (quad * 0x85EBCA77u)) & ~3u;
      }
      const uint4 v = table[(idx + slot) & mask];
      acc ^= v.x;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;   // keep the loads alive
}

static double run(const uint4* table, size_t table_bytes, int blocks, int iters, int mode, uint32_t stride_bytes, uint32_t distinct) {
  const uint32_t mask = (uint32_t)(table_bytes / 16) - 1u;
  uint32_t* sink;
  CHECK(hipMalloc(&sink, 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(tag_kernel, dim3(blocks), dim3(256), 0, 0, table, mask, iters / 8, mode, stride_bytes / 16, distinct, sink);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(tag_kernel, dim3(blocks), dim3(256), 0, 0, table, mask, iters, mode, stride_bytes / 16, distinct, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipFree(sink));
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
  return ms * 1e-3;
}

int main(int argc, char** argv) {
  int cus = 0, clock_khz = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
  const size_t table_bytes = argc > 1 ? (size_t)atol(argv[1]) << 10 : (size_t)1 << 20;       // KiB on the command line; 1 MiB: L2-resident
  uint4* table;
  CHECK(hipMalloc(&table, table_bytes));
  std::vector<uint32_t> h(table_bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) | 1u;
  CHECK(hipMemcpy(table, h.data(), table_bytes, hipMemcpyHostToDevice));
  const int waves_per_cu = 16, blocks = cus * waves_per_cu / 4, iters = 2048;
  const double wave_instr_per_cu = (double)waves_per_cu * iters;
  printf("# gfx950 L1 tag look-up microbenchmark: %d CUs, %d waves/CU, %d 16-byte gathers per wave, table %zu KiB, clock %d MHz (nominal)\n",
         cus, waves_per_cu, iters, table_bytes >> 10, clock_khz / 1000);
  printf("# every printed row is two dispatches of tag_kernel (a short warm-up, then the timed one)\n");
  printf("%-34s %10s %10s\n", "pattern", "ms", "cyc/gather");
  auto row = [&](const char* name, int mode, uint32_t stride, uint32_t distinct) {
    const double s = run(table, table_bytes, blocks, iters, mode, stride, distinct);
    printf("%-34s %10.3f %10.1f\n", name, s * 1e3, s * (clock_khz * 1e3) / wave_instr_per_cu);
    fflush(stdout);
  };
  row("one random block per quad", 3, 64, 1);
  row("four random blocks per quad", 2, 64, 4);
  char name[96];
  for (uint32_t distinct = 2; distinct <= 4; distinct += 2)
    for (uint32_t stride = 64; stride <= (64u << 10) && 4 * (size_t)stride <= table_bytes / 4; stride *= 2) {
      snprintf(name, sizeof name, "quad: %u blocks, stride %u B", distinct, stride);
      row(name, 0, stride, distinct);
    }
  for (uint32_t stride = 64; stride <= (16u << 10) && 16 * (size_t)stride <= table_bytes / 4; stride *= 2) {
    snprintf(name, sizeof name, "wave: 16 blocks, stride %u B", stride);
    row(name, 1, stride, 1);
  }
  for (uint32_t win = 1024; win <= (1u << 20) && win <= table_bytes; win *= 4) {
    snprintf(name, sizeof name, "wave: 16 random blocks in %u KiB", win >> 10);
    row(name, 4, win, 1);
  }
  for (uint32_t stride : {192u, 320u, 576u, 1088u, 2112u, 4160u}) {
    snprintf(name, sizeof name, "wave: 16 blocks, stride %u B", stride);
    row(name, 5, stride, 1);
  }
  return 0;
}
