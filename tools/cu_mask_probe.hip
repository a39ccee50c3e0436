// Which CUs do the bits of a stream's CU mask name on this part?  (measurement aid for DESIGN.md 10.4: copy kernels on CUs
// of their own.)  A kernel of many one-wave blocks writes where each block ran (XCC_ID, and SE / SH / CU of HW_ID); run
// once on a plain stream and once on streams made by hipExtStreamCreateWithCUMask with EIGHT bits cleared - never more, so
// that under any layout of the bits every XCD keeps most of its CUs and no workgroup can be left without a place to run.
// Prints, per mask, the (xcc, se, cu) places that the plain stream used and the masked one did not.
// build: hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe      usage (GPU box): tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <tuple>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where(uint32_t* out) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // a little work, so that blocks spread over the chip instead of all finishing on the first CUs
  float x = (float)threadIdx.x;
  for (int i = 0; i < 2000; ++i) x = x * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = (xcc & 0xFu) | (x == 12345.0f ? 0x80000000u : 0u);
  }
}

using Place = std::tuple<unsigned, unsigned, unsigned, unsigned>;      // xcc, se, sh, cu

static int run(hipStream_t s, uint32_t* d, std::set<Place>* places) {
  const int nblocks = 16384;
  hipLaunchKernelGGL(where, dim3(nblocks), dim3(64), 0, s, d);
  CHECK(hipGetLastError());
  CHECK(hipStreamSynchronize(s));
  std::vector<uint32_t> h(2 * nblocks);
  CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  for (int b = 0; b < nblocks; ++b) {
    const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xFu;
    places->insert(Place{xcc, (hw >> 13) & 7u, (hw >> 12) & 1u, (hw >> 8) & 15u});
  }
  return 0;
}

int main() {
  uint32_t* d = nullptr;
  CHECK(hipMalloc(&d, 2 * 16384 * 4));
  hipStream_t plain;
  CHECK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  std::set<Place> all;
  if (run(plain, d, &all)) return 1;
  printf("plain stream: %zu places (xcc, se, sh, cu)\n", all.size());
  unsigned per_xcc[16] = {0};
  for (const Place& p : all) per_xcc[std::get<0>(p)]++;
  for (int x = 0; x < 16; ++x) if (per_xcc[x]) printf("  xcc %d: %u CUs\n", x, per_xcc[x]);
  const int first_bits[] = {0, 8, 32, 128};
  for (int fb : first_bits) {
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) mask[w] = 0xFFFFFFFFu;
    for (int bit = fb; bit < fb + 8; ++bit) mask[bit / 32] &= ~(1u << (bit % 32));
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("mask without bits %d..%d: hipExtStreamCreateWithCUMask: %s\n", fb, fb + 7, hipGetErrorString(e)); continue; }
    std::set<Place> got;
    if (run(s, d, &got)) return 1;
    printf("mask without bits %3d..%3d: %zu places; not used:", fb, fb + 7, got.size());
    for (const Place& p : all)
      if (!got.count(p)) printf(" (%u,%u,%u,%u)", std::get<0>(p), std::get<1>(p), std::get<2>(p), std::get<3>(p));
    printf("\n");
    CHECK(hipStreamDestroy(s));
  }
  return 0;
}
