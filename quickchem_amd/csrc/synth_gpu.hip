// Synthetic inputs of SURVEY.md §8(d) generated in HBM - TEST AND BENCHMARK SUPPORT, not the product:
// built into lib/libohx_synth_gpu.so, which only tests/, bench.py and __graft_entry__.smoke() load
// (through quickchem_amd/synth.py).  libohxgb.so exports none of this.  Bit-identical to the host
// generator in libohx_synth.so: both compile synth_common.h with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "synth_common.h"

namespace {

constexpr int kBlock = 256;
thread_local std::string g_err;

int grid_for(uint64_t work_items, int num_cus, int blocks_per_cu) {
  uint64_t blocks = (work_items + kBlock - 1) / kBlock;
  uint64_t cap = (uint64_t)num_cus * (uint64_t)blocks_per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

__global__ __launch_bounds__(kBlock) void synth_rows_kernel(uint32_t seed, int im, int jm, int km, uint64_t row_begin,
                                                            uint64_t nrows, float* __restrict__ out) {
  // one thread per (row, feature): consecutive threads write consecutive floats
  const uint64_t total = nrows * OHX_NFEAT;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t plane = (uint64_t)im * (uint64_t)jm;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const uint64_t r = e / OHX_NFEAT;
    const int f = (int)(e - r * OHX_NFEAT);
    const uint64_t m = row_begin + r;
    const int k = (int)(m / plane);
    const uint64_t c = m - (uint64_t)k * plane;
    const int j = (int)(c / (uint64_t)im);
    const int i = (int)(c - (uint64_t)j * (uint64_t)im);
    out[e] = ohx_synth_feature(seed, f, i, j, k, im, jm, km);
  }
}

// feature >= 0: that feature as a MAPL field ((im,jm) or (im,jm,km); PL in Pa);
// feature == -1: TROPP (im,jm) in Pa
__global__ __launch_bounds__(kBlock) void synth_field_kernel(uint32_t seed, int feature, int im, int jm, int km,
                                                             float* __restrict__ out) {
  const uint64_t plane = (uint64_t)im * (uint64_t)jm;
  const bool two_d = feature < 0 || ohx_feature_is_2d(feature);
  const uint64_t total = two_d ? plane : plane * (uint64_t)km;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += stride) {
    const int k = (int)(m / plane);
    const uint64_t c = m - (uint64_t)k * plane;
    const int j = (int)(c / (uint64_t)im);
    const int i = (int)(c - (uint64_t)j * (uint64_t)im);
    float v;
    if (feature < 0) v = ohx_synth_tropp_pa(seed, i, j);
    else if (feature == OHX_F_PL) v = ohx_synth_pl_pa(seed, i, j, k, im, jm, km);
    else v = ohx_synth_feature(seed, feature, i, j, k, im, jm, km);
    out[m] = v;
  }
}

__global__ __launch_bounds__(kBlock) void inject_missing_kernel(float* __restrict__ rows, uint64_t count, uint32_t seed,
                                                                uint32_t rate_per_million, float missing) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const float qnan = __builtin_nanf("");
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
    const uint32_t h = ohx_hash4(seed, 0x99000000u, (uint32_t)(e & 0xFFFFFFFFu), (uint32_t)(e >> 32), 0u);
    if (h % 1000000u < rate_per_million) rows[e] = (h & 0x80000000u) ? missing : qnan;
  }
}


int fail(const std::string& m) {
  g_err = m;
  return -1;
}

int launched(const char* what) {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

const char* ohx_synth_gpu_last_error(void) { return g_err.c_str(); }

// rows [nrows][27] for row_begin..row_begin+nrows of an (im,jm,km) grid (device pointer)
int ohx_synth_rows_device(uint32_t seed, int im, int jm, int km, uint64_t row_begin, uint64_t nrows, float* d_out,
                          void* stream) {
  if (d_out == nullptr && nrows) return fail("ohx_synth_rows_device: d_out is NULL");
  if (im <= 0 || jm <= 0 || km <= 0) return fail("ohx_synth_rows_device: bad grid");
  if (row_begin + nrows > (uint64_t)im * (uint64_t)jm * (uint64_t)km) return fail("ohx_synth_rows_device: row range exceeds the grid");
  if (nrows == 0) return 0;
  hipLaunchKernelGGL(synth_rows_kernel, dim3(grid_for(nrows * OHX_NFEAT, 256, 16)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), seed, im, jm, km, row_begin, nrows, d_out);
  return launched("synth_rows_kernel");
}

// one MAPL field (feature 0..26 in reference order, PL in Pa; feature -1 = TROPP)
int ohx_synth_field_device(uint32_t seed, int feature, int im, int jm, int km, float* d_out, void* stream) {
  if (d_out == nullptr) return fail("ohx_synth_field_device: d_out is NULL");
  if (im <= 0 || jm <= 0 || km <= 0 || feature < -1 || feature >= 27) return fail("ohx_synth_field_device: bad argument");
  const uint64_t plane = (uint64_t)im * (uint64_t)jm;
  const bool two_d = feature < 0 || ohx_feature_is_2d(feature);
  const uint64_t total = two_d ? plane : plane * (uint64_t)km;
  hipLaunchKernelGGL(synth_field_kernel, dim3(grid_for(total, 256, 16)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     seed, feature, im, jm, km, d_out);
  return launched("synth_field_kernel");
}

int ohx_inject_missing_device(float* d_rows, uint64_t count, uint32_t seed, uint32_t rate_per_million, float missing,
                              void* stream) {
  if (d_rows == nullptr && count) return fail("ohx_inject_missing_device: d_rows is NULL");
  if (count == 0) return 0;
  hipLaunchKernelGGL(inject_missing_kernel, dim3(grid_for(count, 256, 16)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), d_rows, count, seed, rate_per_million, missing);
  return launched("inject_missing_kernel");
}

}  // extern "C"
#pragma GCC visibility pop
