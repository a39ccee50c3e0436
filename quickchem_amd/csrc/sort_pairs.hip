// The one library primitive on the path: a device radix sort of (key, row number) pairs for the clustering pass
// (kernels.hip "rows nobody can describe").  Sorting 32-bit pairs is plain library work, like a plain GEMM: rocPRIM's
// onesweep radix sort (header-only, /opt/rocm/include/rocprim) does it; everything around it is this repository's own.
// In a translation unit of its own so that the templates are compiled once.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.hpp"

namespace ohx {

hipError_t sort_pairs_u32(void* temp, size_t* temp_bytes, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                          uint32_t* vals_b, uint64_t n, unsigned key_bits, hipStream_t stream, uint32_t** sorted_vals) {
  rocprim::double_buffer<uint32_t> k(keys_a, keys_b), v(vals_a, vals_b);
  size_t bytes = *temp_bytes;
  hipError_t e = rocprim::radix_sort_pairs(temp, bytes, k, v, (size_t)n, 0u, key_bits, stream);
  *temp_bytes = bytes;
  if (sorted_vals) *sorted_vals = v.current();
  return e;
}

}  // namespace ohx
