// TEST INFRASTRUCTURE (tests/hostsim): host stand-in for the one rocPRIM routine kernels_sort.hip calls.  "Device" memory is host
// memory here, so the stable LSD radix sort of (key, value) pairs is a std::stable_sort on bits [begin_bit, end_bit) of the keys.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <numeric>
#include <vector>

#include "hip/hip_runtime_api.h"

namespace rocprim {
template <typename K, typename V>
inline hipError_t radix_sort_pairs(void* temporary_storage, size_t& storage_size, const K* keys_in, K* keys_out, const V* values_in,
                                   V* values_out, size_t size, unsigned begin_bit, unsigned end_bit, hipStream_t) {
  if (!temporary_storage) {
    storage_size = 256;
    return hipSuccess;
  }
  const unsigned bits = end_bit - begin_bit;
  const K mask = bits >= sizeof(K) * 8 ? ~(K)0 : (K)((((K)1) << bits) - 1);
  std::vector<size_t> idx(size);
  std::iota(idx.begin(), idx.end(), (size_t)0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) {
    return ((keys_in[a] >> begin_bit) & mask) < ((keys_in[b] >> begin_bit) & mask);
  });
  for (size_t i = 0; i < size; ++i) {
    keys_out[i] = keys_in[idx[i]];
    values_out[i] = values_in[idx[i]];
  }
  return hipSuccess;
}
}  // namespace rocprim
