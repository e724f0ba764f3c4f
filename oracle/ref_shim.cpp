/*
 * ref_shim.cpp — builds oracle/_ref/libref_runtime.so FROM THE REFERENCE'S OWN SOURCES,
 * compiled where they lie under /root/reference (nothing is copied into this repo).
 *
 * TEST INFRASTRUCTURE ONLY.  Used to (a) validate oracle/oracle.cpp's restatement and
 * (b) generate the golden vectors in tests/golden/ (oracle/gen_golden.py).
 *
 * Compiled unmodified from the reference tree:
 *   QueryEngine/MurmurHash.cpp (+MurmurHash1Inl.h, MurmurHash3Inl.h)
 *   QueryEngine/GroupByRuntime.cpp       key_hash, get_group_value*, get_group_value_fast*,
 *                                        hash_join_idx*
 *   QueryEngine/JoinHashTable/Runtime/JoinHashTableQueryRuntime.cpp
 *                                        baseline_hash_join_idx_{32,64}
 *   QueryEngine/JoinHashTable/Runtime/JoinHashImpl.h   get_hash_slot, fill_one_to_one_hashtable
 *   QueryEngine/DecodersImpl.h           fixed_width_int_decode, fixed_width_double_decode
 *
 * The only hand-supplied pieces are the three get_matching_group_value* functions that
 * GroupByRuntime.cpp expects its includer (RuntimeFunctions.cpp, which needs Boost) to have
 * defined, and a dynamic_watchdog stub.  They restate RuntimeFunctions.cpp:1953-2075.
 */
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "Shared/funcannotations.h"
#include "QueryEngine/BufferCompaction.h"
#include "QueryEngine/GpuRtConstants.h"

template <typename T>
inline T ref_get_empty_key();
template <>
inline int32_t ref_get_empty_key<int32_t>() { return EMPTY_KEY_32; }
template <>
inline int64_t ref_get_empty_key<int64_t>() { return EMPTY_KEY_64; }

// RuntimeFunctions.cpp:1953-1972
template <typename T>
static inline int64_t* ref_get_matching_group_value(int64_t* groups_buffer, const uint32_t h,
                                                    const T* key, const uint32_t key_count,
                                                    const uint32_t row_size_quad) {
  auto off = h * row_size_quad;
  auto row_ptr = reinterpret_cast<T*>(groups_buffer + off);
  if (*row_ptr == ref_get_empty_key<T>()) {
    memcpy(row_ptr, key, key_count * sizeof(T));
    auto row_ptr_i8 = reinterpret_cast<int8_t*>(row_ptr + key_count);
    return reinterpret_cast<int64_t*>(align_to_int64(row_ptr_i8));
  }
  if (memcmp(row_ptr, key, key_count * sizeof(T)) == 0) {
    auto row_ptr_i8 = reinterpret_cast<int8_t*>(row_ptr + key_count);
    return reinterpret_cast<int64_t*>(align_to_int64(row_ptr_i8));
  }
  return nullptr;
}

// RuntimeFunctions.cpp:1974-1992
extern "C" int64_t* get_matching_group_value(int64_t* groups_buffer, const uint32_t h,
                                             const int64_t* key, const uint32_t key_count,
                                             const uint32_t key_width,
                                             const uint32_t row_size_quad) {
  switch (key_width) {
    case 4:
      return ref_get_matching_group_value(groups_buffer, h,
                                          reinterpret_cast<const int32_t*>(key), key_count,
                                          row_size_quad);
    case 8:
      return ref_get_matching_group_value(groups_buffer, h, key, key_count, row_size_quad);
    default:;
  }
  return nullptr;
}

// RuntimeFunctions.cpp:1994-2047 (columnar variants; not exercised, needed to link)
extern "C" int32_t get_matching_group_value_columnar_slot(int64_t* groups_buffer,
                                                          const uint32_t entry_count,
                                                          const uint32_t h, const int64_t* key,
                                                          const uint32_t key_count,
                                                          const uint32_t key_width) {
  if (key_width != 8) return -1;
  auto off = h;
  if (groups_buffer[off] == EMPTY_KEY_64) {
    for (size_t i = 0; i < key_count; ++i) {
      groups_buffer[off] = key[i];
      off += entry_count;
    }
    return h;
  }
  off = h;
  for (size_t i = 0; i < key_count; ++i) {
    if (groups_buffer[off] != key[i]) return -1;
    off += entry_count;
  }
  return h;
}
extern "C" int64_t* get_matching_group_value_columnar(int64_t* groups_buffer, const uint32_t h,
                                                      const int64_t* key,
                                                      const uint32_t key_qw_count,
                                                      const size_t entry_count) {
  auto off = h;
  if (groups_buffer[off] == EMPTY_KEY_64) {
    for (size_t i = 0; i < key_qw_count; ++i) {
      groups_buffer[off] = key[i];
      off += entry_count;
    }
    return &groups_buffer[off];
  }
  off = h;
  for (size_t i = 0; i < key_qw_count; ++i) {
    if (groups_buffer[off] != key[i]) return nullptr;
    off += entry_count;
  }
  return &groups_buffer[off];
}
extern "C" bool dynamic_watchdog() { return false; }

#include "QueryEngine/GroupByRuntime.cpp"
#include "QueryEngine/JoinHashTable/Runtime/JoinHashTableQueryRuntime.cpp"
#include "QueryEngine/DecodersImpl.h"
