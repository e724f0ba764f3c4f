// dev_common.h — helpers shared by the HIP kernels and the host side of libmi355q.
//
// Semantics restate the reference (heavyai/heavydb) functions cited next to each helper;
// nothing here is derived from its CUDA runtime (cuda_mapd_rt.cu) — the kernels are written
// for gfx950 wave64 directly.
#pragma once

#include <stdint.h>

#include "../../include/mi355q.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MQ_HD __host__ __device__ __forceinline__
#define MQ_D __device__ __forceinline__
#else
#define MQ_HD inline
#define MQ_D inline
#endif

namespace mq {

constexpr int64_t kEmptyKey64 = INT64_MAX;  // GpuRtConstants.h:27
constexpr int32_t kEmptyKey32 = INT32_MAX;  // GpuRtConstants.h:28
constexpr double kNullDouble = 2.2250738585072014e-308;  // NULL_DOUBLE = DBL_MIN
constexpr int64_t kNullDoubleBits = 0x0010000000000000ll;

MQ_HD int type_width(int t) {
  return t == MI355Q_INT8 ? 1 : t == MI355Q_INT16 ? 2 : t == MI355Q_INT32 ? 4 : 8;
}
MQ_HD bool type_is_fp(int t) { return t == MI355Q_DOUBLE; }
// Shared/InlineNullValues.h:29-35
MQ_HD int64_t int_null_of(int t) {
  return t == MI355Q_INT8    ? (int64_t)INT8_MIN
         : t == MI355Q_INT16 ? (int64_t)INT16_MIN
         : t == MI355Q_INT32 ? (int64_t)INT32_MIN
                             : INT64_MIN;
}

MQ_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// MurmurHash3_x86_32 (MurmurHash3Inl.h:11-72) specialised to one 4-byte block, seed 0.
MQ_HD uint32_t murmur3_u32(uint32_t k) {
  uint32_t h1 = 0;
  uint32_t k1 = k * 0xcc9e2d51u;
  k1 = rotl32(k1, 15) * 0x1b873593u;
  h1 ^= k1;
  h1 = rotl32(h1, 13) * 5u + 0xe6546b64u;
  h1 ^= 4u;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}
// ... two 4-byte blocks (an int64 key, little endian), seed 0: key_hash(key, 1, 8)
// (GroupByRuntime.cpp:20-23).
MQ_HD uint32_t murmur3_u64(uint64_t key) {
  uint32_t h1 = 0;
  uint32_t k1 = (uint32_t)key * 0xcc9e2d51u;
  k1 = rotl32(k1, 15) * 0x1b873593u;
  h1 ^= k1;
  h1 = rotl32(h1, 13) * 5u + 0xe6546b64u;
  k1 = (uint32_t)(key >> 32) * 0xcc9e2d51u;
  k1 = rotl32(k1, 15) * 0x1b873593u;
  h1 ^= k1;
  h1 = rotl32(h1, 13) * 5u + 0xe6546b64u;
  h1 ^= 8u;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}
// MurmurHash1 (MurmurHash1Inl.h:22-62) of an int64 key, seed 0 — the keyed join hash
// (JoinHashTableQueryRuntime.cpp:63, HashJoinRuntime.cpp:514).
MQ_HD uint32_t murmur1_u64(uint64_t key) {
  const uint32_t m = 0xc6a4a793u;
  uint32_t h = 0u ^ (8u * m);
  h += (uint32_t)key;
  h *= m;
  h ^= h >> 16;
  h += (uint32_t)(key >> 32);
  h *= m;
  h ^= h >> 16;
  h *= m;
  h ^= h >> 10;
  h *= m;
  h ^= h >> 17;
  return h;
}

// BASELINE.md section 3 generator: u = splitmix64(seed ^ row * golden)
MQ_HD uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// fixed_width_int_decode (DecodersImpl.h:27-55): sign-extending fixed-width load.
MQ_HD int64_t decode_int(const int8_t* col, int type, int64_t pos) {
  switch (type) {
    case MI355Q_INT8: return *(const int8_t*)(col + pos);
    case MI355Q_INT16: return *(const int16_t*)(col + pos * 2);
    case MI355Q_INT32: return *(const int32_t*)(col + pos * 4);
    default: return *(const int64_t*)(col + pos * 8);
  }
}
// fixed_width_double_decode (DecodersImpl.h:121-128)
MQ_HD double decode_dbl(const int8_t* col, int64_t pos) { return *(const double*)(col + pos * 8); }

MQ_HD int64_t dbl_bits(double d) {
  union { double d; int64_t i; } u;
  u.d = d;
  return u.i;
}
MQ_HD double bits_dbl(int64_t i) {
  union { double d; int64_t i; } u;
  u.i = i;
  return u.d;
}

// ------------------------------------------------------------------ device-side plan
struct DevQual {
  int32_t col, op, type, nullable;
  int64_t ival;
  double fval;
};
struct DevTarget {
  int32_t agg, col, table, arg_type;
  int32_t arg_nullable, skip_null, slot, arg_fp;
};
struct DevPlan {
  int32_t n_cols, n_quals, n_targets, slot_count;
  DevQual quals[MI355Q_MAX_QUALS];
  DevTarget targets[MI355Q_MAX_TARGETS];
  int32_t desc_type, keyless, key_width, row_quad;
  int32_t key_quad, group_col, group_type, group_nullable;
  int64_t entry_count, min_val, max_val;
  int64_t init_vals[MI355Q_MAX_SLOTS];
  // join
  int32_t join_col, join_type, join_nullable, join_hash_type;
  const void* join_buf;
  const uint32_t* join_bitmap;  // perfect tables: 1 bit per slot = "slot holds a row id" (or null)
  int64_t join_min, join_max, join_entries;
  const int8_t* inner_cols[MI355Q_MAX_COLS];
};

}  // namespace mq
