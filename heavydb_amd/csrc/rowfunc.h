// rowfunc.h — the per-row logic of the generic kernel family: filter -> join probe -> group
// slot -> aggregate updates, written once and used with atomic slot updates on the device.
//
// tests/emu compiles this header for the host (plain g++) with its own single-threaded stand-ins for the
// atomics, so the row logic can be unit-tested without a GPU; the product library always uses the device atomics.
//
// Reference semantics restated (heavyai/heavydb):
//   filters      DEF_CMP_NULLABLE RuntimeFunctions.cpp:73-83, toBool LogicalIR.cpp:344-352
//   aggregates   agg_* RuntimeFunctions.cpp:362,1151-1169,1313-1431,1444-1470,1558-1584
//   perfect slot get_group_value_fast GroupByRuntime.cpp:208-241, keyless
//                RuntimeFunctions.cpp:2126-2133
//   baseline     get_group_value GroupByRuntime.cpp:25-48 (+ key_hash :20-23); insertion is
//                insert-or-find under CAS like any concurrent build, so slot POSITIONS are
//                order dependent and compared as sets (docs hash_joins.rst)
//   join probe   hash_join_idx GroupByRuntime.cpp:287-318; baseline_hash_join_idx_64
//                JoinHashTableQueryRuntime.cpp:40-94
#pragma once

#include "dev_common.h"

namespace mq {

// The slot-update primitives below are written over this macro set (device atomics).  tests/emu compiles this header for
// the host with single-threaded stand-ins it defines BEFORE including it (tests/emu/emu_atomics.h sets MQ_SLOT_ATOMICS);
// nothing of that test double lives in the product sources.
#ifndef MQ_SLOT_ATOMICS
#define MQ_SLOT_ATOMICS 1
#define MQ_CAS64(p, e, d) atomicCAS((unsigned long long*)(p), (unsigned long long)(e), (unsigned long long)(d))
#define MQ_CAS32(p, e, d) atomicCAS((unsigned int*)(p), (unsigned int)(e), (unsigned int)(d))
#define MQ_ADD64(p, v) atomicAdd((unsigned long long*)(p), (unsigned long long)(v))
#define MQ_ADD32(p, v) atomicAdd((unsigned int*)(p), (unsigned int)(v))
#define MQ_ADDF64(p, v) atomicAdd((double*)(p), (double)(v))
#define MQ_ADDF32(p, v) atomicAdd((float*)(p), (float)(v))
#define MQ_MIN64(p, v) atomicMin((long long*)(p), (long long)(v))
#define MQ_MAX64(p, v) atomicMax((long long*)(p), (long long)(v))
#define MQ_LOAD64(p) __hip_atomic_load((int64_t*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MQ_STORE64(p, v) __hip_atomic_store((int64_t*)(p), (int64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MQ_FENCE() __threadfence()
// Order the agent-scope atomic (write-through) stores of a key's tail before the store that
// publishes its first component: wait until they are acknowledged.  A full release fence would
// also write back the L2 (buffer_wbl2) for plain stores this protocol never issues — that costs
// microseconds per inserted key.
#define MQ_PUBLISH_ORDER() __builtin_amdgcn_s_waitcnt(0)
#define MQ_LOAD32(p) __hip_atomic_load((int32_t*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MQ_STORE32(p, v) __hip_atomic_store((int32_t*)(p), (int32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MQ_FN __device__ __forceinline__
#endif

// ---------------------------------------------------------------- slot update primitives
// All take the slot as int64_t* (8-byte padded slots).  *_skip variants implement the
// reference's "_skip_val" semantics: the slot starts at the NULL sentinel (== skip) and the
// first non-NULL value overwrites it.
template <bool A>
MQ_FN void a_count(int64_t* s) {
  if (A) MQ_ADD64(s, 1ull);
  else *s += 1;
}
template <bool A>
MQ_FN void a_sum_i64(int64_t* s, int64_t v) {
  if (A) MQ_ADD64(s, v);
  else *s += v;
}
template <bool A>
MQ_FN void a_sum_i64_skip(int64_t* s, int64_t v, int64_t skip) {
  if (v == skip) return;
  if (!A) {
    *s = (*s == skip) ? v : *s + v;
    return;
  }
  int64_t old = MQ_LOAD64(s);
  for (;;) {
    const int64_t nv = (old == skip) ? v : old + v;
    const int64_t seen = (int64_t)MQ_CAS64(s, old, nv);
    if (seen == old) return;
    old = seen;
  }
}
template <bool A>
MQ_FN void a_min_i64(int64_t* s, int64_t v) {
  if (A) MQ_MIN64(s, v);
  else *s = *s < v ? *s : v;
}
template <bool A>
MQ_FN void a_max_i64(int64_t* s, int64_t v) {
  if (A) MQ_MAX64(s, v);
  else *s = *s > v ? *s : v;
}
template <bool A>
MQ_FN void a_min_i64_skip(int64_t* s, int64_t v, int64_t skip) {
  if (v == skip) return;
  if (!A) {
    *s = (*s == skip) ? v : (*s < v ? *s : v);
    return;
  }
  int64_t old = MQ_LOAD64(s);
  for (;;) {
    const int64_t nv = (old == skip) ? v : (old < v ? old : v);
    if (nv == old) return;
    const int64_t seen = (int64_t)MQ_CAS64(s, old, nv);
    if (seen == old) return;
    old = seen;
  }
}
template <bool A>
MQ_FN void a_max_i64_skip(int64_t* s, int64_t v, int64_t skip) {
  if (v == skip) return;
  if (!A) {
    *s = (*s == skip) ? v : (*s > v ? *s : v);
    return;
  }
  int64_t old = MQ_LOAD64(s);
  for (;;) {
    const int64_t nv = (old == skip) ? v : (old > v ? old : v);
    if (nv == old) return;
    const int64_t seen = (int64_t)MQ_CAS64(s, old, nv);
    if (seen == old) return;
    old = seen;
  }
}
template <bool A>
MQ_FN void a_sum_f64(int64_t* s, double v) {
  if (A) MQ_ADDF64(s, v);
  else *s = dbl_bits(bits_dbl(*s) + v);
}
template <bool A>
MQ_FN void a_sum_f64_skip(int64_t* s, double v, double skip) {
  if (v == skip) return;
  const int64_t skip_bits = dbl_bits(skip);
  if (!A) {
    *s = (*s == skip_bits) ? dbl_bits(v) : dbl_bits(bits_dbl(*s) + v);
    return;
  }
  int64_t old = MQ_LOAD64(s);
  for (;;) {
    const int64_t nv = (old == skip_bits) ? dbl_bits(v) : dbl_bits(bits_dbl(old) + v);
    const int64_t seen = (int64_t)MQ_CAS64(s, old, nv);
    if (seen == old) return;
    old = seen;
  }
}
// MIN/MAX on doubles: CAS loops (std::min/max semantics of agg_min_double/agg_max_double)
template <bool A, bool IS_MAX, bool SKIP>
MQ_FN void a_minmax_f64(int64_t* s, double v, double skip) {
  if (SKIP && v == skip) return;
  const int64_t skip_bits = dbl_bits(skip);
  int64_t old = A ? MQ_LOAD64(s) : *s;
  for (;;) {
    int64_t nv;
    if (SKIP && old == skip_bits) {
      nv = dbl_bits(v);
    } else {
      const double o = bits_dbl(old);
      const double r = IS_MAX ? (o < v ? v : o) : (v < o ? v : o);
      nv = dbl_bits(r);
    }
    if (nv == old) return;
    if (!A) {
      *s = nv;
      return;
    }
    const int64_t seen = (int64_t)MQ_CAS64(s, old, nv);
    if (seen == old) return;
    old = seen;
  }
}

// FLOAT arguments: single precision on the low 4 bytes of the 8-byte slot (agg_sum_float,
// agg_min_float, agg_max_float and their _skip_val forms, RuntimeFunctions.cpp:1496-1520,
// :1586-1594); the upper 4 bytes keep what the row initialisation put there.
template <bool A>
MQ_FN void a_sum_f32(int64_t* s, float v) {
  if (A) MQ_ADDF32(s, v);
  else *(float*)s += v;
}
template <bool A>
MQ_FN void a_sum_f32_skip(int64_t* s, float v, int32_t skip_bits) {
  if (v == bits_flt(skip_bits)) return;
  int32_t* s32 = (int32_t*)s;
  if (!A) {
    *s32 = (*s32 == skip_bits) ? flt_bits(v) : flt_bits(bits_flt(*s32) + v);
    return;
  }
  int32_t old = MQ_LOAD32(s32);
  for (;;) {
    const int32_t nv = (old == skip_bits) ? flt_bits(v) : flt_bits(bits_flt(old) + v);
    const int32_t seen = (int32_t)MQ_CAS32(s32, old, nv);
    if (seen == old) return;
    old = seen;
  }
}
template <bool A, bool IS_MAX, bool SKIP>
MQ_FN void a_minmax_f32(int64_t* s, float v, int32_t skip_bits) {
  if (SKIP && v == bits_flt(skip_bits)) return;
  int32_t* s32 = (int32_t*)s;
  int32_t old = A ? MQ_LOAD32(s32) : *s32;
  for (;;) {
    int32_t nv;
    if (SKIP && old == skip_bits) {
      nv = flt_bits(v);
    } else {
      const float o = bits_flt(old);
      nv = flt_bits(IS_MAX ? (o < v ? v : o) : (v < o ? v : o));
    }
    if (nv == old) return;
    if (!A) {
      *s32 = nv;
      return;
    }
    const int32_t seen = (int32_t)MQ_CAS32(s32, old, nv);
    if (seen == old) return;
    old = seen;
  }
}

// ---------------------------------------------------------------- filter
// The value of a column element as one 64-bit pattern: integers decoded and sign-extended, DOUBLE its bits, FLOAT its
// bits in the low word — what the projected-expression evaluator (expr.h) and qual_on_value work on.
MQ_FN int64_t col_value_bits(const int8_t* col, int code, int64_t pos) {
  const int st = tc_storage(code);
  if (st == MI355Q_FLOAT) return (int64_t)*(const uint32_t*)(col + pos * 4);
  if (st == MI355Q_DOUBLE) return *(const int64_t*)(col + pos * 8);
  return decode_int(col, code, pos);
}
// `col <op> literal` on a value already fetched (col_value_bits)
MQ_FN bool qual_on_value(const DevQual& q, int64_t bits) {
  const int st = tc_storage(q.type);
  if (q.op == MI355Q_IS_NULL || q.op == MI355Q_IS_NOT_NULL) {
    // codegenIsNull (LogicalIR.cpp:381-432): false on a NOT NULL type, else value == inline NULL
    bool is_null = false;
    if (q.nullable) {
      if (st == MI355Q_FLOAT) is_null = bits_flt((int32_t)(uint32_t)bits) == kNullFloat;
      else if (st == MI355Q_DOUBLE) is_null = bits_dbl(bits) == kNullDouble;
      else is_null = bits == int_null_of(q.type);
    }
    return q.op == MI355Q_IS_NULL ? is_null : !is_null;
  }
  if (st == MI355Q_FLOAT) {  // the literal is folded to the column's type: compared in float
    const float v = bits_flt((int32_t)(uint32_t)bits);
    const float lit = (float)q.fval;
    if (q.nullable && v == kNullFloat) return false;
    switch (q.op) {
      case MI355Q_EQ: return v == lit;
      case MI355Q_NE: return v != lit;
      case MI355Q_LT: return v < lit;
      case MI355Q_GT: return v > lit;
      case MI355Q_LE: return v <= lit;
      default: return v >= lit;
    }
  }
  if (st == MI355Q_DOUBLE) {
    const double v = bits_dbl(bits);
    if (q.nullable && v == kNullDouble) return false;
    switch (q.op) {
      case MI355Q_EQ: return v == q.fval;
      case MI355Q_NE: return v != q.fval;
      case MI355Q_LT: return v < q.fval;
      case MI355Q_GT: return v > q.fval;
      case MI355Q_LE: return v <= q.fval;
      default: return v >= q.fval;
    }
  }
  const int64_t v = bits;
  if (q.nullable && v == int_null_of(q.type)) return false;
  switch (q.op) {
    case MI355Q_EQ: return v == q.ival;
    case MI355Q_NE: return v != q.ival;
    case MI355Q_LT: return v < q.ival;
    case MI355Q_GT: return v > q.ival;
    case MI355Q_LE: return v <= q.ival;
    default: return v >= q.ival;
  }
}
MQ_FN bool eval_qual(const DevQual& q, const int8_t* col, int64_t pos) {
  return qual_on_value(q, col_value_bits(col, q.type, pos));
}

// ---------------------------------------------------------------- join probe
// Keyed tables: components of `width` bytes; one-to-one slots are (key..., payload), one-to-many
// slots are the key alone.  Returns the slot index, -2 when probing stops at an empty slot
// (kNotPresent), -1 when the table wrapped (kNoMatch)
// (baseline_hash_join_idx_impl / get_composite_key_index_impl,
// JoinHashTableQueryRuntime.cpp:35-94,140-163).
MQ_FN int64_t keyed_slot_of(const void* tab, uint32_t entries, int n_keys, int width, int stride,
                            const int64_t* keys) {
  if (!entries) return -1;
  uint32_t words[2 * MI355Q_MAX_GROUP_COLS];
  const int n_words = pack_join_key(keys, n_keys, width, words);
  const uint32_t h = murmur1_words(words, n_words) % entries;
  uint32_t hp = h;
  do {
    bool same = true, empty;
    if (width == 4) {
      const int32_t* e = (const int32_t*)tab + (size_t)hp * stride;
      for (int i = 0; i < n_keys; ++i) same = same && e[i] == (int32_t)keys[i];
      empty = e[0] == kEmptyKey32;
    } else {
      const int64_t* e = (const int64_t*)tab + (size_t)hp * stride;
      for (int i = 0; i < n_keys; ++i) same = same && e[i] == keys[i];
      empty = e[0] == kEmptyKey64;
    }
    if (same) return hp;
    if (empty) return -2;
    hp = hp + 1 == entries ? 0 : hp + 1;
  } while (hp != h);
  return -1;
}

// The set of inner rows matching one outer row: `count` row ids at `ids`, or (one-to-one
// tables) the single row id `single`.  count == 0: no match.
struct JoinMatch {
  const int32_t* ids;
  int64_t single;
  int32_t count;
};

MQ_FN JoinMatch join_lookup(const DevPlan& p, const int64_t* keys) {
  JoinMatch m{nullptr, -1, 0};
  const int64_t n = p.join_entries;
  switch (p.join_hash_type) {
    case 0: {  // hash_join_idx (GroupByRuntime.cpp:287-297)
      if (keys[0] >= p.join_min && keys[0] <= p.join_max) {
        m.single = ((const int32_t*)p.join_buf)[keys[0] - p.join_min];
        m.count = m.single >= 0;
      }
      break;
    }
    case 1: {
      const int stride = p.join_n_keys + 1;
      const int64_t slot = keyed_slot_of(p.join_buf, (uint32_t)n, p.join_n_keys, p.join_width, stride, keys);
      if (slot >= 0) {
        m.single = p.join_width == 4 ? (int64_t)((const int32_t*)p.join_buf)[slot * stride + p.join_n_keys]
                                     : ((const int64_t*)p.join_buf)[slot * stride + p.join_n_keys];
        m.count = m.single >= 0;
      }
      break;
    }
    case 2: {  // HashJoin::codegenMatchingSet: offsets | counts | payloads, all int32
      if (keys[0] >= p.join_min && keys[0] <= p.join_max) {
        const int32_t* offsets = (const int32_t*)p.join_buf;
        const int32_t off = offsets[keys[0] - p.join_min];
        if (off >= 0) {
          m.count = offsets[n + (keys[0] - p.join_min)];
          m.ids = offsets + 2 * n + off;
        }
      }
      break;
    }
    default: {  // keys | offsets | counts | payloads
      const int64_t slot = keyed_slot_of(p.join_buf, (uint32_t)n, p.join_n_keys, p.join_width, p.join_n_keys, keys);
      if (slot >= 0) {
        const int32_t* offsets =
            (const int32_t*)((const int8_t*)p.join_buf + (size_t)n * p.join_n_keys * p.join_width);
        const int32_t off = offsets[slot];
        if (off >= 0) {
          m.count = offsets[n + slot];
          m.ids = offsets + 2 * n + off;
        }
      }
    }
  }
  return m;
}

// ---------------------------------------------------------------- group slot
// Returns pointer to the first aggregate slot of the row's group, or nullptr when the
// baseline table is full (caller raises the out-of-slots error).
MQ_FN int64_t* baseline_find_or_insert(int64_t* buf, uint32_t entry_count, int row_quad,
                                       int key_width, int64_t key) {
  if (key_width == 4) {
    const int32_t k32 = (int32_t)key;
    const uint32_t h = murmur3_u32((uint32_t)k32) % entry_count;
    uint32_t hp = h;
    do {
      int64_t* row = buf + (size_t)hp * row_quad;
      const uint32_t old = MQ_CAS32(row, (uint32_t)kEmptyKey32, (uint32_t)k32);
      if (old == (uint32_t)kEmptyKey32 || old == (uint32_t)k32) return row + 1;
      hp = hp + 1 == entry_count ? 0 : hp + 1;
    } while (hp != h);
    return nullptr;
  }
  const uint32_t h = murmur3_u64((uint64_t)key) % entry_count;
  uint32_t hp = h;
  do {
    int64_t* row = buf + (size_t)hp * row_quad;
    const int64_t old = (int64_t)MQ_CAS64(row, kEmptyKey64, key);
    if (old == kEmptyKey64 || old == key) return row + 1;
    hp = hp + 1 == entry_count ? 0 : hp + 1;
  } while (hp != h);
  return nullptr;
}

// Multi-column keys (key_count components of key_width bytes each, packed at the row start).
// The reference's GPU build claims the first component with a CAS and lets late arrivals spin
// until the last component is written (cuda_mapd_rt.cu get_matching_group_value); here the
// first component doubles as a write lock: EMPTY -> LOCKED (EMPTY - 1, a value the reference
// also keeps out of int32 key ranges: is_valid_int32_range) -> the real value, published with
// a release fence after the other components are in place.  One loop, no early return: the
// winner publishes inside the iteration it won in, so lanes of its own wave that lost the CAS
// see the key on their next trip (no intra-wave deadlock).  Returns the first slot, nullptr
// when the table is full, and sets *bad when the key itself is unrepresentable.
constexpr int64_t kLockedKey64 = INT64_MAX - 1;
constexpr int32_t kLockedKey32 = INT32_MAX - 1;
constexpr int kMaxLockSpins = 1 << 24;

MQ_FN int64_t* baseline_find_or_insert_multi(int64_t* buf, uint32_t entry_count, int row_quad,
                                             int key_width, int key_count, const int64_t* keys,
                                             bool* bad) {
  const int key_quad = (key_count * key_width + 7) >> 3;
  uint32_t words[2 * MI355Q_MAX_GROUP_COLS];
  int n_words;
  if (key_width == 4) {
    for (int i = 0; i < key_count; ++i) words[i] = (uint32_t)(int32_t)keys[i];
    n_words = key_count;
    if ((int32_t)keys[0] == kLockedKey32 || (int32_t)keys[0] == kEmptyKey32) *bad = true;
  } else {
    for (int i = 0; i < key_count; ++i) {
      words[2 * i] = (uint32_t)(uint64_t)keys[i];
      words[2 * i + 1] = (uint32_t)((uint64_t)keys[i] >> 32);
    }
    n_words = 2 * key_count;
    if (keys[0] == kLockedKey64 || keys[0] == kEmptyKey64) *bad = true;
  }
  if (*bad) return nullptr;
  const uint32_t h = murmur3_words(words, n_words) % entry_count;
  uint32_t hp = h;
  int64_t* found = nullptr;
  bool done = false;
  int spins = 0;
  while (!done) {
    int64_t* row = buf + (size_t)hp * row_quad;
    bool advance = false;
    if (key_width == 4) {
      int32_t* r32 = (int32_t*)row;
      const int32_t old = (int32_t)MQ_CAS32(r32, (uint32_t)kEmptyKey32, (uint32_t)kLockedKey32);
      if (old == kEmptyKey32) {
        for (int i = 1; i < key_count; ++i) MQ_STORE32(r32 + i, (int32_t)keys[i]);
        MQ_PUBLISH_ORDER();
        MQ_STORE32(r32, (int32_t)keys[0]);
        found = row + key_quad;
        done = true;
      } else if (old == kLockedKey32) {
        if (++spins > kMaxLockSpins) {
          *bad = true;
          done = true;
        }
      } else if (old == (int32_t)keys[0]) {
        // the loads below depend on the CAS result and are agent-scope atomic loads, so they
        // observe what the winner published before it released the first component
        bool same = true;
        for (int i = 1; i < key_count; ++i) same = same && MQ_LOAD32(r32 + i) == (int32_t)keys[i];
        if (same) {
          found = row + key_quad;
          done = true;
        } else {
          advance = true;
        }
      } else {
        advance = true;
      }
    } else {
      const int64_t old = (int64_t)MQ_CAS64(row, kEmptyKey64, kLockedKey64);
      if (old == kEmptyKey64) {
        for (int i = 1; i < key_count; ++i) MQ_STORE64(row + i, keys[i]);
        MQ_PUBLISH_ORDER();
        MQ_STORE64(row, keys[0]);
        found = row + key_quad;
        done = true;
      } else if (old == kLockedKey64) {
        if (++spins > kMaxLockSpins) {
          *bad = true;
          done = true;
        }
      } else if (old == keys[0]) {
        bool same = true;
        for (int i = 1; i < key_count; ++i) same = same && MQ_LOAD64(row + i) == keys[i];
        if (same) {
          found = row + key_quad;
          done = true;
        } else {
          advance = true;
        }
      } else {
        advance = true;
      }
    }
    if (advance) {
      hp = hp + 1 == entry_count ? 0 : hp + 1;
      if (hp == h) done = true;  // wrapped: table full
    }
  }
  return found;
}

// Bulk insertion of keys that are known to be pairwise DISTINCT and absent from the table (a
// finished table being re-emitted): a row is claimed by CAS on its first component and never
// searched for, so there is nothing to publish — an occupied row is simply skipped.
MQ_FN int64_t* baseline_insert_distinct_multi(int64_t* buf, uint32_t entry_count, int row_quad,
                                              int key_width, int key_count, const int64_t* keys) {
  const int key_quad = (key_count * key_width + 7) >> 3;
  uint32_t words[2 * MI355Q_MAX_GROUP_COLS];
  const int n_words = pack_join_key(keys, key_count, key_width, words);
  const uint32_t h = murmur3_words(words, n_words) % entry_count;
  uint32_t hp = h;
  do {
    int64_t* row = buf + (size_t)hp * row_quad;
    if (key_width == 4) {
      int32_t* r32 = (int32_t*)row;
      if ((int32_t)MQ_CAS32(r32, (uint32_t)kEmptyKey32, (uint32_t)(int32_t)keys[0]) == kEmptyKey32) {
        for (int i = 1; i < key_count; ++i) r32[i] = (int32_t)keys[i];
        return row + key_quad;
      }
    } else {
      if ((int64_t)MQ_CAS64(row, kEmptyKey64, keys[0]) == kEmptyKey64) {
        for (int i = 1; i < key_count; ++i) row[i] = keys[i];
        return row + key_quad;
      }
    }
    hp = hp + 1 == entry_count ? 0 : hp + 1;
  } while (hp != h);
  return nullptr;
}

// component i of the key stored at the start of `row`
MQ_FN int64_t row_key_component(const int64_t* row, int key_width, int i) {
  return key_width == 4 ? (int64_t)((const int32_t*)row)[i] : row[i];
}

// ---------------------------------------------------------------- one target, one row
template <bool A>
MQ_FN void apply_target(const DevTarget& t, int64_t* slots, const int8_t* const* cols,
                        int64_t pos, const int8_t* const* inner_cols, int64_t inner_pos,
                        const int64_t* key_vals) {
  if (t.agg == MI355Q_PROJECT_KEY) {
    if (t.slot >= 0) {  // agg_id
      if (A) MQ_STORE64(slots + t.slot, key_vals[t.key_idx]);
      else slots[t.slot] = key_vals[t.key_idx];
    }
    return;
  }
  int64_t* s = slots + t.slot;
  int agg = t.agg;
  if (agg == MI355Q_COUNT_IF || agg == MI355Q_SUM_IF) {
    // the condition is TRUE only when it evaluates to 1: a NULL operand is not TRUE
    // (agg_count_if_skip_val, codegenConditionalAggregateCondValSelector)
    if (!eval_qual(t.cond, cols[t.cond.col], pos)) return;
    if (agg == MI355Q_COUNT_IF) {
      a_count<A>(s);
      return;
    }
    agg = MI355Q_SUM;
  }
  if (t.col < 0) {
    a_count<A>(s);
    return;
  }
  // LEFT join, no match: every inner column is NULL (codegenOuterJoinNullPlaceholder,
  // ColumnIR.cpp) and inner columns are nullable under an outer join, so each aggregate's
  // _skip_val form leaves its slot alone
  if (t.table && inner_pos < 0) return;
  const int8_t* col = t.table ? inner_cols[t.col] : cols[t.col];
  const int64_t p = t.table ? inner_pos : pos;
  if (t.arg_f32) {
    const float v = decode_flt(col, p);
    switch (agg) {
      case MI355Q_COUNT:
        if (!t.skip_null || v != kNullFloat) a_count<A>(s);
        break;
      case MI355Q_SUM:
        if (t.skip_null) a_sum_f32_skip<A>(s, v, kNullFloatBits);
        else a_sum_f32<A>(s, v);
        break;
      case MI355Q_AVG:
        if (t.skip_null) {
          if (v != kNullFloat) {
            a_sum_f32_skip<A>(s, v, kNullFloatBits);
            a_count<A>(s + 1);
          }
        } else {
          a_sum_f32<A>(s, v);
          a_count<A>(s + 1);
        }
        break;
      case MI355Q_MIN:
        if (t.skip_null) a_minmax_f32<A, false, true>(s, v, kNullFloatBits);
        else a_minmax_f32<A, false, false>(s, v, 0);
        break;
      default:
        if (t.skip_null) a_minmax_f32<A, true, true>(s, v, kNullFloatBits);
        else a_minmax_f32<A, true, false>(s, v, 0);
    }
    return;
  }
  if (t.arg_fp) {
    const double v = decode_dbl(col, p);
    switch (agg) {
      case MI355Q_COUNT:
        if (!t.skip_null || v != kNullDouble) a_count<A>(s);
        break;
      case MI355Q_SUM:
        if (t.skip_null) a_sum_f64_skip<A>(s, v, kNullDouble);
        else a_sum_f64<A>(s, v);
        break;
      case MI355Q_AVG:
        if (t.skip_null) {
          if (v != kNullDouble) {
            a_sum_f64_skip<A>(s, v, kNullDouble);
            a_count<A>(s + 1);
          }
        } else {
          a_sum_f64<A>(s, v);
          a_count<A>(s + 1);
        }
        break;
      case MI355Q_MIN:
        if (t.skip_null) a_minmax_f64<A, false, true>(s, v, kNullDouble);
        else a_minmax_f64<A, false, false>(s, v, 0.0);
        break;
      default:
        if (t.skip_null) a_minmax_f64<A, true, true>(s, v, kNullDouble);
        else a_minmax_f64<A, true, false>(s, v, 0.0);
    }
    return;
  }
  const int64_t raw = decode_int(col, t.arg_type, p);
  const int64_t null_t = int_null_of(t.arg_type);
  switch (agg) {
    case MI355Q_COUNT:
      if (!t.skip_null || raw != null_t) a_count<A>(s);
      break;
    case MI355Q_SUM:
    case MI355Q_AVG:
      if (t.skip_null) {
        if (raw != null_t) {
          // slot starts at NULL_BIGINT for SUM (skip value) and at 0 for AVG.sum
          a_sum_i64_skip<A>(s, raw, INT64_MIN);
          if (agg == MI355Q_AVG) a_count<A>(s + 1);
        }
      } else {
        a_sum_i64<A>(s, raw);
        if (agg == MI355Q_AVG) a_count<A>(s + 1);
      }
      break;
    case MI355Q_MIN:
      if (t.skip_null) a_min_i64_skip<A>(s, raw, null_t);
      else a_min_i64<A>(s, raw);
      break;
    default:
      if (t.skip_null) a_max_i64_skip<A>(s, raw, null_t);
      else a_max_i64<A>(s, raw);
  }
}

// ---------------------------------------------------------------- reduce one target
// this_slots (op)= that_slots with the target's aggregate, init value as the skip value
// (ResultSetStorage::reduceOneSlot, ResultSetReduction.cpp:1496-1640).
template <bool A>
MQ_FN void reduce_target(const DevTarget& t, const int64_t* init_vals, int64_t* this_slots,
                         const int64_t* that_slots) {
  if (t.slot < 0) return;
  int64_t* a = this_slots + t.slot;
  const int64_t b = that_slots[t.slot];
  const int64_t init = init_vals[t.slot];
  const bool fp = t.arg_fp && t.agg != MI355Q_COUNT;
  const int agg = t.agg == MI355Q_COUNT_IF ? MI355Q_COUNT : t.agg == MI355Q_SUM_IF ? MI355Q_SUM : t.agg;
  if (t.arg_f32 && agg != MI355Q_COUNT) {  // reduceOneSlot with chosen_bytes = sizeof(float)
    const float bf = bits_flt((int32_t)b);
    const int32_t init32 = (int32_t)init;
    switch (agg) {
      case MI355Q_AVG:
        a_sum_i64<A>(a + 1, that_slots[t.slot + 1]);
        // fallthrough
      case MI355Q_SUM:
        if (t.skip_null) a_sum_f32_skip<A>(a, bf, init32);
        else a_sum_f32<A>(a, bf);
        break;
      case MI355Q_MIN:
        if (t.skip_null) a_minmax_f32<A, false, true>(a, bf, init32);
        else a_minmax_f32<A, false, false>(a, bf, 0);
        break;
      default:
        if (t.skip_null) a_minmax_f32<A, true, true>(a, bf, init32);
        else a_minmax_f32<A, true, false>(a, bf, 0);
    }
    return;
  }
  switch (agg) {
    case MI355Q_COUNT:
      a_sum_i64<A>(a, b);
      break;
    case MI355Q_AVG:
      a_sum_i64<A>(a + 1, that_slots[t.slot + 1]);
      // fallthrough
    case MI355Q_SUM:
      if (t.skip_null) {
        if (fp) a_sum_f64_skip<A>(a, bits_dbl(b), bits_dbl(init));
        else a_sum_i64_skip<A>(a, b, init);
      } else {
        if (fp) a_sum_f64<A>(a, bits_dbl(b));
        else a_sum_i64<A>(a, b);
      }
      break;
    case MI355Q_MIN:
      if (fp) {
        if (t.skip_null) a_minmax_f64<A, false, true>(a, bits_dbl(b), bits_dbl(init));
        else a_minmax_f64<A, false, false>(a, bits_dbl(b), 0.0);
      } else {
        if (t.skip_null) a_min_i64_skip<A>(a, b, init);
        else a_min_i64<A>(a, b);
      }
      break;
    case MI355Q_MAX:
      if (fp) {
        if (t.skip_null) a_minmax_f64<A, true, true>(a, bits_dbl(b), bits_dbl(init));
        else a_minmax_f64<A, true, false>(a, bits_dbl(b), 0.0);
      } else {
        if (t.skip_null) a_max_i64_skip<A>(a, b, init);
        else a_max_i64<A>(a, b);
      }
      break;
    default:
      if (b != init) {
        if (A) MQ_STORE64(a, b);
        else *a = b;
      }
  }
}

// isEmptyEntry (ResultSetIteration.cpp:2457-2492)
MQ_FN bool is_empty_row(const DevPlan& p, const int64_t* row, int idx_target_as_key) {
  if (p.desc_type == MI355Q_NON_GROUPED_AGGREGATE) return false;
  if (p.keyless && p.slot_width == 4)
    return ((const int32_t*)row)[idx_target_as_key] == (int32_t)p.init_vals[idx_target_as_key];
  if (p.keyless) return row[idx_target_as_key] == p.init_vals[idx_target_as_key];
  if (p.key_width == 4) return *(const int32_t*)row == kEmptyKey32;
  return row[0] == kEmptyKey64;
}

// ---------------------------------------------------------------- 4-byte slots
// The compact layouts (slot_width 4) only ever hold COUNT(*) and projections of keys of at most
// 4 bytes (pick_target_compact_width).  The step itself runs on the 8-byte layout of the same
// plan; narrow_row turns a finished row into its compact image (same entry, same key bytes).
MQ_FN void narrow_row(const int64_t* wide, int key_quad, int slot_count, int narrow_row_quad, int64_t* dst) {
  for (int k = 0; k < key_quad; ++k) dst[k] = wide[k];
  int32_t* s32 = (int32_t*)(dst + key_quad);
  const int n32 = (narrow_row_quad - key_quad) * 2;
  for (int s = 0; s < n32; ++s) s32[s] = s < slot_count ? (int32_t)wide[key_quad + s] : 0;
}
// ---------------------------------------------------------------- columnar output
// A columnar descriptor (output_columnar_) stores entry e of group column g at
// g * 8 * entry_count + e * 8 (getPrependedGroupColOffInBytes, QueryMemoryDescriptor.cpp:962-975;
// no group columns when keyless) and of slot s at keys + s * align8(slot_width * entry_count) +
// e * slot_width (getColOffInBytes :906-929).  The step runs on the row-wise form of the same
// decisions; these move one finished entry between the two forms (same entry index, same values).
MQ_FN void entry_to_columns(const ColLayout& L, const int64_t* row, int8_t* cols, int64_t e) {
  for (int k = 0; k < L.key_quads; ++k) ((int64_t*)(cols + (int64_t)k * 8 * L.entry_count))[e] = row[k];
  int8_t* sc = cols + (int64_t)L.key_quads * 8 * L.entry_count;
  if (L.slot_width == 8) {
    for (int s = 0; s < L.slot_count; ++s) ((int64_t*)(sc + (int64_t)s * L.slot_col_bytes))[e] = row[L.key_quads + s];
  } else {
    const int32_t* s32 = (const int32_t*)(row + L.key_quads);
    for (int s = 0; s < L.slot_count; ++s) ((int32_t*)(sc + (int64_t)s * L.slot_col_bytes))[e] = s32[s];
  }
}
MQ_FN void entry_from_columns(const ColLayout& L, const int8_t* cols, int64_t e, int64_t* row) {
  for (int k = 0; k < L.key_quads; ++k) row[k] = ((const int64_t*)(cols + (int64_t)k * 8 * L.entry_count))[e];
  const int8_t* sc = cols + (int64_t)L.key_quads * 8 * L.entry_count;
  if (L.slot_width == 8) {
    for (int s = 0; s < L.slot_count; ++s) row[L.key_quads + s] = ((const int64_t*)(sc + (int64_t)s * L.slot_col_bytes))[e];
  } else {
    int32_t* s32 = (int32_t*)(row + L.key_quads);
    const int n32 = (L.row_quad - L.key_quads) * 2;  // an odd slot count leaves one zero padding word
    for (int s = 0; s < n32; ++s)
      s32[s] = s < L.slot_count ? ((const int32_t*)(sc + (int64_t)s * L.slot_col_bytes))[e] : 0;
  }
}

// this (op)= that for one target of a compact row: agg_sum on the 32-bit COUNT
// (AGGREGATE_ONE_COUNT with chosen_bytes 4), projections copied when set
template <bool A>
MQ_FN void reduce_target_compact(const DevTarget& t, const int64_t* init_vals, int32_t* this_slots,
                                 const int32_t* that_slots) {
  if (t.slot < 0) return;
  const int32_t b = that_slots[t.slot];
  if (t.agg == MI355Q_PROJECT_KEY) {
    if (b != (int32_t)init_vals[t.slot]) {
      if (A) MQ_STORE32(this_slots + t.slot, b);
      else this_slots[t.slot] = b;
    }
    return;
  }
  if (A) MQ_ADD32(this_slots + t.slot, b);
  else this_slots[t.slot] = (int32_t)((uint32_t)this_slots[t.slot] + (uint32_t)b);
}

// ---------------------------------------------------------------- reduce one entry
// Entry `e` of another buffer of the same layout folded into this_buf: baseline rows are
// re-hashed (get_group_value_reduction, ResultSetReduction.cpp:783-826), perfect / non-grouped
// rows are index aligned with the key columns copied from the right-hand side
// (ResultSetReductionJIT.cpp:705-711).  Returns 0 or MI355Q_ERR_OUT_OF_SLOTS.
template <bool A>
MQ_FN int32_t reduce_entry(const DevPlan& p, int idx_target_as_key, int64_t* this_buf,
                           const int64_t* src, int64_t e) {
  if (is_empty_row(p, src, idx_target_as_key)) return 0;
  int64_t* slots;
  if (p.desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
    if (p.n_group <= 1) {
      slots = baseline_find_or_insert(this_buf, (uint32_t)p.entry_count, p.row_quad, p.key_width,
                                      row_key_component(src, p.key_width, 0));
    } else {
      int64_t keys[MI355Q_MAX_GROUP_COLS];
      for (int g = 0; g < p.n_group; ++g) keys[g] = row_key_component(src, p.key_width, g);
      bool bad = false;
      slots = baseline_find_or_insert_multi(this_buf, (uint32_t)p.entry_count, p.row_quad,
                                            p.key_width, p.n_group, keys, &bad);
    }
    if (!slots) return MI355Q_ERR_OUT_OF_SLOTS;
  } else {
    int64_t* row = this_buf + e * p.row_quad;
    for (int k = p.key_quad - 1; k >= 0; --k) {
      if (A) MQ_STORE64(row + k, src[k]);
      else row[k] = src[k];
    }
    slots = row + p.key_quad;
  }
  const int64_t* that_slots = src + p.key_quad;
  if (p.slot_width == 4) {
    for (int i = 0; i < p.n_targets; ++i) {
      reduce_target_compact<A>(p.targets[i], p.init_vals, (int32_t*)slots, (const int32_t*)that_slots);
    }
    return 0;
  }
  for (int i = 0; i < p.n_targets; ++i) {
    reduce_target<A>(p.targets[i], p.init_vals, slots, that_slots);
  }
  return 0;
}

// The whole filter of a step: every plain qual TRUE and, of every disjunction (quals sharing an or_group), at least one
// member TRUE — the reference's logical_and / logical_or over nullable booleans followed by toBool (LogicalIR.cpp:299-352:
// a NULL condition is not TRUE).
MQ_FN bool quals_pass(const DevPlan& p, const int8_t* const* cols, int64_t pos) {
  uint32_t seen = 0, any = 0;
  for (int i = 0; i < p.n_quals; ++i) {
    const DevQual& q = p.quals[i];
    const bool t = eval_qual(q, cols[q.col], pos);
    if (q.or_group == 0) {
      if (!t) return false;
    } else {
      seen |= 1u << q.or_group;
      if (t) any |= 1u << q.or_group;
    }
  }
  return seen == any;
}

// ---------------------------------------------------------------- the row function
// Returns 0, or a HeavyDB-style error: < 0 when the baseline table is full.
template <bool A>
MQ_FN int32_t process_row(const DevPlan& p, const int8_t* const* cols, int64_t pos,
                          int64_t* out_buf, int64_t* nongrouped_slots) {
  if (!quals_pass(p, cols, pos)) return 0;
  JoinMatch jm{nullptr, -1, 1};  // no join: one pass with no inner row
  if (p.join_col >= 0) {
    int64_t jk[MI355Q_MAX_GROUP_COLS];
    bool null_key = false;
    for (int i = 0; i < p.join_n_keys; ++i) {
      jk[i] = decode_int(cols[p.join_cols[i]], p.join_types[i], pos);
      null_key = null_key || (p.join_nullables[i] && jk[i] == int_null_of(p.join_types[i]));
    }
    jm.count = 0;
    if (!null_key) jm = join_lookup(p, jk);  // a NULL key matches nothing (hash_join_idx_nullable)
    if (jm.count <= 0) {
      if (p.join_kind != MI355Q_JOIN_LEFT) return 0;  // INNER: the row is dropped
      jm.ids = nullptr;  // LEFT: kept once, inner side NULL
      jm.single = -1;
      jm.count = 1;
    }
  }
  int64_t* slots;
  int64_t keys[MI355Q_MAX_GROUP_COLS] = {0, 0, 0, 0};  // the group columns' values as decoded
  if (p.desc_type == MI355Q_NON_GROUPED_AGGREGATE) {
    slots = nongrouped_slots;
  } else {
    for (int g = 0; g < p.n_group; ++g) {
      // DOUBLE keys: decode_int's 8-byte load IS the bit pattern; FLOAT keys are widened to double
      // first (castToTypeIn(group_key, 64) then bitcast, IRCodegen.cpp:1505-1507)
      keys[g] = type_is_f32(p.group_types[g])
                    ? dbl_bits((double)*(const float*)(cols[p.group_cols[g]] + pos * 4))
                    : decode_int(cols[p.group_cols[g]], p.group_types[g], pos);
    }
    if (p.desc_type == MI355Q_GROUP_BY_PERFECT_HASH) {
      // entry index: single column key - min (get_group_value_fast, GroupByRuntime.cpp:208-223);
      // several columns sum_i (key_i - min_i) * prod_{j<i} card_j (perfect_key_hash,
      // GroupByAndAggregate.cpp:1546-1598).  The key columns of the row hold the TRANSLATED
      // keys (NULL -> max + 1), as both get_group_value_fast and
      // get_matching_group_value_perfect_hash (RuntimeFunctions.cpp:2077-2091) store them.
      int64_t tk[MI355Q_MAX_GROUP_COLS];
      int64_t idx = 0;
      for (int g = 0; g < p.n_group; ++g) {
        int64_t k = keys[g];
        if (p.group_translate[g] && k == int_null_of(p.group_types[g])) k = p.group_null_key[g];
        tk[g] = k;
        int64_t d = k - p.group_min[g];
        if (p.group_bucket[g]) d /= p.group_bucket[g];
        if (d < 0 || d >= p.group_card[g]) return MI355Q_ERR_OUT_OF_SLOTS;
        idx += d * p.group_mul[g];
      }
      if (idx < 0 || idx >= p.entry_count) return MI355Q_ERR_OUT_OF_SLOTS;
      int64_t* row = out_buf + idx * p.row_quad;
      if (!p.keyless) {
        if (MQ_LOAD64(row) == kEmptyKey64) {
          for (int g = p.n_group - 1; g >= 0; --g) MQ_STORE64(row + g, tk[g]);
        }
        slots = row + p.n_group;
      } else {
        slots = row;
        if (p.col0_key_quirk) a_min_i64<A>(slots, tk[0]);
      }
    } else {
      if (p.n_group == 1) {
        slots = baseline_find_or_insert(out_buf, (uint32_t)p.entry_count, p.row_quad, p.key_width,
                                        keys[0]);
      } else {
        bool bad = false;
        slots = baseline_find_or_insert_multi(out_buf, (uint32_t)p.entry_count, p.row_quad,
                                              p.key_width, p.n_group, keys, &bad);
        if (bad) return MI355Q_ERR_INVALID_PLAN;
      }
      if (!slots) {
        const int64_t code = -(pos + 1);
        return code < INT32_MIN ? INT32_MIN : (int32_t)code;
      }
    }
  }
  // one joined row per matching inner row (JoinLoop over the matching set); the group keys
  // are outer columns, so the group slot is the same for all of them
  for (int m = 0; m < jm.count; ++m) {
    const int64_t inner_pos = jm.ids ? (int64_t)jm.ids[m] : jm.single;
    for (int i = 0; i < p.n_targets; ++i) {
      apply_target<A>(p.targets[i], slots, cols, pos, p.inner_cols, inner_pos, keys);
    }
  }
  return 0;
}

}  // namespace mq
