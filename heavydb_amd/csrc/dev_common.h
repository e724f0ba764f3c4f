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
#define MQ_D_NOINLINE static __device__ __attribute__((noinline))
#else
#define MQ_HD inline
#define MQ_D inline
#define MQ_D_NOINLINE static __attribute__((noinline))
#endif

namespace mq {

constexpr int64_t kEmptyKey64 = INT64_MAX;  // GpuRtConstants.h:27
constexpr int32_t kEmptyKey32 = INT32_MAX;  // GpuRtConstants.h:28
constexpr double kNullDouble = 2.2250738585072014e-308;  // NULL_DOUBLE = DBL_MIN
constexpr int64_t kNullDoubleBits = 0x0010000000000000ll;
constexpr float kNullFloat = 1.17549435e-38f;            // NULL_FLOAT = FLT_MIN
constexpr int32_t kNullFloatBits = 0x00800000;

// A column "type code" packs what the decoders need into one int: the storage type, the
// encoding, the SQL type after decoding and (for encoded columns) the nullable flag:
//   code = storage | encoding << 4 | logical << 8 | nullable << 12
// A plain (unencoded) column's code is just its mi355q_type, so `code == MI355Q_INT64` style
// tests keep selecting plain columns only — encoded columns never match the fast families.
MQ_HD int tc_storage(int c) { return c & 15; }
MQ_HD int tc_enc(int c) { return (c >> 4) & 15; }
MQ_HD int tc_logical(int c) { return ((c >> 8) & 15) ? ((c >> 8) & 15) : (c & 15); }
MQ_HD int tc_nullable(int c) { return (c >> 12) & 1; }
MQ_HD int tc_make(int storage, int enc, int logical, int nullable) {
  return storage | (enc << 4) | (logical << 8) | ((nullable ? 1 : 0) << 12);
}

MQ_HD int plain_width(int t) {
  return t == MI355Q_INT8 ? 1 : t == MI355Q_INT16 ? 2 : (t == MI355Q_INT32 || t == MI355Q_FLOAT) ? 4 : 8;
}
// bytes per element of the chunk as stored
MQ_HD int type_width(int code) { return plain_width(tc_storage(code)); }
MQ_HD bool type_is_fp(int code) { return tc_storage(code) == MI355Q_DOUBLE; }
MQ_HD bool type_is_f32(int code) { return tc_storage(code) == MI355Q_FLOAT; }
// Shared/InlineNullValues.h:29-35
MQ_HD int64_t plain_int_null(int t) {
  return t == MI355Q_INT8    ? (int64_t)INT8_MIN
         : t == MI355Q_INT16 ? (int64_t)INT16_MIN
         : t == MI355Q_INT32 ? (int64_t)INT32_MIN
                             : INT64_MIN;
}
// NULL sentinel of the column's values AFTER decoding (the logical type's)
MQ_HD int64_t int_null_of(int code) { return plain_int_null(tc_logical(code)); }

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
// ... n 4-byte blocks (a multi-column key of n_words * 4 bytes), seed 0:
// key_hash(key, key_count, key_byte_width) (GroupByRuntime.cpp:20-23)
MQ_HD uint32_t murmur3_words(const uint32_t* w, int n_words) {
  uint32_t h1 = 0;
  for (int i = 0; i < n_words; ++i) {
    uint32_t k1 = w[i] * 0xcc9e2d51u;
    k1 = rotl32(k1, 15) * 0x1b873593u;
    h1 ^= k1;
    h1 = rotl32(h1, 13) * 5u + 0xe6546b64u;
  }
  h1 ^= (uint32_t)(n_words * 4);
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

// MurmurHash1 over n 4-byte words, seed 0: the hash of a (composite) keyed-join key of
// n_words * 4 bytes (get_composite_key_index_impl / baseline_hash_join_idx_impl,
// JoinHashTableQueryRuntime.cpp:35-94,140-163)
MQ_HD uint32_t murmur1_words(const uint32_t* w, int n_words) {
  const uint32_t m = 0xc6a4a793u;
  uint32_t h = 0u ^ ((uint32_t)(n_words * 4) * m);
  for (int i = 0; i < n_words; ++i) {
    h += w[i];
    h *= m;
    h ^= h >> 16;
  }
  h *= m;
  h ^= h >> 10;
  h *= m;
  h ^= h >> 17;
  return h;
}

// Pack the components of a join key the way the table stores them (int32[] or int64[]);
// returns the number of 4-byte words.
MQ_HD int pack_join_key(const int64_t* keys, int n_keys, int width, uint32_t* words) {
  if (width == 4) {
    for (int i = 0; i < n_keys; ++i) words[i] = (uint32_t)(int32_t)keys[i];
    return n_keys;
  }
  for (int i = 0; i < n_keys; ++i) {
    words[2 * i] = (uint32_t)(uint64_t)keys[i];
    words[2 * i + 1] = (uint32_t)((uint64_t)keys[i] >> 32);
  }
  return 2 * n_keys;
}

// BASELINE.md section 3 generator: u = splitmix64(seed ^ row * golden)
MQ_HD uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// fixed_width_int_decode (DecodersImpl.h:27-55): sign-extending fixed-width load.
MQ_HD int64_t load_int(const int8_t* col, int t, int64_t pos) {
  switch (t) {
    case MI355Q_INT8: return *(const int8_t*)(col + pos);
    case MI355Q_INT16: return *(const int16_t*)(col + pos * 2);
    case MI355Q_INT32: return *(const int32_t*)(col + pos * 4);
    default: return *(const int64_t*)(col + pos * 8);
  }
}
constexpr int64_t kSecsPerDay = 86400;
// Column element fetch for a type code: the plain load, or the encoded variants
//   ENC_FIXED         fixed_width_int_decode + codgenAdjustFixedEncNull (ColumnIR.cpp:456-495):
//                     the storage sentinel becomes the logical type's sentinel
//   ENC_DICT          fixed_width_unsigned_decode (DecodersImpl.h:57-85) for 1/2-byte ids,
//                     NULL = 255 / 65535 -> NULL_INT
//   ENC_DATE_IN_DAYS  fixed_width_small_date_decode (DecodersImpl.h:130-139): days -> seconds,
//                     storage NULL -> NULL_BIGINT
// ... the same decoders applied to a value ALREADY LOADED from the chunk (`v` = the storage integer, sign-extended): the
// vector loads of the streaming kernels fetch four rows at once and decode afterwards
MQ_HD int64_t decode_loaded(int code, int64_t v) {
  if (code < 16) return v;
  const int st = tc_storage(code);
  switch (tc_enc(code)) {
    case MI355Q_ENC_FIXED:
      return (tc_nullable(code) && v == plain_int_null(st)) ? plain_int_null(tc_logical(code)) : v;
    case MI355Q_ENC_DICT: {
      if (st == MI355Q_INT8) {
        const int64_t u = v & 0xff;
        return (tc_nullable(code) && u == 255) ? (int64_t)INT32_MIN : u;
      }
      if (st == MI355Q_INT16) {
        const int64_t u = v & 0xffff;
        return (tc_nullable(code) && u == 65535) ? (int64_t)INT32_MIN : u;
      }
      return v;
    }
    case MI355Q_ENC_DATE_IN_DAYS:
      return v == plain_int_null(st) ? INT64_MIN : v * kSecsPerDay;
    default:
      return v;
  }
}
MQ_HD int64_t decode_int(const int8_t* col, int code, int64_t pos) {
  if (code < 16) return load_int(col, code, pos);
  return decode_loaded(code, load_int(col, tc_storage(code), pos));
}
// fixed_width_double_decode (DecodersImpl.h:121-128)
MQ_HD double decode_dbl(const int8_t* col, int64_t pos) { return *(const double*)(col + pos * 8); }

MQ_HD int64_t dbl_bits(double d) {
  union { double d; int64_t i; } u;
  u.d = d;
  return u.i;
}
// float bits in the low half of an 8-byte slot; the value an initialised slot holds is the
// int32 pattern sign-extended (get_agg_initial_val returns int64, OutputBufferInitialization.cpp)
MQ_HD int32_t flt_bits(float f) {
  union { float f; int32_t i; } u;
  u.f = f;
  return u.i;
}
MQ_HD float bits_flt(int32_t i) {
  union { float f; int32_t i; } u;
  u.i = i;
  return u.f;
}
// fixed_width_float_decode (DecodersImpl.h:109-119)
MQ_HD float decode_flt(const int8_t* col, int64_t pos) { return *(const float*)(col + pos * 4); }
MQ_HD double bits_dbl(int64_t i) {
  union { double d; int64_t i; } u;
  u.i = i;
  return u.d;
}

// ------------------------------------------------------------------ device-side plan
struct DevQual {
  int32_t col, op, type, nullable;  // type: type code of the column; op: the comparison alone
  int64_t ival;
  double fval;
  int32_t or_group, pad_;           // 0 = a conjunct of its own; 1..3 = member of that disjunction
};
struct DevTarget {
  int32_t agg, col, table, arg_type;  // arg_type: type code of the argument column
  int32_t arg_nullable, skip_null, slot, arg_fp;
  int32_t key_idx, arg_f32;  // PROJECT_KEY: which group column; FLOAT argument (slot = float bits)
  // the argument column's ExpressionRange where it is a valid integer range inside INT32 (arg_rng = 1: [arg_lo, arg_hi],
  // arg_has_nulls as the range says): what the index-partitioned family packs narrow records from (kernels_idx.hip) — a
  // HINT, every kernel that uses it must stay exact for a value outside it
  int32_t arg_rng, arg_lo, arg_hi, arg_has_nulls;
  DevQual cond;           // COUNT_IF / SUM_IF
};
struct DevPlan {
  int32_t n_cols, n_quals, n_targets, slot_count;
  DevQual quals[MI355Q_MAX_QUALS];
  DevTarget targets[MI355Q_MAX_TARGETS];
  int32_t desc_type, keyless, key_width, row_quad;
  int32_t key_quad, group_col, group_type, group_nullable;  // first (or only) group column
  int64_t entry_count, min_val, max_val;
  // all group columns (n_group == 1 repeats the fields above)
  int32_t n_group, slot_width;  // slot_width: 8, or 4 for the compact COUNT(*)-only layouts
  int32_t group_cols[MI355Q_MAX_GROUP_COLS], group_types[MI355Q_MAX_GROUP_COLS];
  int32_t group_translate[MI355Q_MAX_GROUP_COLS];  // perfect hash: NULL key -> group_null_key
  int64_t group_min[MI355Q_MAX_GROUP_COLS], group_card[MI355Q_MAX_GROUP_COLS];
  int64_t group_mul[MI355Q_MAX_GROUP_COLS], group_null_key[MI355Q_MAX_GROUP_COLS];
  int64_t group_bucket[MI355Q_MAX_GROUP_COLS];  // 0 = not bucketed
  int64_t init_vals[MI355Q_MAX_SLOTS];
  // join
  int32_t join_col, join_type, join_nullable, join_hash_type;
  const void* join_buf;
  const uint32_t* join_bitmap;  // perfect tables: 1 bit per slot = "slot holds a row id" (or null)
  int64_t join_min, join_max, join_entries;
  // composite keys / one-to-many / LEFT (join_hash_type: 0 perfect 1:1, 1 keyed 1:1,
  // 2 perfect 1:N, 3 keyed 1:N — layouts in include/mi355q.h)
  int32_t join_n_keys, join_width;  // key components and their width in bytes (keyed tables)
  int32_t join_cols[MI355Q_MAX_GROUP_COLS], join_types[MI355Q_MAX_GROUP_COLS];
  int32_t join_nullables[MI355Q_MAX_GROUP_COLS];
  int32_t join_kind;
  // columnar keyless single-column perfect hash whose first slot starts at EMPTY_KEY_64 (MIN over a NOT NULL
  // 8-byte integer): get_columnar_group_bin_offset (GroupByRuntime.cpp:228-239) takes the first slot's column for
  // the key column and writes the (translated) key into a slot that still holds the init value before the
  // aggregate runs — per kernel and in row order that is MIN(key, values); applied here as one more MIN
  int32_t col0_key_quirk;
  // the step's filter is NOT in quals[] (n_quals = 0) but a BoolFilter compiled at plan time (boolfilter.h), handed to the
  // kernel beside the plan: only families that take one may run the step
  int32_t bf_active;
  const int8_t* inner_cols[MI355Q_MAX_COLS];
};

// ------------------------------------------------------------------ projected expressions
// mi355q_expr lowered by plan.cpp (lower_exprs): every node knows the type and nullability of what it
// pops, so the evaluator (expr.h) is a flat loop without type inference.
enum : int32_t { EXF_NULLABLE = 1, EXF_LHS_NULLABLE = 2, EXF_RHS_NULLABLE = 4, EXF_SHORT_CIRCUIT = 8 };
struct DevExprNode {
  int32_t op, type;  // mi355q_expr_op; plain mi355q_type of the result
  int32_t arg;       // EX_COL: column index; EX_CAST / comparisons / EX_IS_NULL: the operand's type
  int32_t flags;     // EXF_*: result / lhs (or the operand of a unary op) / rhs nullable; EX_AND / EX_OR: the short-circuit form
  int64_t ilit;      // EX_LIT (integers); EX_COL: the column's type code
  double flit;       // EX_LIT (DOUBLE / FLOAT)
};
struct DevExpr {
  int32_t n_nodes, type, nullable;
  int32_t store_type;  // the type of the dense temporary column a projection pass writes: `type`, or INT32 for a BOOLEAN
                       // that only quals read (the typed families filter on 4- and 8-byte columns; plan.cpp lower_exprs)
  DevExprNode nodes[MI355Q_MAX_EXPR_NODES];
};
struct DevExprSet {
  int32_t n, n_cols;  // expressions; physical columns (expression k is written as column n_cols + k)
  // k_project: the physical columns it loads up front for every row of a tile (expr.h eval_expr_rows `raw`), as the COL
  // node that reads them describes them (type / type code)
  int32_t n_pre, pre_col[4], pre_type[4], pre_code[4];
  int32_t noerr_mask;  // bit k: expression k cannot raise an error (no arithmetic, cast or unary minus in it)
  int32_t pad_[2];
  DevExpr e[MI355Q_MAX_EXPRS];
};

// geometry of a columnar result buffer (output_columnar_; rowfunc.h entry_to_columns)
struct ColLayout {
  int64_t entry_count;
  int64_t slot_col_bytes;  // align_to_int64(slot_width * entry_count)
  int32_t key_quads;       // group columns stored (0 when keyless)
  int32_t slot_count, slot_width, row_quad;
};

}  // namespace mq
