/*
 * oracle.cpp — CPU restatement of HeavyDB's CPU executor semantics for ONE query
 * step (scan/filter -> group-by + aggregate -> join probe).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (libmi355q.so)
 * never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED — every function below is a restatement of the cited
 * reference lines and is checked (tests/test_oracle_golden.py) against
 *   (1) the known-answer vectors in tests/golden/ref_vectors.json, generated from the
 *       reference's own sources compiled in place (oracle/_ref, oracle/Makefile,
 *       oracle/gen_golden.py), and
 *   (2) ports of the reference's own ResultSet fill / reduce / iterate tests
 *       (Tests/ResultSetTest.cpp Reduce.* / Iterate.*, row-wise and columnar, one and two key
 *       columns, keyed and keyless, baseline) in tests/test_resultset_style.py.
 *
 * All citations are relative to the heavyai/heavydb tree.
 */
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <thread>
#include <vector>

#include "../include/mi355q.h"

#define ORC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr int64_t kEmptyKey64 = INT64_MAX;  // GpuRtConstants.h:27 EMPTY_KEY_64
constexpr int32_t kEmptyKey32 = INT32_MAX;  // GpuRtConstants.h:28 EMPTY_KEY_32
constexpr size_t kMaxBufferSize = size_t(1) << 30;  // GroupByAndAggregate.cpp:57

// ---------------------------------------------------------------- hashing
inline uint32_t rotl32(uint32_t x, int8_t r) {
  return (x << r) | (x >> (32 - r));
}

// MurmurHash3Inl.h:11-72 (MurmurHash3_x86_32)
uint32_t murmur3(const void* key, int len, uint32_t seed) {
  const uint8_t* data = static_cast<const uint8_t*>(key);
  const int nblocks = len / 4;
  uint32_t h1 = seed;
  const uint32_t c1 = 0xcc9e2d51, c2 = 0x1b873593;
  for (int i = 0; i < nblocks; ++i) {
    uint32_t k1;
    memcpy(&k1, data + 4 * i, 4);
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64;
  }
  const uint8_t* tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3:
      k1 ^= tail[2] << 16;
      [[fallthrough]];
    case 2:
      k1 ^= tail[1] << 8;
      [[fallthrough]];
    case 1:
      k1 ^= tail[0];
      k1 *= c1;
      k1 = rotl32(k1, 15);
      k1 *= c2;
      h1 ^= k1;
  }
  h1 ^= len;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6b;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35;
  h1 ^= h1 >> 16;
  return h1;
}

// MurmurHash1Inl.h:22-62
uint32_t murmur1(const void* key, int len, uint32_t seed) {
  const unsigned int m = 0xc6a4a793;
  unsigned int h = seed ^ (len * m);
  const unsigned char* data = static_cast<const unsigned char*>(key);
  while (len >= 4) {
    unsigned int k;
    memcpy(&k, data, 4);
    h += k;
    h *= m;
    h ^= h >> 16;
    data += 4;
    len -= 4;
  }
  switch (len) {
    case 3:
      h += data[2] << 16;
      [[fallthrough]];
    case 2:
      h += data[1] << 8;
      [[fallthrough]];
    case 1:
      h += data[0];
      h *= m;
      h ^= h >> 16;
  }
  h *= m;
  h ^= h >> 10;
  h *= m;
  h ^= h >> 17;
  return h;
}

// ---------------------------------------------------------------- types / nulls
inline int type_width(int t) {
  switch (t) {
    case MI355Q_INT8: return 1;
    case MI355Q_INT16: return 2;
    case MI355Q_INT32: return 4;
    case MI355Q_INT64: return 8;
    case MI355Q_DOUBLE: return 8;
    case MI355Q_FLOAT: return 4;
  }
  return 0;
}
inline bool type_is_fp(int t) { return t == MI355Q_DOUBLE; }
inline bool type_is_f32(int t) { return t == MI355Q_FLOAT; }

// Shared/InlineNullValues.h:29-35
inline int64_t int_null_of(int t) {
  switch (t) {
    case MI355Q_INT8: return INT8_MIN;
    case MI355Q_INT16: return INT16_MIN;
    case MI355Q_INT32: return INT32_MIN;
    default: return INT64_MIN;
  }
}
inline int64_t dbl_bits(double d) {
  int64_t r;
  memcpy(&r, &d, 8);
  return r;
}
inline double bits_dbl(int64_t b) {
  double r;
  memcpy(&r, &b, 8);
  return r;
}
constexpr double kNullDouble = DBL_MIN;  // NULL_DOUBLE
constexpr float kNullFloat = FLT_MIN;    // NULL_FLOAT
inline int32_t flt_bits(float f) {
  int32_t r;
  memcpy(&r, &f, 4);
  return r;
}
inline float bits_flt(int32_t b) {
  float r;
  memcpy(&r, &b, 4);
  return r;
}

// DecodersImpl.h:27-55 fixed_width_int_decode (sign-extending load), :121-128 double
inline int64_t decode_int(const int8_t* col, int t, int64_t pos) {
  switch (t) {
    case MI355Q_INT8: return *reinterpret_cast<const int8_t*>(col + pos);
    case MI355Q_INT16: return *reinterpret_cast<const int16_t*>(col + pos * 2);
    case MI355Q_INT32: return *reinterpret_cast<const int32_t*>(col + pos * 4);
    default: return *reinterpret_cast<const int64_t*>(col + pos * 8);
  }
}
// DecodersImpl.h:57-85 fixed_width_unsigned_decode
inline int64_t decode_unsigned(const int8_t* col, int t, int64_t pos) {
  switch (t) {
    case MI355Q_INT8: return *reinterpret_cast<const uint8_t*>(col + pos);
    case MI355Q_INT16: return *reinterpret_cast<const uint16_t*>(col + pos * 2);
    case MI355Q_INT32: return *reinterpret_cast<const uint32_t*>(col + pos * 4);
    default: return (int64_t) * reinterpret_cast<const uint64_t*>(col + pos * 8);
  }
}
// The SQL type a column's values have once decoded (get_col_bit_width / the column's
// SQLTypeInfo): ENC_FIXED carries it, dictionary ids are INT, dates-in-days are 64-bit.
inline int logical_type_of(const mi355q_col_desc& cd) {
  switch (cd.encoding) {
    case MI355Q_ENC_FIXED: return cd.logical_type;
    case MI355Q_ENC_DICT: return MI355Q_INT32;
    case MI355Q_ENC_DATE_IN_DAYS: return MI355Q_INT64;
    default: return cd.type;
  }
}
// Column fetch as the row function sees it (CodeGenerator::codegenFixedLengthColVar,
// ColumnIR.cpp:258-306): decoder chosen by get_col_decoder, then for nullable FIXED / small
// DICT columns the storage NULL is widened to the logical NULL (codgenAdjustFixedEncNull
// :456-495 -> cast_<from>_to_<to>_nullable); FixedWidthSmallDate
// (fixed_width_small_date_decode, DecodersImpl.h:130-139) maps NULL and scales days to seconds.
inline int64_t decode_col(const mi355q_col_desc& cd, const int8_t* col, int64_t pos) {
  switch (cd.encoding) {
    case MI355Q_ENC_FIXED: {
      const int64_t v = decode_int(col, cd.type, pos);
      if (cd.nullable && v == int_null_of(cd.type)) return int_null_of(cd.logical_type);
      return v;
    }
    case MI355Q_ENC_DICT: {
      if (type_width(cd.type) >= 4) return decode_int(col, cd.type, pos);
      const int64_t v = decode_unsigned(col, cd.type, pos);
      const int64_t enc_null = cd.type == MI355Q_INT8 ? 255 : 65535;  // inline_fixed_encoding_null_val
      if (cd.nullable && v == enc_null) return INT32_MIN;
      return v;
    }
    case MI355Q_ENC_DATE_IN_DAYS: {
      const int64_t v = decode_int(col, cd.type, pos);
      return v == int_null_of(cd.type) ? INT64_MIN : v * 86400;
    }
    default:
      return decode_int(col, cd.type, pos);
  }
}
inline double decode_dbl(const int8_t* col, int64_t pos) {
  return *reinterpret_cast<const double*>(col + pos * 8);
}
// DecodersImpl.h:109-119 fixed_width_float_decode
inline float decode_flt(const int8_t* col, int64_t pos) {
  return *reinterpret_cast<const float*>(col + pos * 4);
}

// ---------------------------------------------------------------- aggregates (CPU)
// RuntimeFunctions.cpp:362
inline void agg_count(int64_t* agg) { ++*reinterpret_cast<uint64_t*>(agg); }
// :1151
// (the reference's `*agg += val` wraps on x86; written with unsigned arithmetic so that the wrap is
// defined behaviour here too)
inline void agg_sum(int64_t* agg, int64_t val) { *agg = (int64_t)((uint64_t)*agg + (uint64_t)val); }
// :1163-1169
inline void agg_max(int64_t* agg, int64_t val) { *agg = std::max(*agg, val); }
inline void agg_min(int64_t* agg, int64_t val) { *agg = std::min(*agg, val); }
// :1313-1325
inline void agg_sum_skip_val(int64_t* agg, int64_t val, int64_t skip_val) {
  const auto old = *agg;
  if (val != skip_val) {
    if (old != skip_val) {
      agg_sum(agg, val);
    } else {
      *agg = val;
    }
  }
}
// :1401-1431 DEF_SKIP_AGG(agg_max / agg_min)
inline void agg_max_skip_val(int64_t* agg, int64_t val, int64_t skip_val) {
  if (val != skip_val) {
    const int64_t old = *agg;
    if (old != skip_val) {
      agg_max(agg, val);
    } else {
      *agg = val;
    }
  }
}
inline void agg_min_skip_val(int64_t* agg, int64_t val, int64_t skip_val) {
  if (val != skip_val) {
    const int64_t old = *agg;
    if (old != skip_val) {
      agg_min(agg, val);
    } else {
      *agg = val;
    }
  }
}
// :1444-1470
inline void agg_sum_double(int64_t* agg, double val) {
  *agg = dbl_bits(bits_dbl(*agg) + val);
}
inline void agg_max_double(int64_t* agg, double val) {
  *agg = dbl_bits(std::max(bits_dbl(*agg), val));
}
inline void agg_min_double(int64_t* agg, double val) {
  *agg = dbl_bits(std::min(bits_dbl(*agg), val));
}
// :1558-1584 DEF_SKIP_AGG for double
inline void agg_sum_double_skip_val(int64_t* agg, double val, double skip_val) {
  if (val != skip_val) {
    const int64_t old = *agg;
    if (old != dbl_bits(skip_val)) {
      agg_sum_double(agg, val);
    } else {
      *agg = dbl_bits(val);
    }
  }
}
inline void agg_max_double_skip_val(int64_t* agg, double val, double skip_val) {
  if (val != skip_val) {
    const int64_t old = *agg;
    if (old != dbl_bits(skip_val)) {
      agg_max_double(agg, val);
    } else {
      *agg = dbl_bits(val);
    }
  }
}
inline void agg_min_double_skip_val(int64_t* agg, double val, double skip_val) {
  if (val != skip_val) {
    const int64_t old = *agg;
    if (old != dbl_bits(skip_val)) {
      agg_min_double(agg, val);
    } else {
      *agg = dbl_bits(val);
    }
  }
}

// RuntimeFunctions.cpp:1496-1520 agg_sum_float / agg_max_float / agg_min_float: single precision
// on the low 4 bytes of the slot; :1586-1594 DEF_SKIP_AGG for float
inline void agg_sum_float(int32_t* agg, float val) { *agg = flt_bits(bits_flt(*agg) + val); }
inline void agg_max_float(int32_t* agg, float val) { *agg = flt_bits(std::max(bits_flt(*agg), val)); }
inline void agg_min_float(int32_t* agg, float val) { *agg = flt_bits(std::min(bits_flt(*agg), val)); }
#define ORC_SKIP_AGG_F32(name)                                                   \
  inline void name##_skip_val(int32_t* agg, float val, float skip_val) {         \
    if (val != skip_val) {                                                       \
      const int32_t old = *agg;                                                  \
      if (old != flt_bits(skip_val)) {                                           \
        name(agg, val);                                                          \
      } else {                                                                   \
        *agg = flt_bits(val);                                                    \
      }                                                                          \
    }                                                                            \
  }
ORC_SKIP_AGG_F32(agg_sum_float)
ORC_SKIP_AGG_F32(agg_max_float)
ORC_SKIP_AGG_F32(agg_min_float)

// ---------------------------------------------------------------- layout decisions
struct TargetDesc {
  int agg;
  int col;
  int table;
  int arg_type;      // 0 for COUNT(*)
  bool arg_nullable;
  bool arg_fp;
  bool arg_f32;      // FLOAT argument (takes_float_argument): 4-byte slot arithmetic
  bool skip_null;    // TargetInfo.skip_null_val after TargetExprBuilder.cpp:684-690
  bool constrained;  // constrained_not_null(arg, quals)
  int slot;          // first slot, -1 if read from key columns
  int n_slots;
  int key_idx;       // PROJECT_KEY: which group column
  const mi355q_col_desc* cd;  // argument column (nullptr for COUNT(*))
  const mi355q_qual* cond;    // COUNT_IF / SUM_IF
};

const mi355q_col_desc& col_desc_of(const mi355q_plan& p, int table, int col) {
  return table ? p.inner_cols[col] : p.cols[col];
}
const mi355q_range& col_range_of(const mi355q_plan& p, int table, int col) {
  return table ? p.inner_col_ranges[col] : p.col_ranges[col];
}

// get_agg_initial_val (OutputBufferInitialization.cpp:132-289) for 8-byte slots.
// `notnull` is the init type's notnull flag (forced false for non-grouped, :79-81).
int64_t agg_initial_val(int agg, int arg_type, bool notnull) {
  const bool fp = type_is_fp(arg_type);
  if (type_is_f32(arg_type)) {
    // byte_width 4 cases (:139-236): the int32 bit pattern of the float, returned as int64
    switch (agg) {
      case MI355Q_SUM:
      case MI355Q_SUM_IF: return notnull ? flt_bits(0.0f) : flt_bits(kNullFloat);
      case MI355Q_MIN: return notnull ? flt_bits(FLT_MAX) : flt_bits(kNullFloat);
      case MI355Q_MAX: return notnull ? flt_bits(-FLT_MAX) : flt_bits(kNullFloat);
      default: return 0;
    }
  }
  switch (agg) {
    case MI355Q_SUM:
    case MI355Q_SUM_IF:  // OutputBufferInitialization.cpp:139-176: kSUM and kSUM_IF share the case
      if (!notnull) {
        // SUM(int) has result type BIGINT -> NULL_BIGINT; SUM(double) -> NULL_DOUBLE bits
        return fp ? dbl_bits(kNullDouble) : INT64_MIN;
      }
      return fp ? dbl_bits(0.0) : 0;
    case MI355Q_AVG:
    case MI355Q_COUNT:
    case MI355Q_COUNT_IF:
      return 0;
    case MI355Q_MIN:
      if (fp) {
        return notnull ? dbl_bits(DBL_MAX) : dbl_bits(kNullDouble);
      }
      return notnull ? INT64_MAX : int_null_of(arg_type);
    case MI355Q_MAX:
      if (fp) {
        return notnull ? dbl_bits(-DBL_MAX) : dbl_bits(kNullDouble);
      }
      return notnull ? INT64_MIN : int_null_of(arg_type);
    default:
      return 0;  // non-agg targets: 0 (init_agg_val_vec :35-47)
  }
}

int build_targets(const mi355q_plan& p, bool is_group_by, std::vector<TargetDesc>& out) {
  out.clear();
  for (int i = 0; i < p.n_targets; ++i) {
    const auto& t = p.targets[i];
    TargetDesc d{};
    d.agg = t.agg;
    d.col = t.col;
    d.table = t.table;
    if (t.agg == MI355Q_PROJECT) {  // a Projection's target (TargetInfo.is_agg == false, no GROUP BY)
      if (is_group_by || t.col < 0) return MI355Q_ERR_INVALID_PLAN;
      if (t.table != 0 && (p.join_outer_col < 0 || t.col >= p.n_inner_cols)) return MI355Q_ERR_INVALID_PLAN;
    }
    if (t.agg == MI355Q_PROJECT_KEY) {
      if (!is_group_by) return MI355Q_ERR_INVALID_PLAN;
      d.key_idx = t.col < 0 ? 0 : t.col;
      if (d.key_idx >= p.n_group_cols) return MI355Q_ERR_INVALID_PLAN;
      d.col = p.group_cols[d.key_idx];
      d.table = 0;
    }
    if (d.col >= 0) {
      const auto& cd = col_desc_of(p, d.table, d.col);
      d.cd = &cd;
      d.arg_type = logical_type_of(cd);
      d.arg_nullable = cd.nullable != 0 || (d.table && p.join_kind == MI355Q_JOIN_LEFT);
      d.arg_fp = type_is_fp(cd.type);
      d.arg_f32 = type_is_f32(cd.type);
    } else if (t.agg != MI355Q_COUNT && t.agg != MI355Q_COUNT_IF) {
      return MI355Q_ERR_INVALID_PLAN;
    }
    if (t.agg == MI355Q_COUNT_IF || t.agg == MI355Q_SUM_IF) {
      if (t.cond.col < 0 || t.cond.col >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
      d.cond = &t.cond;
      if (t.agg == MI355Q_COUNT_IF) {  // the condition is the argument
        d.col = -1;
        d.cd = nullptr;
        d.arg_type = 0;
        d.arg_fp = false;
        d.arg_f32 = false;
        d.arg_nullable = false;
      }
    }
    // constrained_not_null (OutputBufferInitialization.cpp:301-324): a qual `arg IS NOT NULL` (or
    // NOT(arg IS NULL), the same thing here) on the aggregate's own argument expression
    d.constrained = false;
    if (d.col >= 0 && d.table == 0 && t.agg != MI355Q_PROJECT_KEY && t.agg != MI355Q_PROJECT) {
      for (int k = 0; k < p.n_quals; ++k)
        if (p.quals[k].op == MI355Q_IS_NOT_NULL && p.quals[k].col == d.col) d.constrained = true;
    }
    // TargetInfo.cpp:64-81: skip_null_val = !arg.notnull ; TargetExprBuilder.cpp:684-690:
    // non-grouped aggregates with an argument force skip_null_val = true; otherwise a constrained
    // argument turns it off.
    d.skip_null = (d.col >= 0 && t.agg != MI355Q_PROJECT_KEY && t.agg != MI355Q_PROJECT) &&
                  ((d.arg_nullable && !d.constrained) || !is_group_by);
    // COUNT_IF: skip_null_val follows the nullability of the condition (its argument)
    // (`x IS [NOT] NULL` is itself a NOT NULL BOOLEAN)
    if (t.agg == MI355Q_COUNT_IF)
      d.skip_null = (p.cols[t.cond.col].nullable != 0 && t.cond.op != MI355Q_IS_NULL && t.cond.op != MI355Q_IS_NOT_NULL) ||
                    !is_group_by;
    d.n_slots = (t.agg == MI355Q_AVG) ? 2 : 1;
    out.push_back(d);
  }
  return 0;
}

// getExpressionRange-driven part of get_keyless_info (GroupByAndAggregate.cpp:489-648).
// Returns {keyless, slot index of the target acting as key}.
std::pair<bool, int> keyless_info(const mi355q_plan& p, const std::vector<TargetDesc>& ts) {
  bool keyless = true, found = false;
  int index = 0;
  for (const auto& t : ts) {
    const bool is_agg = t.agg != MI355Q_PROJECT_KEY;
    if (!found && is_agg) {
      const mi355q_range* r = t.col >= 0 ? &col_range_of(p, t.table, t.col) : nullptr;
      // getExpressionRange marks a column projected from the inner side of an outer join as
      // having nulls (ExpressionRange.cpp, is_outer_join_proj -> setHasNulls)
      mi355q_range outer_r;
      if (r && t.table && p.join_kind == MI355Q_JOIN_LEFT) {
        outer_r = *r;
        outer_r.has_nulls = 1;
        r = &outer_r;
      }
      switch (t.agg) {
        case MI355Q_AVG:
          ++index;
          if (t.col >= 0 && t.arg_nullable) {
            if (!r->valid || r->has_nulls) break;
          }
          found = true;
          break;
        case MI355Q_COUNT:
          if (t.col >= 0 && t.arg_nullable) {
            if (!r->valid || r->has_nulls) break;
          }
          found = true;
          break;
        case MI355Q_SUM:
          // "if (constrained_not_null(arg_expr, quals)) arg_ti.set_notnull(true)" (:531) — SUM only
          if (t.arg_nullable && !t.constrained) {
            if (r->valid && !r->has_nulls) found = true;
          } else if (r->valid) {
            if ((t.arg_fp || t.arg_f32)) {
              if (r->fp_max < 0 || r->fp_min > 0) found = true;
            } else {
              if (r->max < 0 || r->min > 0) found = true;
            }
          }
          break;
        case MI355Q_MIN: {
          if (!r->valid) break;
          const int64_t init_max = agg_initial_val(MI355Q_MIN, t.arg_type, !t.arg_nullable);
          if ((t.arg_fp || t.arg_f32)) {
            if (r->fp_max < bits_dbl(init_max)) found = true;
          } else {
            if (r->max < init_max) found = true;
          }
          break;
        }
        case MI355Q_MAX: {
          if (!r->valid || r->has_nulls) break;
          const int64_t init_min = agg_initial_val(MI355Q_MAX, t.arg_type, !t.arg_nullable);
          if ((t.arg_fp || t.arg_f32)) {
            if (r->fp_min > bits_dbl(init_min)) found = true;
          } else {
            if (r->min > init_min) found = true;
          }
          break;
        }
        default:
          keyless = false;
      }
    }
    if (!keyless) break;
    if (!found) ++index;
  }
  return {keyless && found, index};
}

// getBucketedCardinality (GroupByAndAggregate.cpp:367-375)
int64_t bucketed_cardinality(const mi355q_range& r) {
  int64_t c = r.max - r.min;
  if (r.bucket > 0) c /= r.bucket;
  return c + 1 + (r.has_nulls ? 1 : 0);
}

// QueryMemoryDescriptor::init, case QueryDescriptionType::Projection (Descriptors/QueryMemoryDescriptor.cpp:394-410)
// + the constructor (:452-546).  A projection's groupby_exprs is one nullptr: get_col_byte_widths gives that "group
// column" — the row offset — 8 bytes (group_col_widths = {8}); entry_count = scan_limit, else
// max_groups_buffer_entry_count; col_slot_context = ColSlotContext(target_exprs, {}) (ColSlotContext.cpp:35-100): one
// slot per target of the target type's LOGICAL width, padded size unset — the constructor pads every unset slot to 8
// (setAllUnsetSlotsPaddedSize(8), :507) unless the buffer is a columnar projection, whose slots keep their logical
// widths (isLogicalSizedColumnsAllowed :1129-1135 -> setAllSlotsPaddedSizeToLogicalSize :540-546).
int qmd_init_projection(const mi355q_plan& p, const std::vector<TargetDesc>& ts, mi355q_qmd& q) {
  if (p.n_group_cols != 0) return MI355Q_ERR_INVALID_PLAN;
  if (p.scan_limit < 0 || p.output_columnar_hint < 0 || p.output_columnar_hint > 1) return MI355Q_ERR_INVALID_PLAN;
  q.desc_type = MI355Q_PROJECTION;
  q.n_targets = p.n_targets;
  q.group_col_count = 1;
  q.idx_target_as_key = -1;
  q.key_width = 8;
  q.key_bytes = 8;
  q.entry_count = p.scan_limit ? p.scan_limit : (p.max_groups_buffer_entry_guess > 0 ? p.max_groups_buffer_entry_guess : 16384);
  if (q.entry_count > INT32_MAX) return MI355Q_ERR_UNSUPPORTED;  // (uint32 pos / int32 total_matched of the runtime)
  q.output_columnar = p.output_columnar_hint == MI355Q_OUTPUT_COLUMNAR;
  q.slot_count = p.n_targets;
  q.slot_width = 8;
  for (int i = 0; i < p.n_targets; ++i) {
    const TargetDesc& t = ts[i];
    q.target_agg[i] = MI355Q_PROJECT;
    q.target_slot[i] = i;
    q.target_is_fp[i] = t.arg_fp || t.arg_f32;
    // row-wise: the value is cast to the slot's 64 bits before agg_id / agg_id_double (castToTypeIn(target_lv,
    // chosen_bytes << 3), TargetExprBuilder.cpp:485-505): a FLOAT becomes a double; columnar: agg_id_float on 4 bytes
    q.target_arg_is_fp[i] = t.arg_fp || (t.arg_f32 && !q.output_columnar);
    q.target_arg_is_f32[i] = t.arg_f32 && q.output_columnar;
    q.slot_bytes[i] = q.output_columnar ? type_width(t.arg_type) : 8;
    q.init_vals[i] = 0;  // init_agg_val_vec: `if (!agg_info.is_agg ...) push_back(0)` (OutputBufferInitialization.cpp:40-47)
    if (!t.arg_nullable) q.target_null[i] = kEmptyKey64;
    else if (t.arg_fp) q.target_null[i] = dbl_bits(kNullDouble);
    else if (t.arg_f32) q.target_null[i] = q.output_columnar ? (int64_t)flt_bits(kNullFloat) : dbl_bits((double)kNullFloat);
    else q.target_null[i] = int_null_of(t.arg_type);
  }
  q.row_size = 8 + 8 * q.slot_count;  // getRowSize: align8(key bytes) + padded slots
  return 0;
}

int qmd_init(const mi355q_plan& p, mi355q_qmd& q) {
  memset(&q, 0, sizeof(q));
  if (p.n_targets < 1 || p.n_targets > MI355Q_MAX_TARGETS) return MI355Q_ERR_INVALID_PLAN;
  if (p.n_group_cols < 0 || p.n_group_cols > MI355Q_MAX_GROUP_COLS) return MI355Q_ERR_INVALID_PLAN;
  const bool is_group_by = p.n_group_cols > 0;
  std::vector<TargetDesc> ts;
  if (int e = build_targets(p, is_group_by, ts)) return e;
  {
    int n_project = 0;
    for (int i = 0; i < p.n_targets; ++i) n_project += p.targets[i].agg == MI355Q_PROJECT;
    if (n_project && n_project != p.n_targets) return MI355Q_ERR_INVALID_PLAN;
    if (n_project) return qmd_init_projection(p, ts, q);
    if (p.scan_limit != 0) return MI355Q_ERR_INVALID_PLAN;
  }

  q.n_targets = p.n_targets;
  q.group_col_count = p.n_group_cols;
  q.idx_target_as_key = -1;
  q.key_width = 8;
  // getColRangeInfo, ExpressionRangeType::Float / Double (GroupByAndAggregate.cpp:199-207): a
  // floating-point group key always takes the baseline layout
  bool fp_key = false;
  for (int g = 0; g < p.n_group_cols; ++g)
    fp_key = fp_key || type_is_fp(p.cols[p.group_cols[g]].type) || type_is_f32(p.cols[p.group_cols[g]].type);
  bool baseline = fp_key;
  if (!is_group_by) {
    q.desc_type = MI355Q_NON_GROUPED_AGGREGATE;  // QueryMemoryDescriptor.cpp:271-300
    q.entry_count = 1;
  } else if (p.n_group_cols == 1) {
    const auto& r = p.col_ranges[p.group_cols[0]];
    // getColRangeInfo, single-column case (GroupByAndAggregate.cpp:295-349)
    if (fp_key || !r.valid || r.min > r.max) {
      baseline = true;
    } else {
      const int64_t col_count = p.n_group_cols + p.n_targets;
      const int64_t max_entry_count = kMaxBufferSize / (col_count * sizeof(int64_t));
      // is_column_range_too_big_for_perfect_hash (:130-139), overflow -> too big
      __int128 span = (__int128)r.max - (__int128)r.min;
      const bool is_baseline_candidate = span > INT64_MAX || (int64_t)span >= max_entry_count;
      // :312-343 a dictionary-encoded string key without a bucket: too big a range only means baseline when
      // the step has filters (no cardinality estimate reaches this seam: ":337 !group_cardinality_estimation_");
      // without filters the perfect hash is kept ("we are better off attempting perfect hash ... and failing
      // later due to excessive memory use") — Tests/GroupByTest.cpp BaselineFallbackTest / BaselineNoFilters
      const bool dict_key = p.cols[p.group_cols[0]].encoding == MI355Q_ENC_DICT && !(r.bucket > 0);
      if (dict_key) {
        if (p.n_quals > 0 && is_baseline_candidate) baseline = true;
      } else if (is_baseline_candidate && !(r.bucket > 0)) {
        // ":344  else if (is_baseline_candidate && !col_range_info.bucket)"
        baseline = true;
      }
      if (!baseline && span / (r.bucket > 0 ? r.bucket : 1) >= INT32_MAX) return MI355Q_ERR_UNSUPPORTED;
    }
    if (!baseline) {
      q.desc_type = MI355Q_GROUP_BY_PERFECT_HASH;
      q.min_val = r.min;
      q.max_val = r.max;
      q.bucket = r.bucket > 0 ? r.bucket : 0;
      q.has_nulls = r.has_nulls;
      q.entry_count = std::max<int64_t>(bucketed_cardinality(r), 1);
      q.group_min[0] = r.min;
      q.group_card[0] = bucketed_cardinality(r);
      q.group_bucket[0] = q.bucket;
      q.group_null_key[0] = r.max + (q.bucket ? q.bucket : 1);  // translated_null_value
      q.group_has_nulls[0] = r.has_nulls;
    }
  } else {
    // getColRangeInfo, several group columns (GroupByAndAggregate.cpp:241-283): the product of
    // the per-column bucketed cardinalities decides; zero, > g_baseline_groupby_threshold
    // (1 000 000, Execute.cpp:113) or a checked_int64_t overflow -> baseline
    __int128 cardinality = 1;
    bool has_nulls = false;
    for (int g = 0; g < p.n_group_cols && !baseline; ++g) {
      const auto& r = p.col_ranges[p.group_cols[g]];
      if (!r.valid || r.min > r.max) {  // get_expr_range_info: not a perfect-hash candidate
        baseline = true;
        break;
      }
      const __int128 crt = ((__int128)r.max - (__int128)r.min) / (r.bucket > 0 ? r.bucket : 1) + 1 +
                           (r.has_nulls ? 1 : 0);
      cardinality *= crt;
      if (crt > INT64_MAX || cardinality > INT64_MAX) baseline = true;
      has_nulls = has_nulls || r.has_nulls;
    }
    if (!baseline && (cardinality == 0 || cardinality > 1000000)) baseline = true;
    if (!baseline) {
      q.desc_type = MI355Q_GROUP_BY_PERFECT_HASH;
      q.min_val = 0;
      q.max_val = (int64_t)cardinality;
      q.has_nulls = has_nulls;
      q.entry_count = (int64_t)cardinality;  // QueryMemoryDescriptor.cpp:336-339
      for (int g = 0; g < p.n_group_cols; ++g) {
        const auto& r = p.col_ranges[p.group_cols[g]];
        q.group_min[g] = r.min;
        q.group_card[g] = bucketed_cardinality(r);
        q.group_bucket[g] = r.bucket > 0 ? r.bucket : 0;
        q.group_null_key[g] = r.max + (r.bucket > 0 ? r.bucket : 1);
        q.group_has_nulls[g] = r.has_nulls;
      }
    }
  }
  if (is_group_by && !baseline) {
    auto ki = keyless_info(p, ts);
    // QueryMemoryDescriptor.cpp:322-327: "... && !col_range_info.bucket && keyless_info.keyless"
    q.keyless = ki.first && !q.bucket;
    q.idx_target_as_key = ki.second;
  } else if (baseline) {
    q.desc_type = MI355Q_GROUP_BY_BASELINE_HASH;
    q.entry_count = p.max_groups_buffer_entry_guess > 0 ? p.max_groups_buffer_entry_guess
                                                        : 16384;
    // pick_baseline_key_width (QueryMemoryDescriptor.cpp:113-146): the widest component;
    // "group_col_compact_width = output_columnar ? 8 : pick_baseline_key_width" (:386-388)
    int kw = p.output_columnar_hint ? 8 : 4;
    for (int g = 0; g < p.n_group_cols && !p.output_columnar_hint; ++g) {
      const auto& gcd = p.cols[p.group_cols[g]];
      const auto& r = p.col_ranges[p.group_cols[g]];
      int w = 8;
      // pick_baseline_key_component_width: "No compaction for floating point yet" -> 8
      if (r.valid && !type_is_fp(gcd.type) && !type_is_f32(gcd.type)) {
        if (type_width(logical_type_of(gcd)) == 8 && r.has_nulls) {
          w = 8;
        } else {
          w = (r.min > INT32_MIN && r.max < kEmptyKey32 - 1) ? 4 : 8;  // is_valid_int32_range
        }
      }
      kw = std::max(kw, w);
    }
    q.key_width = kw;
  }
  // slots: ColSlotContext (ColSlotContext.cpp:35-100), all padded to 8 bytes
  int slot = 0;
  for (int i = 0; i < p.n_targets; ++i) {
    auto& t = ts[i];
    q.target_agg[i] = t.agg;
    q.target_skip_null[i] = t.skip_null;
    q.target_key_idx[i] = t.key_idx;
    q.target_arg_is_fp[i] = t.arg_fp && t.agg != MI355Q_COUNT;
    q.target_arg_is_f32[i] = t.arg_f32 && t.agg != MI355Q_COUNT;
    q.target_is_fp[i] = (t.agg == MI355Q_AVG) || ((t.arg_fp || t.arg_f32) && t.agg != MI355Q_COUNT);
    if (t.agg == MI355Q_PROJECT_KEY && q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
      // target_groupby_indices >= 0 -> zero-width slot (ColSlotContext.cpp:45-49)
      q.target_slot[i] = -1;
      t.slot = -1;
    } else {
      if (slot + t.n_slots > MI355Q_MAX_SLOTS) return MI355Q_ERR_INVALID_PLAN;
      q.target_slot[i] = slot;
      t.slot = slot;
      // init_agg_val_vec (OutputBufferInitialization.cpp:24-84)
      // init_agg_val_vec(targets, quals, qmd) :280-290: constrained -> set_notnull(true); non-grouped
      // aggregates are forced nullable afterwards (:79-81)
      const bool init_notnull = is_group_by ? (!t.arg_nullable || t.constrained) : false;
      q.init_vals[slot] = agg_initial_val(t.agg, t.arg_type, init_notnull);
      if (t.agg == MI355Q_AVG) q.init_vals[slot + 1] = 0;
      slot += t.n_slots;
    }
    // null_val_bit_pattern (ResultSetBufferAccessors.h:229-245) of the result type
    switch (t.agg) {
      case MI355Q_AVG: q.target_null[i] = dbl_bits(kNullDouble); break;
      case MI355Q_SUM:
      case MI355Q_SUM_IF:
        q.target_null[i] = t.arg_f32 ? flt_bits(kNullFloat) : t.arg_fp ? dbl_bits(kNullDouble) : INT64_MIN;
        break;
      case MI355Q_COUNT:
      case MI355Q_COUNT_IF: q.target_null[i] = p.bigint_count ? INT64_MIN : INT32_MIN; break;
      default:
        q.target_null[i] = t.arg_f32 ? flt_bits(kNullFloat)
                                     : t.arg_fp ? dbl_bits(kNullDouble) : int_null_of(t.arg_type);
    }
    // ResultSet::isNull (ResultSetIteration.cpp) checks ti.get_notnull() first: the projection of a
    // NOT NULL key column is never NULL whatever bits it holds; EMPTY_KEY_64 (never a key) = no pattern
    // the key column holds a FLOAT key as the double it was cast to (castToTypeIn(group_key, 64)), its
    // NULL (FLT_MIN) included
    if (t.agg == MI355Q_PROJECT_KEY && t.arg_f32) q.target_null[i] = dbl_bits((double)kNullFloat);
    if (t.agg == MI355Q_PROJECT_KEY && !t.arg_nullable) q.target_null[i] = kEmptyKey64;
  }
  q.slot_count = slot;
  // pick_target_compact_width (QueryMemoryDescriptor.cpp:748-840) -> setAllSlotsPaddedSize (:265):
  // 8 unless !g_bigint_count, exactly one group-by expression, every target either COUNT(*) or a
  // non-aggregate integer of at most 4 bytes (here: the projected key), and the input tables hold
  // at most UINT32_MAX tuples — then 4.  Baseline hash rebuilds the slot context afterwards (:382-384): its
  // slots are left unset and the constructor pads them to 8 (:507), so it never keeps the narrow width.
  bool compact = !p.bigint_count && p.n_group_cols == 1 && q.desc_type != MI355Q_GROUP_BY_BASELINE_HASH &&
                 (uint64_t)std::max<int64_t>(p.num_tuples, 0) <= (uint64_t)UINT32_MAX;
  for (const auto& t : ts) {
    if (t.agg == MI355Q_COUNT && t.col < 0) continue;
    if (t.agg == MI355Q_PROJECT_KEY && type_width(t.arg_type) <= 4 && !t.arg_f32) continue;  // is_int_and_no_bigger_than
    compact = false;
  }
  q.slot_width = compact ? 4 : 8;
  for (int j = 0; j < q.slot_count; ++j) q.slot_bytes[j] = q.slot_width;
  // getRowSize (QueryMemoryDescriptor.cpp:848-860); slots packed back to back
  // (ColSlotContext::getAlignedPaddedSizeForRange, ColSlotContext.cpp:152-166)
  q.key_bytes = 0;
  if (is_group_by && !q.keyless) {
    q.key_bytes = (q.group_col_count * q.key_width + 7) & ~7;
  }
  q.row_size = q.key_bytes + ((q.slot_width * q.slot_count + 7) & ~7);
  if (q.row_size == 0) return MI355Q_ERR_INVALID_PLAN;
  // output_columnar_ = output_columnar_hint for GroupByPerfectHash / GroupByBaselineHash /
  // NonGroupedAggregate without distinct / quantile / mode targets (QueryMemoryDescriptor.cpp:515-531)
  if (p.output_columnar_hint < 0 || p.output_columnar_hint > 2) return MI355Q_ERR_INVALID_PLAN;
  q.output_columnar = p.output_columnar_hint == MI355Q_OUTPUT_COLUMNAR;
  // the library refuses the one columnar shape whose reference behaviour it does not restate
  // (see columnar_bin_single below): keyless single-column perfect hash whose first slot starts at
  // EMPTY_KEY_64
  //   ... restated (the literal get_columnar_group_bin_offset call of the row function) where it is deterministic:
  // an unbucketed key and a plain MIN over a NOT NULL integer in that slot — "key first, then the aggregate" is
  // MIN(key, values) however the rows are dealt to kernels; a bucketed key would leave the FIRST row's key there
  if (q.output_columnar && q.keyless && p.n_group_cols == 1 && q.slot_width == 8 && q.slot_count > 0 &&
      q.init_vals[0] == kEmptyKey64) {
    bool min_first = false;
    for (int i = 0; i < p.n_targets; ++i)
      if (q.target_slot[i] == 0 && q.target_agg[i] == MI355Q_MIN && !q.target_skip_null[i] && !q.target_arg_is_fp[i])
        min_first = true;
    if (q.bucket > 0 || !min_first) return MI355Q_ERR_UNSUPPORTED;
  }
  return 0;
}

// ---------------------------------------------------------------- columnar layout
inline int64_t align_to_int64(int64_t v) { return (v + 7) & ~(int64_t)7; }
// getPrependedGroupColOffInBytes (QueryMemoryDescriptor.cpp:962-975): max(groupColWidth, 8) per key
int64_t col_group_off(const mi355q_qmd& q, int group_idx) {
  int64_t offset = 0;
  for (int col_idx = 0; col_idx < group_idx; ++col_idx)
    offset += align_to_int64(std::max<int64_t>(q.key_width, 8) * q.entry_count);
  return offset;
}
// getColOffInBytes, output_columnar_ branch (:906-929)
int64_t col_slot_off(const mi355q_qmd& q, int col_idx) {
  int64_t offset = 0;
  if (!q.keyless) offset += col_group_off(q, q.group_col_count);  // getPrependedGroupBufferSizeInBytes
  for (int index = 0; index < col_idx; ++index) offset += align_to_int64((int64_t)q.slot_bytes[index] * q.entry_count);
  return offset;
}
// getBufferSizeBytes (:1084-1111): 8 * group columns * entries + getTotalBytesOfColumnarBuffers
int64_t buffer_bytes(const mi355q_qmd& q) {
  if (!q.output_columnar) return q.entry_count * (int64_t)q.row_size;
  int64_t total = q.keyless ? 0 : (int64_t)sizeof(int64_t) * q.group_col_count * q.entry_count;
  for (int s = 0; s < q.slot_count; ++s) total += align_to_int64((int64_t)q.slot_bytes[s] * q.entry_count);
  return total;
}
// one entry of a columnar buffer <-> the row image the row-wise code works on (key quads, slots)
void col_gather(const mi355q_qmd& q, const int64_t* buf, int64_t e, int64_t* row) {
  const int8_t* b = reinterpret_cast<const int8_t*>(buf);
  const int kq = q.key_bytes / 8;
  for (int k = 0; k < kq; ++k) row[k] = reinterpret_cast<const int64_t*>(b + col_group_off(q, k))[e];
  for (int w = kq; w < q.row_size / 8; ++w) row[w] = 0;
  for (int sl = 0; sl < q.slot_count; ++sl) {
    const int8_t* c = b + col_slot_off(q, sl);
    if (q.slot_width == 8) row[kq + sl] = reinterpret_cast<const int64_t*>(c)[e];
    else reinterpret_cast<int32_t*>(row + kq)[sl] = reinterpret_cast<const int32_t*>(c)[e];
  }
}
void col_scatter_slots(const mi355q_qmd& q, const int64_t* slots, int64_t* buf, int64_t e) {
  int8_t* b = reinterpret_cast<int8_t*>(buf);
  for (int sl = 0; sl < q.slot_count; ++sl) {
    int8_t* c = b + col_slot_off(q, sl);
    if (q.slot_width == 8) reinterpret_cast<int64_t*>(c)[e] = slots[sl];
    else reinterpret_cast<int32_t*>(c)[e] = reinterpret_cast<const int32_t*>(slots)[sl];
  }
}
void col_scatter(const mi355q_qmd& q, const int64_t* row, int64_t* buf, int64_t e) {
  int8_t* b = reinterpret_cast<int8_t*>(buf);
  const int kq = q.key_bytes / 8;
  for (int k = 0; k < kq; ++k) reinterpret_cast<int64_t*>(b + col_group_off(q, k))[e] = row[k];
  col_scatter_slots(q, row + kq, buf, e);
}
// the row-wise descriptor of the same decisions
mi355q_qmd rowwise_of(const mi355q_qmd& q) {
  mi355q_qmd r = q;
  r.output_columnar = 0;
  return r;
}
std::vector<int64_t> col_to_rows(const mi355q_qmd& q, const int64_t* buf) {
  const int rq = q.row_size / 8;
  std::vector<int64_t> rows((size_t)q.entry_count * rq);
  for (int64_t e = 0; e < q.entry_count; ++e) col_gather(q, buf, e, rows.data() + e * rq);
  return rows;
}
// get_columnar_group_bin_offset (GroupByRuntime.cpp:227-239).  The reference emits it for the
// keyless layout as well (GroupByAndAggregate.cpp:1425-1430), where key_base_ptr is the FIRST SLOT's
// column: an entry equal to EMPTY_KEY_64 there is overwritten with the key.
inline uint32_t get_columnar_group_bin_offset(int64_t* key_base_ptr, int64_t key, int64_t min_key, int64_t bucket) {
  int64_t off = key - min_key;
  if (bucket) off /= bucket;
  if (key_base_ptr[off] == kEmptyKey64) key_base_ptr[off] = key;
  return (uint32_t)off;
}
// set_matching_group_value_perfect_hash_columnar (RuntimeFunctions.cpp:2109-2120)
inline void set_matching_group_value_perfect_hash_columnar(int64_t* groups_buffer, uint32_t hashed_index,
                                                           const int64_t* key, uint32_t key_count,
                                                           uint32_t entry_count) {
  if (groups_buffer[hashed_index] == kEmptyKey64) {
    for (uint32_t i = 0; i < key_count; i++) groups_buffer[(size_t)i * entry_count + hashed_index] = key[i];
  }
}
// get_matching_group_value_columnar_slot<int64_t> (RuntimeFunctions.cpp:1994-2017)
inline int32_t get_matching_group_value_columnar_slot(int64_t* groups_buffer, uint32_t entry_count, uint32_t h,
                                                      const int64_t* key, uint32_t key_count) {
  size_t off = h;
  if (groups_buffer[off] == kEmptyKey64) {
    for (size_t i = 0; i < key_count; ++i) {
      groups_buffer[off] = key[i];
      off += entry_count;
    }
    return (int32_t)h;
  }
  off = h;
  for (size_t i = 0; i < key_count; ++i) {
    if (groups_buffer[off] != key[i]) return -1;
    off += entry_count;
  }
  return (int32_t)h;
}
// get_group_value_columnar_slot (GroupByRuntime.cpp:84-105), key_width 8
inline int32_t get_group_value_columnar_slot(int64_t* groups_buffer, uint32_t entry_count, const int64_t* key,
                                             uint32_t key_count) {
  const uint32_t h = murmur3(key, (int)(key_count * sizeof(int64_t)), 0) % entry_count;
  if (get_matching_group_value_columnar_slot(groups_buffer, entry_count, h, key, key_count) != -1) return (int32_t)h;
  uint32_t h_probe = (h + 1) % entry_count;
  while (h_probe != h) {
    if (get_matching_group_value_columnar_slot(groups_buffer, entry_count, h_probe, key, key_count) != -1)
      return (int32_t)h_probe;
    h_probe = (h_probe + 1) % entry_count;
  }
  return -1;
}

// ---------------------------------------------------------------- buffer init
// QueryMemoryInitializer::initRowGroups (QueryMemoryInitializer.cpp:617-698)
void init_buffer(const mi355q_qmd& q, int64_t* buf) {
  const int rq = q.row_size / 8;
  const int kq = q.key_bytes / 8;
  if (q.output_columnar) {
    // initColumnarGroups (QueryMemoryInitializer.cpp:713-780): each key column EMPTY_KEY_64, each
    // slot column its init value at the slot's width, the pointer re-aligned after every column
    int8_t* buffer_ptr = reinterpret_cast<int8_t*>(buf);
    if (!q.keyless) {
      for (int i = 0; i < q.group_col_count; ++i) {
        int64_t* c = reinterpret_cast<int64_t*>(buffer_ptr);
        for (int64_t e = 0; e < q.entry_count; ++e) c[e] = kEmptyKey64;
        buffer_ptr += 8 * q.entry_count;
      }
    }
    // (:738) a Projection's slot columns are NOT initialised; zero here so that the bytes are defined
    if (q.desc_type == MI355Q_PROJECTION) {
      memset(buffer_ptr, 0, (size_t)(buffer_bytes(q) - 8 * q.entry_count));
      return;
    }
    for (int i = 0; i < q.slot_count; ++i) {
      if (q.slot_width == 4) {
        int32_t* c = reinterpret_cast<int32_t*>(buffer_ptr);
        for (int64_t e = 0; e < q.entry_count; ++e) c[e] = (int32_t)q.init_vals[i];
        if (q.entry_count & 1) c[q.entry_count] = 0;  // the padding word (uninitialised in the reference)
      } else {
        int64_t* c = reinterpret_cast<int64_t*>(buffer_ptr);
        for (int64_t e = 0; e < q.entry_count; ++e) c[e] = q.init_vals[i];
      }
      buffer_ptr += align_to_int64((int64_t)q.slot_width * q.entry_count);
    }
    return;
  }
  for (int64_t e = 0; e < q.entry_count; ++e) {
    int64_t* row = buf + e * rq;
    if (kq) {  // result_set::fill_empty_key: every component EMPTY, padding zero
      if (q.key_width == 4) {
        int32_t* k32 = reinterpret_cast<int32_t*>(row);
        for (int i = 0; i < 2 * kq; ++i) k32[i] = i < q.group_col_count ? kEmptyKey32 : 0;
      } else {
        for (int i = 0; i < kq; ++i) row[i] = kEmptyKey64;
      }
    }
    if (q.slot_width == 4) {
      int32_t* s32 = reinterpret_cast<int32_t*>(row + kq);
      for (int s = 0; s < (rq - kq) * 2; ++s) s32[s] = s < q.slot_count ? (int32_t)q.init_vals[s] : 0;
    } else {
      for (int s = 0; s < q.slot_count; ++s) row[kq + s] = q.init_vals[s];
    }
  }
}

// ---------------------------------------------------------------- group slot lookup
// RuntimeFunctions.cpp:1953-1992 get_matching_group_value<T>, single key column
template <typename T>
int64_t* get_matching_group_value_t(int64_t* groups_buffer, uint32_t h, T key,
                                    uint32_t row_size_quad, T empty) {
  int64_t* row = groups_buffer + (size_t)h * row_size_quad;
  T* row_ptr = reinterpret_cast<T*>(row);
  if (*row_ptr == empty) {
    *row_ptr = key;
    return row + 1;  // align_to_int64(row_ptr + 1)
  }
  if (*row_ptr == key) return row + 1;
  return nullptr;
}

// GroupByRuntime.cpp:25-48 get_group_value (+ key_hash :20-23)
int64_t* get_group_value(int64_t* groups_buffer, uint32_t entry_count, int64_t key,
                         uint32_t key_width, uint32_t row_size_quad) {
  uint32_t h;
  if (key_width == 4) {
    int32_t k32 = (int32_t)key;
    h = murmur3(&k32, 4, 0) % entry_count;
    auto m = get_matching_group_value_t<int32_t>(groups_buffer, h, k32, row_size_quad,
                                                 kEmptyKey32);
    if (m) return m;
    uint32_t hp = (h + 1) % entry_count;
    while (hp != h) {
      m = get_matching_group_value_t<int32_t>(groups_buffer, hp, k32, row_size_quad,
                                              kEmptyKey32);
      if (m) return m;
      hp = (hp + 1) % entry_count;
    }
    return nullptr;
  }
  h = murmur3(&key, 8, 0) % entry_count;
  auto m = get_matching_group_value_t<int64_t>(groups_buffer, h, key, row_size_quad,
                                               kEmptyKey64);
  if (m) return m;
  uint32_t hp = (h + 1) % entry_count;
  while (hp != h) {
    m = get_matching_group_value_t<int64_t>(groups_buffer, hp, key, row_size_quad,
                                            kEmptyKey64);
    if (m) return m;
    hp = (hp + 1) % entry_count;
  }
  return nullptr;
}

// RuntimeFunctions.cpp:1953-1992 get_matching_group_value<T>, key_count components:
// an empty first component claims the row (memcpy of the whole key), otherwise memcmp.
template <typename T>
int64_t* get_matching_group_value_n(int64_t* groups_buffer, uint32_t h, const T* key,
                                    uint32_t key_count, uint32_t row_size_quad, T empty) {
  int64_t* row = groups_buffer + (size_t)h * row_size_quad;
  T* row_ptr = reinterpret_cast<T*>(row);
  const size_t key_bytes = key_count * sizeof(T);
  int64_t* slots = row + (key_bytes + 7) / 8;  // align_to_int64(row_ptr + key_count)
  if (*row_ptr == empty) {
    memcpy(row_ptr, key, key_bytes);
    return slots;
  }
  if (memcmp(row_ptr, key, key_bytes) == 0) return slots;
  return nullptr;
}

// GroupByRuntime.cpp:25-48 get_group_value over a key of key_count components of key_width
// bytes: h = MurmurHash3(key, key_count * key_width, 0) % entry_count, linear probing.
// `key` points at the packed components (int32[] or int64[]).
int64_t* get_group_value_n(int64_t* groups_buffer, uint32_t entry_count, const void* key,
                           uint32_t key_count, uint32_t key_width, uint32_t row_size_quad) {
  const uint32_t h = murmur3(key, key_count * key_width, 0) % entry_count;
  auto match = [&](uint32_t hh) -> int64_t* {
    if (key_width == 4) {
      return get_matching_group_value_n<int32_t>(groups_buffer, hh, static_cast<const int32_t*>(key),
                                                 key_count, row_size_quad, kEmptyKey32);
    }
    return get_matching_group_value_n<int64_t>(groups_buffer, hh, static_cast<const int64_t*>(key),
                                               key_count, row_size_quad, kEmptyKey64);
  };
  if (int64_t* m = match(h)) return m;
  uint32_t hp = (h + 1) % entry_count;
  while (hp != h) {
    if (int64_t* m = match(hp)) return m;
    hp = (hp + 1) % entry_count;
  }
  return nullptr;
}

// RuntimeFunctions.cpp:2077-2091 get_matching_group_value_perfect_hash: 64-bit key columns
// prepended to the row, written when the first one is still empty.
inline int64_t* get_matching_group_value_perfect_hash(int64_t* groups_buffer, uint32_t hashed_index,
                                                      const int64_t* key, uint32_t key_count,
                                                      uint32_t row_size_quad) {
  const size_t off = (size_t)hashed_index * row_size_quad;
  if (groups_buffer[off] == kEmptyKey64) {
    for (uint32_t i = 0; i < key_count; ++i) groups_buffer[off + i] = key[i];
  }
  return groups_buffer + off + key_count;
}

// GroupByRuntime.cpp:208-223 get_group_value_fast: the row's key column receives the key it
// was called with — the TRANSLATED key (NULL -> max + 1, GroupByAndAggregate.cpp:1339-1345).
// (:225-241 _with_original_key is only emitted under must_use_baseline_sort.)
inline int64_t* get_group_value_fast(int64_t* buf, int64_t key, int64_t min_key, int64_t bucket,
                                     uint32_t row_size_quad) {
  int64_t key_diff = key - min_key;
  if (bucket) key_diff /= bucket;
  int64_t off = key_diff * row_size_quad;
  if (buf[off] == kEmptyKey64) buf[off] = key;
  return buf + off + 1;
}
// RuntimeFunctions.cpp:2126-2133 get_group_value_fast_keyless
inline int64_t* get_group_value_fast_keyless(int64_t* buf, int64_t key, int64_t min_key,
                                             uint32_t row_size_quad) {
  return buf + row_size_quad * (key - min_key);
}

// ---------------------------------------------------------------- join tables
// Layouts (docs/source/execution/hash_joins.rst "Hash Join Buffers"):
//   0 OneToOne perfect  int32 slot[max-min+1]
//   1 OneToOne keyed    entry_count x (key components..., payload), 4- or 8-byte integers
//   2 OneToMany perfect offsets | counts | payloads (int32 each)
//   3 OneToMany keyed   keys | offsets | counts | payloads
struct OrcJoin {
  int hash_type = 0;
  int64_t min_key = 0, max_key = 0;
  int64_t entry_count = 0;
  int n_keys = 1, width = 8;
  std::vector<int8_t> buf;  // the whole hash join buffer, byte for byte
  const int32_t* perfect() const { return reinterpret_cast<const int32_t*>(buf.data()); }
};

// GroupByRuntime.cpp:287-297 hash_join_idx, :311-318 _nullable
inline int64_t hash_join_idx(const int32_t* buff, int64_t key, int64_t min_key,
                             int64_t max_key) {
  if (key >= min_key && key <= max_key) return buff[key - min_key];
  return -1;
}
// JoinHashTableQueryRuntime.cpp:35-94 baseline_hash_join_idx_impl<T>: slots of
// (key components..., payload); -2 (kNotPresent) at an empty slot, -1 (kNoMatch) on wrap
template <typename T>
int64_t baseline_hash_join_idx(const T* buff, const T* key, int key_count, size_t entry_count,
                               T empty) {
  if (!entry_count) return -1;
  const uint32_t h = murmur1(key, key_count * sizeof(T), 0) % entry_count;
  auto slot = [&](uint32_t hh) -> int64_t {
    const T* e = buff + (size_t)hh * (key_count + 1);
    if (memcmp(e, key, key_count * sizeof(T)) == 0) return e[key_count];
    if (e[0] == empty) return -2;
    return -1;
  };
  int64_t m = slot(h);
  if (m != -1) return m;
  uint32_t hp = (h + 1) % entry_count;
  while (hp != h) {
    m = slot(hp);
    if (m != -1) return m;
    hp = (hp + 1) % entry_count;
  }
  return -1;
}
inline int64_t baseline_hash_join_idx_64(const int64_t* buff, int64_t key, size_t entry_count) {
  return baseline_hash_join_idx<int64_t>(buff, &key, 1, entry_count, kEmptyKey64);
}
// JoinHashTableQueryRuntime.cpp:140-163 get_composite_key_index_impl<T>: index of the key in
// the key dictionary of a one-to-many keyed table, -1 if absent
template <typename T>
int64_t get_composite_key_index(const T* key, size_t key_count, const T* dict, size_t entry_count,
                                T empty) {
  const uint32_t h = murmur1(key, key_count * sizeof(T), 0) % entry_count;
  uint32_t off = h * key_count;
  if (memcmp(&dict[off], key, key_count * sizeof(T)) == 0) return h;
  uint32_t hp = (h + 1) % entry_count;
  while (hp != h) {
    off = hp * key_count;
    if (memcmp(&dict[off], key, key_count * sizeof(T)) == 0) return hp;
    if (dict[off] == empty) return -1;
    hp = (hp + 1) % entry_count;
  }
  return -1;
}

// The matching set of one outer key (HashJoin::codegenMatchingSet, HashJoin.cpp; one-to-one
// tables: codegenSlot): `count` inner row ids at `ids`, or the single id.
struct Matches {
  const int32_t* ids = nullptr;
  int64_t single = -1;
  int32_t count = 0;
};
Matches join_lookup(const OrcJoin& j, const int64_t* keys) {
  Matches m;
  const int64_t n = j.entry_count;
  int32_t k32[MI355Q_MAX_GROUP_COLS];
  for (int i = 0; i < j.n_keys; ++i) k32[i] = (int32_t)keys[i];
  switch (j.hash_type) {
    case 0:
      m.single = hash_join_idx(j.perfect(), keys[0], j.min_key, j.max_key);
      m.count = m.single >= 0;
      break;
    case 1:
      m.single = j.width == 4
                     ? baseline_hash_join_idx<int32_t>(reinterpret_cast<const int32_t*>(j.buf.data()), k32,
                                                       j.n_keys, n, kEmptyKey32)
                     : baseline_hash_join_idx<int64_t>(reinterpret_cast<const int64_t*>(j.buf.data()), keys,
                                                       j.n_keys, n, kEmptyKey64);
      m.count = m.single >= 0;
      break;
    case 2: {
      const int32_t* offsets = j.perfect();
      const int64_t off = hash_join_idx(offsets, keys[0], j.min_key, j.max_key);
      if (off >= 0) {
        m.count = (int32_t)hash_join_idx(offsets + n, keys[0], j.min_key, j.max_key);
        m.ids = offsets + 2 * n + off;
      }
      break;
    }
    default: {
      const int64_t idx =
          j.width == 4 ? get_composite_key_index<int32_t>(k32, j.n_keys,
                                                          reinterpret_cast<const int32_t*>(j.buf.data()), n,
                                                          kEmptyKey32)
                       : get_composite_key_index<int64_t>(keys, j.n_keys,
                                                          reinterpret_cast<const int64_t*>(j.buf.data()), n,
                                                          kEmptyKey64);
      if (idx >= 0) {
        const int32_t* offsets =
            reinterpret_cast<const int32_t*>(j.buf.data() + (size_t)n * j.n_keys * j.width);
        if (offsets[idx] >= 0) {
          m.count = offsets[n + idx];
          m.ids = offsets + 2 * n + offsets[idx];
        }
      }
    }
  }
  return m;
}

// ---------------------------------------------------------------- row function

// ================================================================ projected expressions
// A CPU restatement of what the reference's code generator emits for the expression shapes of the plan ABI
// (mi355q_expr: casts and + - * over columns and literals); the product evaluates them in a projection pass
// (heavydb_amd/csrc/expr.h), this file per row inside the row function, as the reference does.
//   typing       the analyzer has already given both operands of a BinOper one type (ArithmeticIR.cpp:61
//                CHECK_EQ(lhs_type.get_type(), rhs_type.get_type())); a ColumnVar has the column's SQL type
//   casts        CodeGenerator::codegenCast (CastIR.cpp:71-135): integer operands go to
//                codegenCastBetweenIntTypes (:424-495) or codegenCastToFp (:555-594), floating-point ones to
//                codegenCastFromFp (:596-653); nullable operands call cast_<from>_to_<to>_nullable
//                (RuntimeFunctions.cpp:262-268), fp -> integer the rounding form (:283-293); a NARROWING
//                integer cast is preceded by codegenCastBetweenIntTypesOverflowChecks (:497-553):
//                over = v > max(to), under = v <= min(to) (sic), a NULL operand is exempt
//   + - *        CodeGenerator::codegenArith (ArithmeticIR.cpp:39-75).  Integers, CPU device:
//                codegenBinOpWithOverflowForCPU (:861-909) = llvm.s{add,sub,mul}.with.overflow at the operand
//                type's width, error ErrorCode::OVERFLOW_OR_UNDERFLOW (enums.h) on overflow; with a nullable
//                operand codegenSkipOverflowCheckForNull jumps past the check and the result is the type's
//                NULL.  Floating point: plain fadd / fsub / fmul, or add_/sub_/mul_<type>_nullable[_lhs|_rhs]
//                (RuntimeFunctions.cpp:46-71) — NULL if a nullable operand equals the NULL value.
//   where        target and group-by expressions are emitted inside the filter's true branch (after the join
//                loop found a match); an expression inside a qual runs for every row.
struct OrcVal {
  int type;      // mi355q_type
  bool nullable; // get_notnull() == false
  int64_t i;     // integers (sign-extended)
  double d;      // DOUBLE
  float f;       // FLOAT
};

inline int64_t int_max_of_type(int t) {
  switch (t) {
    case MI355Q_INT8: return INT8_MAX;
    case MI355Q_INT16: return INT16_MAX;
    case MI355Q_INT32: return INT32_MAX;
    default: return INT64_MAX;
  }
}
inline bool is_int_type(int t) { return t >= MI355Q_INT8 && t <= MI355Q_INT64; }
inline int int_bytes(int t) { return t == MI355Q_INT8 ? 1 : t == MI355Q_INT16 ? 2 : t == MI355Q_INT32 ? 4 : 8; }
inline bool val_is_null(const OrcVal& v) {
  if (!v.nullable) return false;
  if (v.type == MI355Q_DOUBLE) return v.d == kNullDouble;
  if (v.type == MI355Q_FLOAT) return v.f == kNullFloat;
  return v.i == int_null_of(v.type);
}
inline OrcVal null_of(int type) {
  OrcVal r{type, true, 0, 0.0, 0.0f};
  if (type == MI355Q_DOUBLE) r.d = kNullDouble;
  else if (type == MI355Q_FLOAT) r.f = kNullFloat;
  else r.i = int_null_of(type);
  return r;
}

// codegenCast for one value; returns 0 or ErrorCode 7
inline int32_t cast_value(const OrcVal& v, int to, OrcVal* out) {
  OrcVal r{to, v.nullable, 0, 0.0, 0.0f};
  if (is_int_type(v.type)) {
    if (is_int_type(to)) {
      if (int_bytes(to) < int_bytes(v.type) && !val_is_null(v)) {  // narrowing: the overflow checks
        if (v.i > int_max_of_type(to) || v.i <= int_null_of(to)) return MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
      }
      r.i = val_is_null(v) ? int_null_of(to) : v.i;  // cast_<from>_to_<to>_nullable / sext / trunc
    } else if (to == MI355Q_DOUBLE) {
      r.d = val_is_null(v) ? kNullDouble : (double)v.i;  // sitofp
    } else {
      r.f = val_is_null(v) ? kNullFloat : (float)v.i;
    }
  } else if (v.type == MI355Q_DOUBLE) {
    if (to == MI355Q_DOUBLE) r.d = v.d;
    else if (to == MI355Q_FLOAT) r.f = val_is_null(v) ? kNullFloat : (float)v.d;  // fptrunc
    else r.i = val_is_null(v) ? int_null_of(to) : (int64_t)(v.d + (v.d < 0.0 ? -0.5 : 0.5));
  } else {
    if (to == MI355Q_FLOAT) r.f = v.f;
    else if (to == MI355Q_DOUBLE) r.d = val_is_null(v) ? kNullDouble : (double)v.f;  // fpext
    else r.i = val_is_null(v) ? int_null_of(to) : (int64_t)(v.f + (v.f < 0.0f ? -0.5f : 0.5f));
  }
  *out = r;
  return 0;
}

// codegenArith for one pair of values of type `t`
inline int32_t arith_value(int op, int t, const OrcVal& a, const OrcVal& b, OrcVal* out) {
  OrcVal r{t, a.nullable || b.nullable, 0, 0.0, 0.0f};
  if (op == MI355Q_EX_DIV || op == MI355Q_EX_MOD) {
    // codegenDiv (ArithmeticIR.cpp:431-560, g_null_div_by_zero off): with a nullable operand, a NULL PATTERN in either operand
    // skips the zero check (codegenSkipOverflowCheckForNull :343-357 tests both against the type's NULL); codegenMod
    // (:731-760) tests the divisor first, whatever the NULLs.  Then div_/mod_<type>_nullable[_lhs|_rhs]
    // (RuntimeFunctions.cpp:46-71): NULL when a NULLABLE operand is NULL, else lhs op rhs.
    const bool any_nullable = a.nullable || b.nullable;
    auto pattern_null = [&](const OrcVal& v) {
      return is_int_type(t) ? v.i == int_null_of(t) : t == MI355Q_DOUBLE ? v.d == kNullDouble : v.f == kNullFloat;
    };
    const bool skip = op == MI355Q_EX_DIV && any_nullable && (pattern_null(a) || pattern_null(b));
    const bool zero = is_int_type(t) ? b.i == 0 : t == MI355Q_DOUBLE ? !(b.d < 0.0 || b.d > 0.0) : !(b.f < 0.0f || b.f > 0.0f);
    if (op == MI355Q_EX_MOD && !is_int_type(t)) return MI355Q_ERR_INVALID_PLAN;
    if (!skip && zero) return MI355Q_ERR_DIV_BY_ZERO;
    if (val_is_null(a) || val_is_null(b)) {
      *out = null_of(t);
      out->nullable = r.nullable;
      return 0;
    }
    if (is_int_type(t)) {
      int64_t w;
      if (b.i == 0) w = int_null_of(t);                        // (undefined in the reference: INT_MIN / 0 behind the skip)
      else if (b.i == -1) w = op == MI355Q_EX_DIV ? (int64_t)(0 - (uint64_t)a.i) : 0;   // INT_MIN / -1: wraps here, traps there
      else w = op == MI355Q_EX_DIV ? a.i / b.i : a.i % b.i;
      r.i = t == MI355Q_INT8 ? (int64_t)(int8_t)w : t == MI355Q_INT16 ? (int64_t)(int16_t)w : t == MI355Q_INT32 ? (int64_t)(int32_t)w : w;
    } else if (t == MI355Q_DOUBLE) {
      r.d = a.d / b.d;
    } else {
      r.f = a.f / b.f;
    }
    *out = r;
    return 0;
  }
  if (val_is_null(a) || val_is_null(b)) {
    *out = null_of(t);
    out->nullable = r.nullable;
    return 0;
  }
  if (is_int_type(t)) {
    __int128 w = op == MI355Q_EX_ADD ? (__int128)a.i + b.i : op == MI355Q_EX_SUB ? (__int128)a.i - b.i : (__int128)a.i * b.i;
    if (w > (__int128)int_max_of_type(t) || w < (__int128)int_null_of(t)) return MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
    r.i = (int64_t)w;
  } else if (t == MI355Q_DOUBLE) {
    r.d = op == MI355Q_EX_ADD ? a.d + b.d : op == MI355Q_EX_SUB ? a.d - b.d : a.d * b.d;
  } else {
    r.f = op == MI355Q_EX_ADD ? a.f + b.f : op == MI355Q_EX_SUB ? a.f - b.f : a.f * b.f;
  }
  *out = r;
  return 0;
}

// codegenCmp for one pair of values of one type (CompareIR.cpp:230-330): icmp / fcmp when neither operand can be NULL, else
// <op>_<type>_nullable[_lhs|_rhs] (RuntimeFunctions.cpp:73-107) with the BOOLEAN NULL (the int8 sentinel) as null_bool_val
inline OrcVal cmp_value(int op, const OrcVal& a, const OrcVal& b) {
  OrcVal r{MI355Q_INT8, a.nullable || b.nullable, 0, 0.0, 0.0f};
  if (val_is_null(a) || val_is_null(b)) {
    r.i = int_null_of(MI355Q_INT8);
    return r;
  }
  auto pick = [&](auto x, auto y) -> bool {
    switch (op) {
      case MI355Q_EX_EQ: return x == y;
      case MI355Q_EX_NE: return x != y;
      case MI355Q_EX_LT: return x < y;
      case MI355Q_EX_LE: return x <= y;
      case MI355Q_EX_GT: return x > y;
      default: return x >= y;
    }
  };
  r.i = is_int_type(a.type) ? pick(a.i, b.i) : a.type == MI355Q_DOUBLE ? pick(a.d, b.d) : pick(a.f, b.f);
  return r;
}

// one expression on one row; `p` is the plan as the caller stated it (physical columns only).  The postfix program is
// evaluated from its ROOT the way the generated code runs: the operands of an operation left to right, a CASE's condition
// and then ONLY the branch it selects (codegenCase, CaseIR.cpp:67-140: each branch is a basic block of its own, so a check
// in the other branch is never executed).  Operand order of MI355Q_EX_CASE on the stack: ELSE, THEN, condition.
struct ExprWalk {
  const mi355q_plan& p;
  const mi355q_expr& x;
  const int8_t* const* cols;
  int64_t pos;
  // the LOWERED plan (lower_plan): describes column n_cols + j, the value of an earlier expression of the plan — which the
  // row function has already evaluated for this row and which `cols` holds like a plain column; nullptr: physical columns only
  const mi355q_plan* low = nullptr;
  const mi355q_col_desc& desc_of(int c) const { return c < p.n_cols || !low ? p.cols[c] : low->cols[c]; }
  // the node where the subtree that ENDS at node `end` starts
  int start_of(int end) const {
    int need = 1, i = end;
    for (;; --i) {
      const int op = x.nodes[i].op;
      const int arity = op == MI355Q_EX_COL || op == MI355Q_EX_LIT ? 0
                        : op == MI355Q_EX_CAST || op == MI355Q_EX_NOT || op == MI355Q_EX_IS_NULL || op == MI355Q_EX_UMINUS ? 1
                        : op == MI355Q_EX_CASE ? 3 : 2;
      need += arity - 1;
      if (need == 0) return i;
    }
  }
  // get_notnull() == false of the subtree's type
  bool may_be_null(int end) const {
    const mi355q_expr_node& n = x.nodes[end];
    if (n.op == MI355Q_EX_COL) return desc_of(n.arg).nullable != 0;
    if (n.op == MI355Q_EX_LIT) return n.reserved == 1;
    if (n.op == MI355Q_EX_CAST || n.op == MI355Q_EX_NOT || n.op == MI355Q_EX_UMINUS) return may_be_null(end - 1);
    if (n.op == MI355Q_EX_IS_NULL) return false;  // (a NOT NULL BOOLEAN)
    const int last = end - 1, mid = start_of(last) - 1;
    if (n.op == MI355Q_EX_CASE) return may_be_null(mid) || may_be_null(start_of(mid) - 1);
    return may_be_null(last) || may_be_null(mid);
  }
  int32_t eval(int end, OrcVal* out) const {
    const mi355q_expr_node& n = x.nodes[end];
    switch (n.op) {
      case MI355Q_EX_COL: {
        if (n.arg >= p.n_cols && !low) return MI355Q_ERR_INVALID_PLAN;
        const mi355q_col_desc& cd = desc_of(n.arg);
        OrcVal v{logical_type_of(cd), cd.nullable != 0, 0, 0.0, 0.0f};
        if (type_is_f32(cd.type)) v.f = decode_flt(cols[n.arg], pos);
        else if (type_is_fp(cd.type)) v.d = decode_dbl(cols[n.arg], pos);
        else v.i = decode_col(cd, cols[n.arg], pos);
        *out = v;
        return 0;
      }
      case MI355Q_EX_LIT: {
        if (n.reserved == 1) {  // the NULL constant (a CASE without ELSE)
          *out = null_of(n.type);
          return 0;
        }
        OrcVal v{n.type, false, 0, 0.0, 0.0f};
        if (n.type == MI355Q_DOUBLE) v.d = n.flit;
        else if (n.type == MI355Q_FLOAT) v.f = (float)n.flit;
        else v.i = n.ilit;
        *out = v;
        return 0;
      }
      case MI355Q_EX_CAST: {
        OrcVal v;
        if (int32_t e = eval(end - 1, &v)) return e;
        return cast_value(v, n.type, out);
      }
      case MI355Q_EX_NOT: {  // codegenLogical(UOper), LogicalIR.cpp:363-379
        OrcVal v;
        if (int32_t e = eval(end - 1, &v)) return e;
        OrcVal r{MI355Q_INT8, v.nullable, 0, 0.0, 0.0f};
        if (v.nullable) r.i = v.i == int_null_of(MI355Q_INT8) ? v.i : (v.i ? 0 : 1);  // logical_not, RuntimeFunctions.cpp:331-334
        else r.i = v.i > 0 ? 0 : 1;                                                  // CreateNot(toBool)
        *out = r;
        return 0;
      }
      case MI355Q_EX_AND:
      case MI355Q_EX_OR: {
        const int rhs_end = end - 1, lhs_end = start_of(rhs_end) - 1;
        const bool is_or = n.op == MI355Q_EX_OR;
        const bool nullable = may_be_null(lhs_end) || may_be_null(rhs_end);  // the BinOper's type
        const int64_t nul = int_null_of(MI355Q_INT8);
        OrcVal a, b;
        OrcVal r{MI355Q_INT8, nullable, 0, 0.0, 0.0f};
        if (int32_t e = eval(lhs_end, &a)) return e;
        if (n.reserved == 1) {
          // codegenLogicalShortCircuit (LogicalIR.cpp:197-297): nullcheck of the first operand, then `first != (op == kOR)`
          // branches to the block that evaluates the second operand; the phi takes NULL / the constant / the second value
          if (nullable && a.i == nul) {
            r.i = nul;
          } else if (a.i == (is_or ? 1 : 0)) {
            r.i = a.i;
          } else {
            if (int32_t e = eval(rhs_end, &b)) return e;
            r.i = b.i;  // (NULL where it is the NULL pattern: the nullcheck_fail block feeds the same value)
          }
          *out = r;
          return 0;
        }
        if (int32_t e = eval(rhs_end, &b)) return e;
        if (!nullable) {
          r.i = is_or ? (a.i > 0 || b.i > 0) : (a.i > 0 && b.i > 0);  // toBool(lhs) op toBool(rhs), :312-320
        } else if (is_or) {  // logical_or, RuntimeFunctions.cpp:348-358
          if (a.i == nul) r.i = b.i == 0 ? nul : b.i;
          else if (b.i == nul) r.i = a.i == 0 ? nul : a.i;
          else r.i = (a.i || b.i) ? 1 : 0;
        } else {  // logical_and, :336-346
          if (a.i == nul) r.i = b.i == 0 ? b.i : nul;
          else if (b.i == nul) r.i = a.i == 0 ? a.i : nul;
          else r.i = (a.i && b.i) ? 1 : 0;
        }
        *out = r;
        return 0;
      }
      case MI355Q_EX_IS_NULL: {  // codegenIsNull, LogicalIR.cpp:381-432
        OrcVal r{MI355Q_INT8, false, 0, 0.0, 0.0f};
        if (may_be_null(end - 1)) {  // (a NOT NULL operand is not evaluated: constant false)
          OrcVal v;
          if (int32_t e = eval(end - 1, &v)) return e;
          r.i = val_is_null(v) ? 1 : 0;
        }
        *out = r;
        return 0;
      }
      case MI355Q_EX_UMINUS: {  // codegenUMinus, ArithmeticIR.cpp:787-838
        OrcVal v;
        if (int32_t e = eval(end - 1, &v)) return e;
        if (val_is_null(v)) {  // uminus_<type>_nullable: NULL stays NULL (and the check is skipped)
          *out = v;
          return 0;
        }
        if (is_int_type(v.type)) {
          if (v.i == int_null_of(v.type)) return MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;  // operand == the type's minimum
          v.i = -v.i;
        } else if (v.type == MI355Q_DOUBLE) {
          v.d = -v.d;
        } else {
          v.f = -v.f;
        }
        *out = v;
        return 0;
      }
      case MI355Q_EX_CASE: {
        const int cond_end = end - 1, then_end = start_of(cond_end) - 1, else_end = start_of(then_end) - 1;
        OrcVal c;
        if (int32_t e = eval(cond_end, &c)) return e;
        const bool take = !val_is_null(c) && c.i != 0;  // toBool
        OrcVal v;
        if (int32_t e = eval(take ? then_end : else_end, &v)) return e;
        v.nullable = may_be_null(then_end) || may_be_null(else_end);  // the CASE's type: nullable as soon as one branch is
        v.type = n.type;
        *out = v;
        return 0;
      }
      default: {
        const int rhs_end = end - 1, lhs_end = start_of(rhs_end) - 1;
        OrcVal a, b;
        if (int32_t e = eval(lhs_end, &a)) return e;
        if (int32_t e = eval(rhs_end, &b)) return e;
        if (n.op >= MI355Q_EX_EQ && n.op <= MI355Q_EX_GE) {
          *out = cmp_value(n.op, a, b);
          return 0;
        }
        return arith_value(n.op, n.type, a, b, out);
      }
    }
  }
};
inline int32_t eval_expression(const mi355q_plan& p, const mi355q_expr& x, const int8_t* const* cols, int64_t pos,
                               OrcVal* out, const mi355q_plan* lowered = nullptr) {
  return ExprWalk{p, x, cols, pos, lowered}.eval(x.n_nodes - 1, out);
}

// The plan with every expression described as the column n_cols + k (type / nullability from the rules
// above, range as stated by the caller), which is what the layout decisions and the row function read.
inline int lower_plan(const mi355q_plan& p, mi355q_plan* out) {
  *out = p;
  if (p.n_exprs == 0) return 0;
  if (p.n_exprs < 0 || p.n_exprs > MI355Q_MAX_EXPRS || p.n_cols + p.n_exprs > MI355Q_MAX_COLS) return MI355Q_ERR_INVALID_PLAN;
  for (int k = 0; k < p.n_exprs; ++k) {
    const mi355q_expr& x = p.exprs[k];
    if (x.n_nodes < 1 || x.n_nodes > MI355Q_MAX_EXPR_NODES) return MI355Q_ERR_INVALID_PLAN;
    int ty[MI355Q_MAX_EXPR_NODES];
    bool nu[MI355Q_MAX_EXPR_NODES];
    int sp = 0;
    for (int i = 0; i < x.n_nodes; ++i) {
      const mi355q_expr_node& n = x.nodes[i];
      if (n.op == MI355Q_EX_COL) {
        // a physical column, or the value of an earlier expression (column n_cols + j, j < k: described just above)
        if (n.arg < 0 || n.arg >= p.n_cols + k) return MI355Q_ERR_INVALID_PLAN;
        const mi355q_col_desc& cd = n.arg < p.n_cols ? p.cols[n.arg] : out->cols[n.arg];
        ty[sp] = logical_type_of(cd);
        nu[sp] = cd.nullable != 0;
        ++sp;
      } else if (n.op == MI355Q_EX_LIT) {
        ty[sp] = n.type;
        nu[sp] = n.reserved == 1;  // the NULL constant
        ++sp;
      } else if (n.op == MI355Q_EX_CAST) {
        if (sp < 1) return MI355Q_ERR_INVALID_PLAN;
        ty[sp - 1] = n.type;
      } else if (n.op == MI355Q_EX_ADD || n.op == MI355Q_EX_SUB || n.op == MI355Q_EX_MUL || n.op == MI355Q_EX_DIV ||
                 n.op == MI355Q_EX_MOD) {
        if (sp < 2 || ty[sp - 1] != n.type || ty[sp - 2] != n.type) return MI355Q_ERR_INVALID_PLAN;
        if (n.op == MI355Q_EX_MOD && !is_int_type(n.type)) return MI355Q_ERR_INVALID_PLAN;
        nu[sp - 2] = nu[sp - 2] || nu[sp - 1];
        --sp;
      } else if (n.op >= MI355Q_EX_EQ && n.op <= MI355Q_EX_GE) {
        if (sp < 2 || ty[sp - 1] != ty[sp - 2] || n.type != MI355Q_INT8) return MI355Q_ERR_INVALID_PLAN;
        nu[sp - 2] = nu[sp - 2] || nu[sp - 1];
        ty[sp - 2] = MI355Q_INT8;
        --sp;
      } else if (n.op == MI355Q_EX_CASE) {  // ELSE, THEN, condition
        if (sp < 3 || ty[sp - 1] != MI355Q_INT8 || ty[sp - 2] != n.type || ty[sp - 3] != n.type) return MI355Q_ERR_INVALID_PLAN;
        nu[sp - 3] = nu[sp - 3] || nu[sp - 2];
        sp -= 2;
      } else if (n.op == MI355Q_EX_NOT) {
        if (sp < 1 || ty[sp - 1] != MI355Q_INT8 || n.type != MI355Q_INT8) return MI355Q_ERR_INVALID_PLAN;
      } else if (n.op == MI355Q_EX_AND || n.op == MI355Q_EX_OR) {
        if (sp < 2 || ty[sp - 1] != MI355Q_INT8 || ty[sp - 2] != MI355Q_INT8 || n.type != MI355Q_INT8 ||
            (n.reserved != 0 && n.reserved != 1))
          return MI355Q_ERR_INVALID_PLAN;
        nu[sp - 2] = nu[sp - 2] || nu[sp - 1];
        --sp;
      } else if (n.op == MI355Q_EX_IS_NULL) {
        if (sp < 1 || n.type != MI355Q_INT8) return MI355Q_ERR_INVALID_PLAN;
        ty[sp - 1] = MI355Q_INT8;
        nu[sp - 1] = false;
      } else if (n.op == MI355Q_EX_UMINUS) {
        if (sp < 1 || ty[sp - 1] != n.type) return MI355Q_ERR_INVALID_PLAN;
      } else {
        return MI355Q_ERR_UNSUPPORTED;
      }
      if (sp > MI355Q_MAX_EXPR_STACK) return MI355Q_ERR_INVALID_PLAN;
    }
    if (sp != 1) return MI355Q_ERR_INVALID_PLAN;
    out->cols[p.n_cols + k] = mi355q_col_desc{ty[0], nu[0] ? 1 : 0, MI355Q_ENC_NONE, 0};
    out->col_ranges[p.n_cols + k] = x.range;
  }
  out->n_cols = p.n_cols + p.n_exprs;
  out->n_exprs = 0;
  return 0;
}

struct ExecCtx {
  const mi355q_plan* plan;   // the LOWERED plan when the caller's has expressions (storage: `lowered`)
  const mi355q_plan* stated = nullptr;  // the caller's plan (expression programs, physical columns)
  mi355q_plan lowered;
  mi355q_qmd qmd;
  std::vector<TargetDesc> ts;
  const OrcJoin* join;
  const int8_t* const* inner_cols;
  int64_t inner_rows;
  int32_t* total_matched = nullptr;  // Projection: the kernel's total_matched word (KernelParam::TOTAL_MATCHED)
};

// get_scan_output_slot / get_columnar_scan_output_offset (GroupByRuntime.cpp:242-269), as written there
inline int64_t* get_scan_output_slot(int64_t* output_buffer, const uint32_t output_buffer_entry_count, const uint32_t pos,
                                     const int64_t offset_in_fragment, const uint32_t row_size_quad) {
  uint64_t off = static_cast<uint64_t>(pos) * static_cast<uint64_t>(row_size_quad);
  if (pos < output_buffer_entry_count) {
    output_buffer[off] = offset_in_fragment;
    return output_buffer + off + 1;
  }
  return NULL;
}
inline int32_t get_columnar_scan_output_offset(int64_t* output_buffer, const uint32_t output_buffer_entry_count,
                                               const uint32_t pos, const int64_t offset_in_fragment) {
  if (pos < output_buffer_entry_count) {
    output_buffer[pos] = offset_in_fragment;
    return pos;
  }
  return -1;
}

// DEF_CMP_NULLABLE (RuntimeFunctions.cpp:73-83) + toBool (>0): NULL operand -> false
inline bool eval_qual(const mi355q_plan& p, const mi355q_qual& q_in, const int8_t* const* cols,
                      int64_t pos) {
  mi355q_qual q = q_in;
  q.op = MI355Q_QUAL_OP(q_in.op);  // (the disjunction a qual belongs to travels in the upper bits)
  const auto& cd = p.cols[q.col];
  if (q.op == MI355Q_IS_NULL || q.op == MI355Q_IS_NOT_NULL) {
    // codegenIsNull (LogicalIR.cpp:381-432): constant false on a NOT NULL type, otherwise the value
    // equals the type's inline NULL (FCMP_OEQ for floating point); IS NOT NULL = NOT(IS NULL)
    bool is_null = false;
    if (cd.nullable) {
      if (type_is_f32(cd.type)) is_null = decode_flt(cols[q.col], pos) == kNullFloat;
      else if (type_is_fp(cd.type)) is_null = decode_dbl(cols[q.col], pos) == kNullDouble;
      else is_null = decode_col(cd, cols[q.col], pos) == int_null_of(logical_type_of(cd));
    }
    return q.op == MI355Q_IS_NULL ? is_null : !is_null;
  }
  if (type_is_f32(cd.type)) {  // lt_float_nullable etc. (DEF_CMP_NULLABLE for float): single precision
    const float v = decode_flt(cols[q.col], pos);
    const float lit = (float)q.fval;
    if (cd.nullable && v == kNullFloat) return false;
    switch (q.op) {
      case MI355Q_EQ: return v == lit;
      case MI355Q_NE: return v != lit;
      case MI355Q_LT: return v < lit;
      case MI355Q_GT: return v > lit;
      case MI355Q_LE: return v <= lit;
      case MI355Q_GE: return v >= lit;
    }
    return false;
  }
  if (type_is_fp(cd.type)) {
    const double v = decode_dbl(cols[q.col], pos);
    if (cd.nullable && v == kNullDouble) return false;
    switch (q.op) {
      case MI355Q_EQ: return v == q.fval;
      case MI355Q_NE: return v != q.fval;
      case MI355Q_LT: return v < q.fval;
      case MI355Q_GT: return v > q.fval;
      case MI355Q_LE: return v <= q.fval;
      case MI355Q_GE: return v >= q.fval;
    }
    return false;
  }
  const int64_t v = decode_col(cd, cols[q.col], pos);
  if (cd.nullable && v == int_null_of(logical_type_of(cd))) return false;
  switch (q.op) {
    case MI355Q_EQ: return v == q.ival;
    case MI355Q_NE: return v != q.ival;
    case MI355Q_LT: return v < q.ival;
    case MI355Q_GT: return v > q.ival;
    case MI355Q_LE: return v <= q.ival;
    case MI355Q_GE: return v >= q.ival;
  }
  return false;
}

// One target update into its slot(s): the agg_* call TargetExprCodegen::codegenAggregate
// (TargetExprBuilder.cpp:470-590) would emit for 8-byte slots.
inline void apply_target(const mi355q_plan& p_, const TargetDesc& t, int64_t* slots,
                         const int8_t* const* cols, int64_t pos, const int8_t* const* inner_cols,
                         int64_t inner_pos, const int64_t* key_vals) {
  int64_t* s = slots + t.slot;
  if (t.agg == MI355Q_PROJECT_KEY) {
    if (t.slot >= 0) *s = key_vals[t.key_idx];  // agg_id (RuntimeFunctions.cpp:1171)
    return;
  }
  if (t.agg == MI355Q_COUNT_IF || t.agg == MI355Q_SUM_IF) {
    // agg_count_if[_skip_val] (RuntimeFunctions.cpp:1356-1375): counted when the condition is
    // neither NULL nor 0; agg_sum_if* (:1157-1161,1341-1346,1450-1456) with the i8 condition of
    // codegenConditionalAggregateCondValSelector (WindowFunctionIR.cpp:1610-1640): == 1
    if (!eval_qual(p_, *t.cond, cols, pos)) return;
    if (t.agg == MI355Q_COUNT_IF) {
      agg_count(s);
      return;
    }
  }
  if (t.col < 0) {  // COUNT(*)
    agg_count(s);
    return;
  }
  // outer join without a match: the inner column value is the NULL placeholder
  // (codegenOuterJoinNullPlaceholder); inner columns are nullable under an outer join, so the
  // _skip_val aggregate leaves the slot untouched
  if (t.table && inner_pos < 0) return;
  const int8_t* col = t.table ? inner_cols[t.col] : cols[t.col];
  const int64_t p = t.table ? inner_pos : pos;
  if (t.arg_f32) {
    // takes_float_argument: agg_chosen_bytes = sizeof(float) (TargetExprBuilder.cpp:477-481);
    // the COUNT component of AVG and COUNT(col) itself stay 8 bytes wide
    const float v = decode_flt(col, p);
    int32_t* s32 = reinterpret_cast<int32_t*>(s);
    switch (t.agg) {
      case MI355Q_COUNT:
        if (!t.skip_null || v != kNullFloat) agg_count(s);
        break;
      case MI355Q_SUM:
      case MI355Q_SUM_IF:
        if (t.skip_null) agg_sum_float_skip_val(s32, v, kNullFloat);
        else agg_sum_float(s32, v);
        break;
      case MI355Q_AVG:
        if (t.skip_null) {
          agg_sum_float_skip_val(s32, v, kNullFloat);
          if (v != kNullFloat) agg_count(s + 1);
        } else {
          agg_sum_float(s32, v);
          agg_count(s + 1);
        }
        break;
      case MI355Q_MIN:
        if (t.skip_null) agg_min_float_skip_val(s32, v, kNullFloat);
        else agg_min_float(s32, v);
        break;
      case MI355Q_MAX:
        if (t.skip_null) agg_max_float_skip_val(s32, v, kNullFloat);
        else agg_max_float(s32, v);
        break;
    }
    return;
  }
  if (t.arg_fp) {
    const double v = decode_dbl(col, p);
    switch (t.agg) {
      case MI355Q_COUNT:
        if (t.skip_null) {
          if (v != kNullDouble) agg_count(s);  // agg_count_double_skip_val :1542
        } else {
          agg_count(s);
        }
        break;
      case MI355Q_SUM:
      case MI355Q_SUM_IF:
        if (t.skip_null) agg_sum_double_skip_val(s, v, kNullDouble);
        else agg_sum_double(s, v);
        break;
      case MI355Q_AVG:
        if (t.skip_null) {
          agg_sum_double_skip_val(s, v, kNullDouble);
          if (v != kNullDouble) agg_count(s + 1);
        } else {
          agg_sum_double(s, v);
          agg_count(s + 1);
        }
        break;
      case MI355Q_MIN:
        if (t.skip_null) agg_min_double_skip_val(s, v, kNullDouble);
        else agg_min_double(s, v);
        break;
      case MI355Q_MAX:
        if (t.skip_null) agg_max_double_skip_val(s, v, kNullDouble);
        else agg_max_double(s, v);
        break;
    }
    return;
  }
  const int64_t raw = decode_col(*t.cd, col, p);
  const int64_t null_t = int_null_of(t.arg_type);
  switch (t.agg) {
    case MI355Q_COUNT:
      if (t.skip_null) {
        if (raw != null_t) agg_count(s);  // agg_count_skip_val :1361
      } else {
        agg_count(s);
      }
      break;
    case MI355Q_SUM:
    case MI355Q_SUM_IF:
    case MI355Q_AVG: {
      if (t.skip_null) {
        // convertNullIfAny: arg NULL -> NULL of the BIGINT sum type
        const int64_t v = (raw == null_t) ? INT64_MIN : raw;
        agg_sum_skip_val(s, v, INT64_MIN);
        if (t.agg == MI355Q_AVG && v != INT64_MIN) agg_count(s + 1);
      } else {
        agg_sum(s, raw);
        if (t.agg == MI355Q_AVG) agg_count(s + 1);
      }
      break;
    }
    case MI355Q_MIN:
      // is_agg_domain_range_equivalent: skip value is the ARG type's null, sign-extended
      if (t.skip_null) agg_min_skip_val(s, raw, null_t);
      else agg_min(s, raw);
      break;
    case MI355Q_MAX:
      if (t.skip_null) agg_max_skip_val(s, raw, null_t);
      else agg_max(s, raw);
      break;
  }
}

// row_func + the loop of query_template / query_group_by_template
// (QueryTemplateGenerator.cpp:265-549, :553-814) with pos_start = 0, pos_step = 1
// (RuntimeFunctions.cpp:1835-1846).  Returns 0 or a HeavyDB error code.
int32_t run_fragment(const ExecCtx& c, const int8_t* const* cols, int64_t num_rows,
                     int64_t* buf) {
  const auto& p = *c.plan;
  const auto& q = c.qmd;
  const int rq = q.row_size / 8;
  const int kq = q.key_bytes / 8;
  const bool grouped = q.desc_type != MI355Q_NON_GROUPED_AGGREGATE;
  const int ng = q.group_col_count;
  // projected expressions: each row's values live in one 8-byte cell per expression, and the column table
  // the rest of the row function reads is extended by pointers biased so that "column[pos]" is that cell
  const mi355q_plan& sp = *c.stated;
  const int nx = sp.n_exprs, np = sp.n_cols;
  const int8_t* cx[MI355Q_MAX_COLS];
  int64_t vcell[MI355Q_MAX_EXPRS] = {0, 0, 0, 0};
  uint32_t qual_exprs = 0;
  if (nx) {
    for (int i = 0; i < np; ++i) cx[i] = cols[i];
    cols = cx;
    for (int i = 0; i < p.n_quals; ++i)
      if (p.quals[i].col >= np) qual_exprs |= 1u << (p.quals[i].col - np);
    // ... and the earlier expressions those read (a filter is evaluated before anything else of the row)
    for (int k = nx - 1; k >= 0; --k)
      if (qual_exprs & (1u << k))
        for (int i = 0; i < sp.exprs[k].n_nodes; ++i)
          if (sp.exprs[k].nodes[i].op == MI355Q_EX_COL && sp.exprs[k].nodes[i].arg >= np)
            qual_exprs |= 1u << (sp.exprs[k].nodes[i].arg - np);
  }
  auto eval_into_cell = [&](int k, int64_t pos) -> int32_t {
    OrcVal v;
    if (int32_t e = eval_expression(sp, sp.exprs[k], cx, pos, &v, &p)) return e;
    const int w = type_width(v.type);
    if (v.type == MI355Q_DOUBLE) vcell[k] = dbl_bits(v.d);
    else if (v.type == MI355Q_FLOAT) vcell[k] = (int64_t)(uint32_t)flt_bits(v.f);
    else vcell[k] = v.i;
    cx[np + k] = reinterpret_cast<const int8_t*>(reinterpret_cast<uintptr_t>(&vcell[k]) - (uintptr_t)pos * (uintptr_t)w);
    return 0;
  };
  for (int64_t pos = 0; pos < num_rows; ++pos) {
    for (int k = 0; k < nx; ++k)  // expressions inside quals: every row
      if (qual_exprs & (1u << k))
        if (int32_t e = eval_into_cell(k, pos)) return e;
    // codegenLogical (LogicalIR.cpp:299-340) + toBool (:344-352): the condition must be TRUE — every plain conjunct, and
    // of every disjunction (quals sharing a group number) at least one member; NULL is not TRUE
    bool pass = true;
    uint32_t or_seen = 0, or_any = 0;
    for (int i = 0; i < p.n_quals && pass; ++i) {
      const int g = MI355Q_QUAL_OR_GROUP(p.quals[i].op);
      const bool t = eval_qual(p, p.quals[i], cols, pos);
      if (g == 0) pass = t;
      else {
        or_seen |= 1u << g;
        if (t) or_any |= 1u << g;
      }
    }
    if (!pass || or_seen != or_any) continue;
    Matches jm;
    jm.count = 1;  // no join: one pass without an inner row
    if (p.join_outer_col >= 0) {
      const int nk = p.n_join_cols > 1 ? p.n_join_cols : 1;
      int64_t jk[MI355Q_MAX_GROUP_COLS];
      bool null_key = false;
      for (int i = 0; i < nk; ++i) {
        const int jc_idx = (i == 0 && p.n_join_cols <= 1) ? p.join_outer_col : p.join_outer_cols[i];
        const auto& jc = p.cols[jc_idx];
        jk[i] = decode_col(jc, cols[jc_idx], pos);
        // hash_join_idx_nullable / NULL never equals anything
        null_key = null_key || (jc.nullable && jk[i] == int_null_of(logical_type_of(jc)));
      }
      jm = null_key ? Matches{} : join_lookup(*c.join, jk);
      if (jm.count <= 0) {
        if (p.join_kind != MI355Q_JOIN_LEFT) continue;  // INNER join: no match drops the row
        jm = Matches{};   // LEFT join: the row survives once with a NULL inner side
        jm.count = 1;
      }
    }
    for (int k = 0; k < nx; ++k)  // group-by and target expressions: rows that reach the body
      if (!(qual_exprs & (1u << k)))
        if (int32_t e = eval_into_cell(k, pos)) return e;
    if (q.desc_type == MI355Q_PROJECTION) {
     // the body runs once per joined row (the join loops enclose it, IRCodegen.cpp buildJoinLoops): one output entry each;
     // an inner column is read through the matched row id, or is the type's NULL where a LEFT join found no match
     // (codegenOuterJoinNullPlaceholder, ColumnIR.cpp)
     for (int jm_i = 0; jm_i < jm.count; ++jm_i) {
      const int64_t inner_pos = p.join_outer_col < 0 ? -1 : jm.ids ? (int64_t)jm.ids[jm_i] : jm.single;
      const int64_t pos_outer = pos;
      // GroupByAndAggregate::codegen (GroupByAndAggregate.cpp:1080-1101): crt_matched = 1, old_total_matched =
      // total_matched++ (the CPU form of the atomic add); codegenOutputSlot (:1255-1275): the entry `old_total_matched`
      // of a buffer of max_matched entries, its key = the row's offset in the fragment
      const int32_t old_total_matched = (*c.total_matched)++;
      const uint32_t max_matched = (uint32_t)q.entry_count;  // Execute.cpp:4278-4280: the scan limit, else the entry count
      int64_t* out_slots = nullptr;
      int32_t out_off = -1;
      if (q.output_columnar) {
        out_off = get_columnar_scan_output_offset(buf, max_matched, (uint32_t)old_total_matched, pos);
      } else {
        out_slots = get_scan_output_slot(buf, max_matched, (uint32_t)old_total_matched, pos, (uint32_t)rq);
      }
      if (q.output_columnar ? out_off < 0 : !out_slots) {
        // "return -pos" (GroupByAndAggregate.cpp:1151-1156).  With a scan limit the template's loop has stopped before
        // (QueryTemplateGenerator.cpp:751-780: the loop goes on while old_total_matched + crt_matched < max_matched).
        if (p.scan_limit) return 0;
        if (pos == 0) continue;  // (-0: the row function's answer for row 0 is "no error"; the row is lost)
        return -(int32_t)pos;
      }
      for (int ti = 0; ti < p.n_targets; ++ti) {
        const TargetDesc& t = c.ts[ti];
        const auto& cd = *t.cd;
        if (t.table && inner_pos < 0) {  // LEFT join, no match: the NULL of the inner column's type
          const int64_t nul = int_null_of(t.arg_type);
          if (!q.output_columnar) {
            out_slots[t.slot] = t.arg_f32 ? dbl_bits((double)kNullFloat) : t.arg_fp ? dbl_bits(kNullDouble) : nul;
          } else {
            int8_t* base = reinterpret_cast<int8_t*>(buf) + col_slot_off(q, t.slot);
            switch (q.slot_bytes[t.slot]) {
              case 1: reinterpret_cast<int8_t*>(base)[out_off] = (int8_t)nul; break;
              case 2: reinterpret_cast<int16_t*>(base)[out_off] = (int16_t)nul; break;
              case 4:
                if (t.arg_f32) reinterpret_cast<float*>(base)[out_off] = kNullFloat;
                else reinterpret_cast<int32_t*>(base)[out_off] = (int32_t)nul;
                break;
              default: reinterpret_cast<int64_t*>(base)[out_off] = t.arg_fp ? dbl_bits(kNullDouble) : nul;
            }
          }
          continue;
        }
        const int8_t* col = t.table ? c.inner_cols[t.col] : cols[t.col];
        const int64_t pos = t.table ? inner_pos : pos_outer;
        if (!q.output_columnar) {
          // agg_id / agg_id_double on the 8-byte slot (RuntimeFunctions.cpp:1171-1173,1466-1470) of the value cast to the
          // slot's width: integers sign-extended, a float widened to double
          int64_t v;
          if (t.arg_f32) v = dbl_bits((double)decode_flt(col, pos));
          else if (t.arg_fp) v = dbl_bits(decode_dbl(col, pos));
          else v = decode_col(cd, col, pos);
          out_slots[t.slot] = v;
        } else {
          // columnar projection: the slot's column at its logical width (agg_id_int8 / 16 / 32, agg_id, agg_id_float,
          // agg_id_double; RuntimeFunctions.cpp:1213-1220,1171,1462-1470)
          int8_t* base = reinterpret_cast<int8_t*>(buf) + col_slot_off(q, t.slot);
          switch (q.slot_bytes[t.slot]) {
            case 1: reinterpret_cast<int8_t*>(base)[out_off] = (int8_t)decode_col(cd, col, pos); break;
            case 2: reinterpret_cast<int16_t*>(base)[out_off] = (int16_t)decode_col(cd, col, pos); break;
            case 4:
              if (t.arg_f32) reinterpret_cast<float*>(base)[out_off] = decode_flt(col, pos);
              else reinterpret_cast<int32_t*>(base)[out_off] = (int32_t)decode_col(cd, col, pos);
              break;
            default:
              reinterpret_cast<int64_t*>(base)[out_off] = t.arg_fp ? dbl_bits(decode_dbl(col, pos)) : decode_col(cd, col, pos);
          }
        }
      }
      if (p.scan_limit && (uint32_t)(old_total_matched + 1) >= max_matched) return 0;  // limit reached: the loop ends
     }
      continue;
    }
    int64_t* slots;
    int64_t keys[MI355Q_MAX_GROUP_COLS] = {0, 0, 0, 0};  // group_by_expr_cache_: values as decoded
    // columnar output: the entry ("bin") comes from the *_columnar runtime functions and the
    // aggregate calls address column + bin * width; here the entry's slots are gathered into a
    // row image, updated by the same agg_* calls, and written back
    int64_t col_bin = -1;
    int64_t col_tmp[MI355Q_MAX_SLOTS + 1];
    const bool columnar = q.output_columnar && grouped;
    if (!grouped) {
      slots = buf;  // one entry: every 8-byte slot column holds one value, same bytes as a row
    } else {
      for (int g = 0; g < ng; ++g) {
        const auto& gcd0 = p.cols[p.group_cols[g]];
        // groupByColumnCodegen: a floating-point key is cast to double and bit-cast to i64
        keys[g] = type_is_f32(gcd0.type) ? dbl_bits((double)decode_flt(cols[p.group_cols[g]], pos))
                  : type_is_fp(gcd0.type) ? dbl_bits(decode_dbl(cols[p.group_cols[g]], pos))
                                          : decode_col(gcd0, cols[p.group_cols[g]], pos);
      }
      if (q.desc_type == MI355Q_GROUP_BY_PERFECT_HASH) {
        // groupByColumnCodegen (IRCodegen.cpp:1413-1512): where the column's range has nulls,
        // a NULL key is translated to max + 1 before hashing / storing
        int64_t tk[MI355Q_MAX_GROUP_COLS];
        for (int g = 0; g < ng; ++g) {
          const auto& gcd = p.cols[p.group_cols[g]];
          const bool translate = gcd.nullable && (ng == 1 || q.group_has_nulls[g]);
          tk[g] = (translate && keys[g] == int_null_of(logical_type_of(gcd))) ? q.group_null_key[g]
                                                                              : keys[g];
          // out-of-range key: undefined behaviour in the reference; flagged here
          const int64_t b = q.group_bucket[g] ? q.group_bucket[g] : 1;
          if (tk[g] < q.group_min[g] || (tk[g] - q.group_min[g]) / b >= q.group_card[g]) {
            return MI355Q_ERR_OUT_OF_SLOTS;
          }
        }
        if (columnar && ng == 1) {
          // codegenSingleColumnPerfectHash: get_columnar_group_bin_offset on groups_buffer (the key
          // column — or, keyless, the first slot's column) with the TRANSLATED key
          if (q.keyless && q.slot_width == 4) {
            // NOT restated: here the reference's call reads (and could overwrite) 8 bytes at
            // key_base_ptr[bin] of a column of 4-byte slots, i.e. other entries' slots or memory past
            // the buffer (AddressSanitizer flags the literal restatement); only the bin is taken
            int64_t off = tk[0] - q.min_val;
            if (q.bucket) off /= q.bucket;
            col_bin = off;
          } else {
            col_bin = get_columnar_group_bin_offset(buf, tk[0], q.min_val, q.bucket);
          }
          slots = nullptr;
        } else if (columnar) {
          int64_t hash = 0;
          for (int g = 0; g < ng; ++g) {
            int64_t term = tk[g] - q.group_min[g];
            if (q.group_bucket[g]) term /= q.group_bucket[g];
            for (int prev = 0; prev < g; ++prev) term *= q.group_card[prev];
            hash += term;
          }
          const uint32_t h32 = (uint32_t)hash;
          // codegenMultiColumnPerfectHash, columnar: keys are set unless keyless
          if (!q.keyless) set_matching_group_value_perfect_hash_columnar(buf, h32, tk, ng, (uint32_t)q.entry_count);
          col_bin = h32;
          slots = nullptr;
        } else if (ng == 1) {
          slots = q.keyless ? get_group_value_fast_keyless(buf, tk[0], q.min_val, rq)
                            : get_group_value_fast(buf, tk[0], q.min_val, q.bucket, rq);
        } else {
          // perfect_key_hash (codegenPerfectHashFunction, GroupByAndAggregate.cpp:1546-1598)
          int64_t hash = 0;
          for (int g = 0; g < ng; ++g) {
            int64_t term = tk[g] - q.group_min[g];
            if (q.group_bucket[g]) term /= q.group_bucket[g];  // CreateSDiv
            for (int prev = 0; prev < g; ++prev) term *= q.group_card[prev];
            hash += term;
          }
          const uint32_t h32 = (uint32_t)hash;  // CreateTrunc to i32
          slots = q.keyless ? buf + (size_t)rq * h32  // ..._perfect_hash_keyless :2098-2103
                            : get_matching_group_value_perfect_hash(buf, h32, tk, ng, rq);
        }
      } else if (columnar) {
        // codegenMultiColumnBaselineHash: get_group_value_columnar_slot, 8-byte components
        const int32_t b = get_group_value_columnar_slot(buf, (uint32_t)q.entry_count, keys, ng);
        col_bin = b;
        slots = b < 0 ? nullptr : col_tmp;
      } else if (ng == 1) {
        slots = get_group_value(buf, (uint32_t)q.entry_count, keys[0], q.key_width, rq);
      } else {
        // the sub-keys are stored into an i32 or i64 array of key_count elements
        // (codegenGroupBy, GroupByAndAggregate.cpp:1316-1372)
        int32_t k32[MI355Q_MAX_GROUP_COLS];
        for (int g = 0; g < ng; ++g) k32[g] = (int32_t)keys[g];
        slots = get_group_value_n(buf, (uint32_t)q.entry_count,
                                  q.key_width == 4 ? (const void*)k32 : (const void*)keys, ng,
                                  q.key_width, rq);
      }
      if (columnar && col_bin >= 0) {
        for (int sl = 0; sl < q.slot_count; ++sl) {
          const int8_t* c = reinterpret_cast<const int8_t*>(buf) + col_slot_off(q, sl);
          if (q.slot_width == 8) col_tmp[sl] = reinterpret_cast<const int64_t*>(c)[col_bin];
          else reinterpret_cast<int32_t*>(col_tmp)[sl] = reinterpret_cast<const int32_t*>(c)[col_bin];
        }
        slots = col_tmp;
      }
      if (!slots) {
        // row_func returns -pos -> "ran out of slots" (GroupByAndAggregate.cpp:1151-1156)
        int64_t code = -(pos + 1);
        return (int32_t)std::max<int64_t>(code, INT32_MIN);
      }
      (void)kq;
    }
    if (q.slot_width == 4) {
      // 4-byte slots: agg_count_int32 (RuntimeFunctions.cpp:1222-1225) / agg_id_int32 (:1259-1261)
      int32_t* s32 = reinterpret_cast<int32_t*>(slots);
      for (int m = 0; m < jm.count; ++m) {
        for (const auto& t : c.ts) {
          if (t.agg == MI355Q_PROJECT_KEY) {
            if (t.slot >= 0) s32[t.slot] = (int32_t)keys[t.key_idx];
          } else {
            ++*reinterpret_cast<uint32_t*>(s32 + t.slot);
          }
        }
      }
      if (col_bin >= 0) col_scatter_slots(q, col_tmp, buf, col_bin);
      continue;
    }
    // one joined row per matching inner row (JoinLoop, Set / Singleton kinds)
    for (int m = 0; m < jm.count; ++m) {
      const int64_t inner_pos = jm.ids ? (int64_t)jm.ids[m] : jm.single;
      for (const auto& t : c.ts) {
        apply_target(p, t, slots, cols, pos, c.inner_cols, inner_pos, keys);
      }
    }
    if (col_bin >= 0) col_scatter_slots(q, col_tmp, buf, col_bin);
  }
  return 0;
}

// ---------------------------------------------------------------- reduce
// ResultSetStorage::reduceOneSlot (ResultSetReduction.cpp:1496-1640): the same aggregate
// applied to the other buffer's slot, init value as skip value.
inline void reduce_one_target(const mi355q_qmd& q, int ti, int64_t* this_slots,
                              const int64_t* that_slots) {
  const int s = q.target_slot[ti];
  if (s < 0) return;
  int64_t* a = this_slots + s;
  const int64_t* b = that_slots + s;
  const int64_t init = q.init_vals[s];
  const bool fp = q.target_arg_is_fp[ti];
  const bool skip = q.target_skip_null[ti];
  if (q.target_arg_is_f32[ti]) {  // chosen_bytes = sizeof(float) (ResultSetReduction.cpp:1514-1520)
    int32_t* a32 = reinterpret_cast<int32_t*>(a);
    const float bf = bits_flt((int32_t)*b);
    const float initf = bits_flt((int32_t)init);
    switch (q.target_agg[ti]) {
      case MI355Q_AVG:
        agg_sum(a + 1, b[1]);
        [[fallthrough]];
      case MI355Q_SUM:
      case MI355Q_SUM_IF:
        if (skip) agg_sum_float_skip_val(a32, bf, initf);
        else agg_sum_float(a32, bf);
        break;
      case MI355Q_MIN:
        if (skip) agg_min_float_skip_val(a32, bf, initf);
        else agg_min_float(a32, bf);
        break;
      case MI355Q_MAX:
        if (skip) agg_max_float_skip_val(a32, bf, initf);
        else agg_max_float(a32, bf);
        break;
    }
    return;
  }
  switch (q.target_agg[ti]) {
    case MI355Q_COUNT:
    case MI355Q_COUNT_IF:  // ResultSetReduction.cpp:1524-1535
      agg_sum(a, *b);  // AGGREGATE_ONE_COUNT
      break;
    case MI355Q_AVG:
      agg_sum(a + 1, b[1]);
      [[fallthrough]];
    case MI355Q_SUM:
    case MI355Q_SUM_IF:  // :1543-1548
      if (skip) {
        if (fp) agg_sum_double_skip_val(a, bits_dbl(*b), bits_dbl(init));
        else agg_sum_skip_val(a, *b, init);
      } else {
        if (fp) agg_sum_double(a, bits_dbl(*b));
        else agg_sum(a, *b);
      }
      break;
    case MI355Q_MIN:
      if (skip) {
        if (fp) agg_min_double_skip_val(a, bits_dbl(*b), bits_dbl(init));
        else agg_min_skip_val(a, *b, init);
      } else {
        if (fp) agg_min_double(a, bits_dbl(*b));
        else agg_min(a, *b);
      }
      break;
    case MI355Q_MAX:
      if (skip) {
        if (fp) agg_max_double_skip_val(a, bits_dbl(*b), bits_dbl(init));
        else agg_max_skip_val(a, *b, init);
      } else {
        if (fp) agg_max_double(a, bits_dbl(*b));
        else agg_max(a, *b);
      }
      break;
    default:  // non-agg projection (:1584-1632, 8-byte case)
      if (*b != init) *a = *b;
  }
}

// ResultSetStorage::isEmptyEntry (ResultSetIteration.cpp:2457-2492)
inline bool is_empty_entry(const mi355q_qmd& q, const int64_t* buf, int64_t e) {
  if (q.desc_type == MI355Q_NON_GROUPED_AGGREGATE) return false;
  if (q.output_columnar) {  // isEmptyEntryColumnar (ResultSet.cpp): first key column / the keyless key slot's column
    const int8_t* b = reinterpret_cast<const int8_t*>(buf);
    if (q.keyless) {
      const int8_t* c = b + col_slot_off(q, q.idx_target_as_key);
      return q.slot_width == 4 ? reinterpret_cast<const int32_t*>(c)[e] == (int32_t)q.init_vals[q.idx_target_as_key]
                               : reinterpret_cast<const int64_t*>(c)[e] == q.init_vals[q.idx_target_as_key];
    }
    return reinterpret_cast<const int64_t*>(b)[e] == kEmptyKey64;
  }
  const int64_t* row = buf + e * (q.row_size / 8);
  if (q.keyless && q.slot_width == 4) {
    return reinterpret_cast<const int32_t*>(row)[q.idx_target_as_key] == (int32_t)q.init_vals[q.idx_target_as_key];
  }
  if (q.keyless) {
    return row[q.idx_target_as_key] == q.init_vals[q.idx_target_as_key];
  }
  if (q.key_width == 4) return *reinterpret_cast<const int32_t*>(row) == kEmptyKey32;
  return row[0] == kEmptyKey64;
}

// reduceOneSlot with 4-byte slots: AGGREGATE_ONE_COUNT on 32 bits, projections copied when set
inline void reduce_one_target_compact(const mi355q_qmd& q, int ti, int64_t* this_slots,
                                      const int64_t* that_slots) {
  const int s = q.target_slot[ti];
  if (s < 0) return;
  int32_t* a = reinterpret_cast<int32_t*>(this_slots) + s;
  const int32_t b = reinterpret_cast<const int32_t*>(that_slots)[s];
  if (q.target_agg[ti] == MI355Q_PROJECT_KEY) {
    if (b != (int32_t)q.init_vals[s]) *a = b;
  } else {
    *a = (int32_t)((uint32_t)*a + (uint32_t)b);
  }
}
inline void reduce_targets(const mi355q_qmd& q, int64_t* this_slots, const int64_t* that_slots) {
  for (int t = 0; t < q.n_targets; ++t) {
    if (q.slot_width == 4) reduce_one_target_compact(q, t, this_slots, that_slots);
    else reduce_one_target(q, t, this_slots, that_slots);
  }
}

// ResultSetStorage::reduce (ResultSetReduction.cpp:203-383); baseline entries re-hash
// into `this` (get_group_value_reduction, :783-826).
int32_t reduce_buffers(const mi355q_qmd& q, int64_t* this_buf, const int64_t* that_buf) {
  const int rq = q.row_size / 8;
  const int kq = q.key_bytes / 8;
  if (q.output_columnar) {
    // ResultSetStorage::reduce on columnar buffers: the same per-entry rules through column offsets
    // (reduceOneEntryNoCollisions; baseline: reduceOneEntryBaseline with
    // get_group_value_columnar_reduction, ResultSetReduction.cpp:621-676).  The slots of the two
    // entries are gathered into row images for reduceOneSlot and written back.
    const mi355q_qmd qr = rowwise_of(q);
    std::vector<int64_t> a(rq), b(rq);
    for (int64_t e = 0; e < q.entry_count; ++e) {
      if (is_empty_entry(q, that_buf, e)) continue;
      col_gather(q, that_buf, e, b.data());
      int64_t bin = e;
      if (q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
        // get_group_value_columnar_reduction: claim an empty bin with the key or find the key
        const uint32_t ec = (uint32_t)q.entry_count;
        const uint32_t h = murmur3(b.data(), (int)(kq * sizeof(int64_t)), 0) % ec;
        bin = -1;
        for (uint32_t i = 0, hp = h; i < ec; ++i, hp = (hp + 1) % ec) {
          if (get_matching_group_value_columnar_slot(this_buf, ec, hp, b.data(), (uint32_t)kq) != -1) {
            bin = hp;
            break;
          }
        }
        if (bin < 0) return MI355Q_ERR_OUT_OF_SLOTS;
      }
      col_gather(q, this_buf, bin, a.data());
      for (int k = 0; k < kq; ++k) a[k] = b[k];  // perfect hash: the key copy from the right-hand side
      reduce_targets(qr, a.data() + kq, b.data() + kq);
      col_scatter(q, a.data(), this_buf, bin);
    }
    return 0;
  }
  if (q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
    for (int64_t e = 0; e < q.entry_count; ++e) {
      if (is_empty_entry(q, that_buf, e)) continue;
      const int64_t* that_row = that_buf + e * rq;
      int64_t* slots;
      if (q.group_col_count <= 1) {
        int64_t key = q.key_width == 4 ? (int64_t) * reinterpret_cast<const int32_t*>(that_row)
                                       : that_row[0];
        slots = get_group_value(this_buf, (uint32_t)q.entry_count, key, q.key_width, rq);
      } else {  // the row's key bytes are the key (get_group_value_reduction)
        slots = get_group_value_n(this_buf, (uint32_t)q.entry_count, that_row, q.group_col_count,
                                  q.key_width, rq);
      }
      if (!slots) return MI355Q_ERR_OUT_OF_SLOTS;
      reduce_targets(q, slots, that_row + kq);
    }
    return 0;
  }
  for (int64_t e = 0; e < q.entry_count; ++e) {
    if (is_empty_entry(q, that_buf, e)) continue;
    int64_t* this_row = this_buf + e * rq;
    const int64_t* that_row = that_buf + e * rq;
    for (int k = 0; k < kq; ++k) this_row[k] = that_row[k];  // key memcpy from rhs (ResultSetReductionJIT.cpp:705-711)
    reduce_targets(q, this_row + kq, that_row + kq);
  }
  return 0;
}

}  // namespace

// ================================================================ exported C API
ORC_EXPORT uint32_t orc_murmur3(const void* key, int len, uint32_t seed) {
  return murmur3(key, len, seed);
}
ORC_EXPORT uint32_t orc_murmur1(const void* key, int len, uint32_t seed) {
  return murmur1(key, len, seed);
}

ORC_EXPORT int32_t orc_qmd_init(const mi355q_plan* plan, mi355q_qmd* out) {
  mi355q_plan lp;
  if (int e = lower_plan(*plan, &lp)) return e;
  return qmd_init(lp, *out);
}

// one projected expression on one row (golden vectors of the cast / arithmetic semantics): the value as a
// 64-bit pattern (integers sign-extended, DOUBLE bits, FLOAT bits in the low word), its type and NULL flag;
// returns 0 or ErrorCode 7
ORC_EXPORT int32_t orc_eval_expr(const mi355q_plan* plan, int32_t k, const void* const* cols, int64_t pos,
                                 int64_t* out_bits, int32_t* out_type, int32_t* out_is_null) {
  if (k < 0 || k >= plan->n_exprs) return MI355Q_ERR_INVALID_PLAN;
  mi355q_plan lp;
  if (int e = lower_plan(*plan, &lp)) return e;
  OrcVal v;
  if (int32_t e = eval_expression(*plan, plan->exprs[k], reinterpret_cast<const int8_t* const*>(cols), pos, &v)) return e;
  *out_bits = v.type == MI355Q_DOUBLE ? dbl_bits(v.d) : v.type == MI355Q_FLOAT ? (int64_t)(uint32_t)flt_bits(v.f) : v.i;
  *out_type = v.type;
  *out_is_null = val_is_null(v) ? 1 : 0;
  return 0;
}

ORC_EXPORT void orc_init_buffer(const mi355q_qmd* q, int64_t* buf) { init_buffer(*q, buf); }

// raw slot-lookup entry points for the golden traces
ORC_EXPORT int64_t orc_get_group_value_slot(int64_t* buf, uint32_t entry_count, int64_t key,
                                            uint32_t key_width, uint32_t row_size_quad) {
  int64_t* p = get_group_value(buf, entry_count, key, key_width, row_size_quad);
  return p ? (p - buf) : -1;
}
ORC_EXPORT int64_t orc_get_group_value_fast_slot(int64_t* buf, int64_t key, int64_t min_key,
                                                 uint32_t row_size_quad) {
  return get_group_value_fast(buf, key, min_key, 0, row_size_quad) - buf;
}
ORC_EXPORT int64_t orc_get_group_value_fast_bucket_slot(int64_t* buf, int64_t key, int64_t min_key,
                                                        int64_t bucket, uint32_t row_size_quad) {
  return get_group_value_fast(buf, key, min_key, bucket, row_size_quad) - buf;
}
// multi-component keys: `key` = key_count packed int32 / int64 components
ORC_EXPORT int64_t orc_get_group_value_n_slot(int64_t* buf, uint32_t entry_count, const void* key,
                                              uint32_t key_count, uint32_t key_width,
                                              uint32_t row_size_quad) {
  int64_t* p = get_group_value_n(buf, entry_count, key, key_count, key_width, row_size_quad);
  return p ? (p - buf) : -1;
}
ORC_EXPORT int64_t orc_perfect_hash_slot(int64_t* buf, uint32_t hashed_index, const int64_t* key,
                                         uint32_t key_count, uint32_t row_size_quad) {
  return get_matching_group_value_perfect_hash(buf, hashed_index, key, key_count, row_size_quad) - buf;
}
// the column fetch of the row function for an encoded column
ORC_EXPORT int64_t orc_decode_col(const mi355q_col_desc* cd, const void* col, int64_t pos) {
  return decode_col(*cd, static_cast<const int8_t*>(col), pos);
}

// ---- join build (restating HashJoinRuntime.cpp:71-86,203-216 one-to-one perfect;
// :346-373,505-538,575-640 keyed; :652-700,895-945,1503-1560 one-to-many perfect;
// :1975-2100 one-to-many keyed).  Serial, so payload runs are in row order.
namespace {

struct KeyCols {
  const int8_t* col[MI355Q_MAX_GROUP_COLS];
  int type[MI355Q_MAX_GROUP_COLS];
  bool nullable[MI355Q_MAX_GROUP_COLS];
  int n;
};
// false: a NULL component -> the row is not inserted (GenericKeyHandler / fill_hash_join_buff_impl)
bool load_key(const KeyCols& kc, int64_t i, int64_t* keys) {
  for (int k = 0; k < kc.n; ++k) {
    keys[k] = decode_int(kc.col[k], kc.type[k], i);
    if (kc.nullable[k] && keys[k] == int_null_of(kc.type[k])) return false;
  }
  return true;
}

// write_baseline_hash_slot / get_matching_baseline_hash_slot_at (HashJoinRuntime.cpp:465-538):
// claim or find the key's slot; stride in components
template <typename T>
int64_t keyed_slot_insert(T* tab, int64_t entry_count, int n_keys, int stride, const int64_t* keys,
                          T empty) {
  T k[MI355Q_MAX_GROUP_COLS];
  for (int i = 0; i < n_keys; ++i) k[i] = (T)keys[i];
  uint32_t h = murmur1(k, n_keys * sizeof(T), 0) % entry_count;
  const uint32_t start = h;
  do {
    T* e = tab + (size_t)h * stride;
    if (e[0] == empty) {
      memcpy(e, k, n_keys * sizeof(T));
      return h;
    }
    if (memcmp(e, k, n_keys * sizeof(T)) == 0) return h;
    h = (h + 1) % entry_count;
  } while (h != start);
  return -1;
}

int32_t build_join(OrcJoin& j, const KeyCols& kc, int64_t num_rows, bool perfect, bool one_to_many,
                   int64_t min_key, int64_t max_key, int64_t keyed_entries) {
  int width = 4;  // BaselineJoinHashTable::getKeyComponentWidth
  for (int i = 0; i < kc.n; ++i)
    if (type_width(kc.type[i]) > 4) width = 8;
  j.n_keys = kc.n;
  j.width = perfect ? 8 : width;
  j.min_key = perfect ? min_key : 0;
  j.max_key = perfect ? max_key : 0;
  // keyed: entry_count = 2 x max(tuples, 1) (BaselineJoinHashTable.cpp:484-486)
  j.entry_count = perfect ? max_key - min_key + 1
                          : (keyed_entries > 0 ? keyed_entries : 2 * std::max<int64_t>(num_rows, 1));
  const int64_t n = j.entry_count;
  int64_t keys[MI355Q_MAX_GROUP_COLS];
  auto keyed_init = [&](int stride) {
    for (int64_t e = 0; e < n; ++e) {
      for (int c = 0; c < stride; ++c) {
        if (width == 4) reinterpret_cast<int32_t*>(j.buf.data())[e * stride + c] = c < kc.n ? kEmptyKey32 : -1;
        else reinterpret_cast<int64_t*>(j.buf.data())[e * stride + c] = c < kc.n ? kEmptyKey64 : -1;
      }
    }
  };
  auto keyed_insert = [&](int stride) -> int64_t {
    return width == 4 ? keyed_slot_insert<int32_t>(reinterpret_cast<int32_t*>(j.buf.data()), n, kc.n, stride,
                                                   keys, kEmptyKey32)
                      : keyed_slot_insert<int64_t>(reinterpret_cast<int64_t*>(j.buf.data()), n, kc.n, stride,
                                                   keys, kEmptyKey64);
  };
  if (!one_to_many) {
    if (perfect) {
      j.hash_type = 0;
      j.buf.assign((size_t)n * 4, (int8_t)0xFF);  // init_hash_join_buff: -1
      int32_t* slots = reinterpret_cast<int32_t*>(j.buf.data());
      for (int64_t i = 0; i < num_rows; ++i) {
        if (!load_key(kc, i, keys)) continue;
        if (keys[0] < min_key || keys[0] > max_key) return MI355Q_ERR_INVALID_PLAN;
        int32_t& slot = slots[keys[0] - min_key];
        if (slot != -1) return MI355Q_ERR_JOIN_NOT_ONE_TO_ONE;  // fill_one_to_one_hashtable CAS failure
        slot = (int32_t)i;
      }
      return 0;
    }
    j.hash_type = 1;
    const int stride = kc.n + 1;
    j.buf.assign((size_t)n * stride * width, 0);
    keyed_init(stride);
    for (int64_t i = 0; i < num_rows; ++i) {
      if (!load_key(kc, i, keys)) continue;
      const int64_t slot = keyed_insert(stride);
      if (slot < 0) return MI355Q_ERR_JOIN_TABLE_FULL;
      if (width == 4) {
        int32_t& pay = reinterpret_cast<int32_t*>(j.buf.data())[slot * stride + kc.n];
        if (pay != -1) return MI355Q_ERR_JOIN_NOT_ONE_TO_ONE;
        pay = (int32_t)i;
      } else {
        int64_t& pay = reinterpret_cast<int64_t*>(j.buf.data())[slot * stride + kc.n];
        if (pay != -1) return MI355Q_ERR_JOIN_NOT_ONE_TO_ONE;
        pay = i;
      }
    }
    return 0;
  }
  // one-to-many
  j.hash_type = perfect ? 2 : 3;
  const size_t key_bytes = perfect ? 0 : (size_t)n * kc.n * width;
  j.buf.assign(key_bytes + (size_t)(2 * n + std::max<int64_t>(num_rows, 1)) * 4, 0);
  if (!perfect) keyed_init(kc.n);
  int32_t* offsets = reinterpret_cast<int32_t*>(j.buf.data() + key_bytes);
  int32_t* counts = offsets + n;
  int32_t* payloads = counts + n;
  std::vector<int64_t> slot_of(num_rows, -1);
  for (int64_t i = 0; i < num_rows; ++i) {  // keys, then count_matches
    if (!load_key(kc, i, keys)) continue;
    int64_t slot;
    if (perfect) {
      if (keys[0] < min_key || keys[0] > max_key) return MI355Q_ERR_INVALID_PLAN;
      slot = keys[0] - min_key;
    } else {
      slot = keyed_insert(kc.n);
      if (slot < 0) return MI355Q_ERR_JOIN_TABLE_FULL;
    }
    slot_of[i] = slot;
    ++counts[slot];
  }
  // inclusive_scan of the shifted counts; pos only where count != 0, -1 elsewhere (:1525-1548)
  int32_t acc = 0;
  for (int64_t e = 0; e < n; ++e) {
    offsets[e] = counts[e] ? acc : -1;
    acc += counts[e];
  }
  std::fill(counts, counts + n, 0);
  for (int64_t i = 0; i < num_rows; ++i) {  // fill_row_ids
    if (slot_of[i] < 0) continue;
    payloads[offsets[slot_of[i]] + counts[slot_of[i]]++] = (int32_t)i;
  }
  return 0;
}

}  // namespace

// key_cols / key_types / key_nullables: n_keys inner key columns.  one_to_many: 0 = OneToOne
// only, 1 = rebuild as OneToMany on a duplicate (HashJoin::getInstance retry), 2 = OneToMany.
ORC_EXPORT void* orc_join_build_n(const void* const* key_cols, const int32_t* key_types,
                                  const int32_t* key_nullables, int32_t n_keys, int64_t num_rows,
                                  int64_t min_key, int64_t max_key, int prefer_baseline,
                                  int64_t max_perfect_entries, int32_t one_to_many,
                                  int64_t keyed_entries, int32_t* err) {
  KeyCols kc{};
  kc.n = n_keys;
  for (int i = 0; i < n_keys; ++i) {
    kc.col[i] = static_cast<const int8_t*>(key_cols[i]);
    kc.type[i] = key_types[i];
    kc.nullable[i] = key_nullables[i] != 0;
  }
  *err = 0;
  if (max_perfect_entries <= 0) max_perfect_entries = INT32_MAX;  // PerfectJoinHashTable.cpp:219
  __int128 span = (__int128)max_key - (__int128)min_key;
  const bool perfect = n_keys == 1 && !prefer_baseline && max_key >= min_key && span < max_perfect_entries;
  auto* j = new OrcJoin();
  for (int attempt = one_to_many == 2 ? 1 : 0; attempt < 2; ++attempt) {
    *err = build_join(*j, kc, num_rows, perfect, attempt == 1, min_key, max_key, keyed_entries);
    if (*err != MI355Q_ERR_JOIN_NOT_ONE_TO_ONE || one_to_many == 0) break;
  }
  if (*err) {
    delete j;
    return nullptr;
  }
  return j;
}
ORC_EXPORT void* orc_join_build(const void* key_col, int key_type, int key_nullable,
                                int64_t num_rows, int64_t min_key, int64_t max_key,
                                int prefer_baseline, int64_t max_perfect_entries,
                                int32_t* err) {
  const void* cols[1] = {key_col};
  const int32_t types[1] = {key_type}, nulls[1] = {key_nullable};
  return orc_join_build_n(cols, types, nulls, 1, num_rows, min_key, max_key, prefer_baseline,
                          max_perfect_entries, 0, 0, err);
}
ORC_EXPORT void orc_join_free(void* j) { delete static_cast<OrcJoin*>(j); }
ORC_EXPORT int64_t orc_join_probe(const void* jp, int64_t key) {
  const auto* j = static_cast<const OrcJoin*>(jp);
  const Matches m = join_lookup(*j, &key);
  if (j->hash_type == 0) return hash_join_idx(j->perfect(), key, j->min_key, j->max_key);
  if (j->hash_type == 1 && j->n_keys == 1 && j->width == 8)
    return baseline_hash_join_idx_64(reinterpret_cast<const int64_t*>(j->buf.data()), key, j->entry_count);
  return m.count ? (m.ids ? m.ids[0] : m.single) : -1;
}
// the matching set of a (composite) key: writes up to max_ids row ids, returns the count
ORC_EXPORT int32_t orc_join_matches(const void* jp, const int64_t* keys, int32_t* ids, int32_t max_ids) {
  const auto* j = static_cast<const OrcJoin*>(jp);
  const Matches m = join_lookup(*j, keys);
  for (int i = 0; i < m.count && i < max_ids; ++i) ids[i] = m.ids ? m.ids[i] : (int32_t)m.single;
  return m.count;
}
ORC_EXPORT int32_t orc_join_info(const void* jp, int32_t* hash_type, int64_t* entry_count) {
  const auto* j = static_cast<const OrcJoin*>(jp);
  *hash_type = j->hash_type;
  *entry_count = j->entry_count;
  return 0;
}
ORC_EXPORT int32_t orc_join_shape(const void* jp, int32_t* n_keys, int32_t* width, int64_t* bytes) {
  const auto* j = static_cast<const OrcJoin*>(jp);
  *n_keys = j->n_keys;
  *width = j->width;
  *bytes = (int64_t)j->buf.size();
  return 0;
}
ORC_EXPORT const void* orc_join_buffer(const void* jp) {
  return static_cast<const OrcJoin*>(jp)->buf.data();
}

// ---- execute: kernel per fragment on `n_threads` host threads, each with a private
// output buffer (Execute.cpp:3121-3153, one ExecutionKernel per fragment on CPU), then the
// buffers are reduced into the first one in order (Execute.cpp:1772-1792).  Input
// pointers are HOST pointers.  out_buf must hold qmd.entry_count * qmd.row_size bytes.
static thread_local int32_t t_last_total_matched = 0;
// total_matched of the calling thread's last Projection run
ORC_EXPORT int64_t orc_last_total_matched() { return t_last_total_matched; }

ORC_EXPORT int32_t orc_execute(const mi355q_plan* plan, const mi355q_inputs* in,
                               const void* join, int32_t n_threads, int64_t* out_buf,
                               mi355q_qmd* out_qmd) {
  ExecCtx c;
  c.stated = plan;
  if (int e = lower_plan(*plan, &c.lowered)) return e;
  c.plan = &c.lowered;
  if (int e = qmd_init(c.lowered, c.qmd)) return e;
  if (int e = build_targets(c.lowered, plan->n_group_cols > 0, c.ts)) return e;
  for (int i = 0; i < plan->n_targets; ++i) c.ts[i].slot = c.qmd.target_slot[i];
  c.join = static_cast<const OrcJoin*>(join);
  c.inner_cols = reinterpret_cast<const int8_t* const*>(in->inner_col_buffers);
  c.inner_rows = in->inner_num_rows;
  if (plan->join_outer_col >= 0 && !c.join) return MI355Q_ERR_INVALID_PLAN;
  if (out_qmd) *out_qmd = c.qmd;
  if (c.qmd.desc_type == MI355Q_PROJECTION) {
    // multifrag_query (RuntimeFunctions.cpp:2434-2471): the fragments one after the other through the same row function,
    // ONE output buffer and ONE total_matched word — the shape of the reference's multi-fragment (GPU) kernel, run here
    // in row order like its CPU kernel (pos_start 0, pos_step 1)
    t_last_total_matched = 0;
    c.total_matched = &t_last_total_matched;
    init_buffer(c.qmd, out_buf);
    for (int f = 0; f < in->n_frags; ++f) {
      if (plan->scan_limit && t_last_total_matched >= c.qmd.entry_count) break;  // (the loop condition of every later fragment fails at once)
      const int8_t* const* cols = reinterpret_cast<const int8_t* const*>(in->col_buffers + (size_t)f * plan->n_cols);
      if (const int32_t e = run_fragment(c, cols, in->num_rows[f], out_buf)) return e;
    }
    return 0;
  }
  const size_t quads = (size_t)(buffer_bytes(c.qmd) / 8);
  n_threads = std::max(1, std::min(n_threads, std::max(1, in->n_frags)));

  std::vector<std::vector<int64_t>> bufs(n_threads);
  std::vector<int32_t> errs(n_threads, 0);
  auto worker = [&](int tid) {
    int64_t* buf = tid == 0 ? out_buf : (bufs[tid].resize(quads), bufs[tid].data());
    init_buffer(c.qmd, buf);
    // fragments are dealt to the kernels statically (f = tid, tid + n_threads, ...): which partial
    // buffer a row lands in — and with it every order-dependent corner of the reference's reduce,
    // e.g. a keyless all-NULL group that looks empty — is the same on every run
    for (int f = tid; f < in->n_frags; f += n_threads) {
      const int8_t* const* cols =
          reinterpret_cast<const int8_t* const*>(in->col_buffers + (size_t)f * plan->n_cols);
      const int32_t e = run_fragment(c, cols, in->num_rows[f], buf);
      if (e) {
        errs[tid] = e;
        break;
      }
    }
  };
  if (n_threads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
  }
  for (int t = 0; t < n_threads; ++t)
    if (errs[t]) return errs[t];
  for (int t = 1; t < n_threads; ++t) {
    if (int e = reduce_buffers(c.qmd, out_buf, bufs[t].data())) return e;
    std::vector<int64_t>().swap(bufs[t]);
  }
  return 0;
}

// the projection runtime on a caller's buffer (golden vectors): the quad index of the slots / the offset, -1 when full
ORC_EXPORT int64_t orc_get_scan_output_slot(int64_t* buf, uint32_t entry_count, uint32_t pos, int64_t offset_in_fragment,
                                            uint32_t row_size_quad) {
  int64_t* p = get_scan_output_slot(buf, entry_count, pos, offset_in_fragment, row_size_quad);
  return p ? p - buf : -1;
}
ORC_EXPORT int32_t orc_get_columnar_scan_output_offset(int64_t* buf, uint32_t entry_count, uint32_t pos, int64_t offset_in_fragment) {
  return get_columnar_scan_output_offset(buf, entry_count, pos, offset_in_fragment);
}
// get_group_value_columnar_slot on a columnar buffer's key columns: the bin, or -1 when full
ORC_EXPORT int32_t orc_get_group_value_columnar_slot(int64_t* buf, uint32_t entry_count, const int64_t* key,
                                                     uint32_t key_count) {
  return get_group_value_columnar_slot(buf, entry_count, key, key_count);
}
// get_columnar_group_bin_offset on a key column (golden vectors)
ORC_EXPORT uint32_t orc_get_columnar_group_bin_offset(int64_t* key_col, int64_t key, int64_t min_key, int64_t bucket) {
  return get_columnar_group_bin_offset(key_col, key, min_key, bucket);
}
ORC_EXPORT int64_t orc_buffer_bytes(const mi355q_qmd* q) { return buffer_bytes(*q); }
ORC_EXPORT int64_t orc_col_group_off(const mi355q_qmd* q, int32_t g) { return col_group_off(*q, g); }
ORC_EXPORT int64_t orc_col_slot_off(const mi355q_qmd* q, int32_t s) { return col_slot_off(*q, s); }

ORC_EXPORT int32_t orc_reduce(const mi355q_qmd* q, int64_t* this_buf, const int64_t* that_buf) {
  return reduce_buffers(*q, this_buf, that_buf);
}

ORC_EXPORT int64_t orc_row_count(const mi355q_qmd* q, const int64_t* buf) {
  int64_t n = 0;
  for (int64_t e = 0; e < q->entry_count; ++e) n += !is_empty_entry(*q, buf, e);
  return n;
}

// ResultSet::getNextRow over all entries (ResultSetIteration.cpp:125-230): per non-empty
// entry, one value per target.  AVG via pair_to_double (ResultSetBufferAccessors.h:197-227).
ORC_EXPORT int32_t orc_fetch_rows(const mi355q_qmd* q, const int64_t* buf, int64_t max_rows,
                                  int64_t* ival, double* dval, int8_t* is_null,
                                  int64_t* n_rows) {
  if (q->desc_type == MI355Q_PROJECTION) {
    // getTargetValueFromBufferRowwise / Colwise over Projection storage: the slot at its padded width, integers
    // sign-extended (read_int_from_buff), a 4-byte fp slot a float, an 8-byte fp slot a double — also for a FLOAT
    // target (ResultSetIteration.cpp makeTargetValue: `chosen_type.is_fp()` reads by slot width); isNull by the
    // target type's inline NULL unless the type is NOT NULL
    const int rq = q->row_size / 8;
    int64_t n = 0;
    for (int64_t e = 0; e < q->entry_count && n < max_rows; ++e) {
      if (is_empty_entry(*q, buf, e)) continue;
      for (int t = 0; t < q->n_targets; ++t) {
        const size_t o = (size_t)n * q->n_targets + t;
        const int s = q->target_slot[t];
        int64_t v;
        if (!q->output_columnar) {
          v = buf[e * rq + 1 + s];
        } else {
          const int8_t* base = reinterpret_cast<const int8_t*>(buf) + col_slot_off(*q, s);
          switch (q->slot_bytes[s]) {
            case 1: v = reinterpret_cast<const int8_t*>(base)[e]; break;
            case 2: v = reinterpret_cast<const int16_t*>(base)[e]; break;
            case 4: v = reinterpret_cast<const int32_t*>(base)[e]; break;
            default: v = reinterpret_cast<const int64_t*>(base)[e];
          }
        }
        ival[o] = 0;
        dval[o] = 0;
        const bool nullable = q->target_null[t] != kEmptyKey64;
        if (q->target_arg_is_f32[t]) {
          dval[o] = (double)bits_flt((int32_t)v);
          is_null[o] = nullable && (int32_t)v == (int32_t)q->target_null[t];
        } else if (q->target_is_fp[t]) {
          dval[o] = bits_dbl(v);
          is_null[o] = nullable && v == q->target_null[t];
        } else {
          ival[o] = v;
          is_null[o] = nullable && v == q->target_null[t];
        }
      }
      ++n;
    }
    *n_rows = n;
    return 0;
  }
  if (q->output_columnar) {  // getTargetValueFromBufferColwise reads the same values through column offsets
    const mi355q_qmd qr = rowwise_of(*q);
    const std::vector<int64_t> rows = col_to_rows(*q, buf);
    return orc_fetch_rows(&qr, rows.data(), max_rows, ival, dval, is_null, n_rows);
  }
  const int rq = q->row_size / 8;
  const int kq = q->key_bytes / 8;
  int64_t n = 0;
  for (int64_t e = 0; e < q->entry_count && n < max_rows; ++e) {
    if (is_empty_entry(*q, buf, e)) continue;
    const int64_t* row = buf + e * rq;
    for (int t = 0; t < q->n_targets; ++t) {
      const size_t o = (size_t)n * q->n_targets + t;
      ival[o] = 0;
      dval[o] = 0;
      is_null[o] = 0;
      const int s = q->target_slot[t];
      if (q->target_agg[t] == MI355Q_PROJECT_KEY && s < 0) {
        const int ki = q->target_key_idx[t];
        ival[o] = q->key_width == 4 ? (int64_t) reinterpret_cast<const int32_t*>(row)[ki] : row[ki];
        is_null[o] = ival[o] == q->target_null[t];
        if (q->target_is_fp[t]) {  // getTargetValueFromBufferRowwise on an 8-byte fp key: a double
          dval[o] = bits_dbl(ival[o]);
          ival[o] = 0;
        }
        continue;
      }
      const int64_t v = q->slot_width == 4 ? (int64_t) reinterpret_cast<const int32_t*>(row + kq)[s] : row[kq + s];
      switch (q->target_agg[t]) {
        case MI355Q_AVG: {
          const int64_t cnt = row[kq + s + 1];
          if (cnt == 0) {
            dval[o] = kNullDouble;
            is_null[o] = 1;
          } else {
            // pair_to_double (ResultSetBufferAccessors.h:197-227): float_argument_input reads a float
            const double dividend = q->target_arg_is_f32[t] ? (double)bits_flt((int32_t)v)
                                    : q->target_arg_is_fp[t] ? bits_dbl(v) : (double)v;
            dval[o] = dividend / (double)cnt;
          }
          break;
        }
        case MI355Q_COUNT:
        case MI355Q_COUNT_IF:
          ival[o] = v;
          break;
        default:
          if (q->target_arg_is_f32[t]) {  // float slot: 4 bytes (getTargetValueFromBufferRowwise)
            dval[o] = (double)bits_flt((int32_t)v);
            is_null[o] = q->target_skip_null[t] && (int32_t)v == (int32_t)q->target_null[t];
          } else if (q->target_is_fp[t]) {
            dval[o] = bits_dbl(v);
            is_null[o] = q->target_skip_null[t] && v == q->target_null[t];
          } else {
            ival[o] = v;
            // ResultSet::isNull: nullable type && value == null bit pattern
            const bool nullable =
                q->target_skip_null[t] || q->target_agg[t] == MI355Q_PROJECT_KEY;
            is_null[o] = nullable && v == q->target_null[t];
          }
      }
    }
    ++n;
  }
  *n_rows = n;
  return 0;
}

// ---- synthetic generator: u = splitmix64(seed ^ row * golden)
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
ORC_EXPORT uint64_t orc_splitmix64(uint64_t x) { return splitmix64(x); }

ORC_EXPORT int32_t orc_generate_column(void* dst, int64_t n_rows, int64_t row_offset,
                                       int32_t kind, uint64_t seed, int64_t a, int64_t b,
                                       int64_t c, double a_f, int32_t null_every) {
  for (int64_t i = 0; i < n_rows; ++i) {
    const uint64_t row = (uint64_t)(row_offset + i);
    const uint64_t u = splitmix64(seed ^ (row * 0x9E3779B97F4A7C15ull));
    const bool is_null = null_every > 0 && (u >> 40) % (uint64_t)null_every == 0;
    switch (kind) {
      case MI355Q_GEN_I32_UNIFORM31:
        static_cast<int32_t*>(dst)[i] = is_null ? INT32_MIN : (int32_t)(u >> 33);
        break;
      case MI355Q_GEN_I32_MOD:
        static_cast<int32_t*>(dst)[i] =
            is_null ? INT32_MIN : (int32_t)((int64_t)(u % (uint64_t)a) + b);
        break;
      case MI355Q_GEN_I64_MOD:
        static_cast<int64_t*>(dst)[i] = is_null ? INT64_MIN : (int64_t)(u % (uint64_t)a) + b;
        break;
      case MI355Q_GEN_I64_MOD_MUL:
        static_cast<int64_t*>(dst)[i] =
            is_null ? INT64_MIN : (int64_t)(u % (uint64_t)a) * b + c;
        break;
      case MI355Q_GEN_F64_UNIT:
        static_cast<double*>(dst)[i] =
            is_null ? kNullDouble : (double)(u >> 11) * 0x1.0p-53 * a_f;
        break;
      default:
        return MI355Q_ERR_INVALID_PLAN;
    }
  }
  return 0;
}

// ================================================================ BASELINE-size runs
// The configurations of BASELINE.json are too large to be handed over as host arrays (1 B rows of
// cfg3-filtered = 20 GB), so this entry point generates every fragment inside the kernel thread that
// scans it, block by block, with the same counter-based generator as orc_generate_column, and
// otherwise runs the step exactly like orc_execute: one kernel per fragment with a private output
// buffer per host thread (Execute.cpp:3121-3153), then ResultSetStorage::reduce of the buffers in
// order (Execute.cpp:1772-1792).

namespace {

// ResultSetStorage::reduce on a row-wise baseline table with one 8-byte key is multi-threaded in the
// reference once `that` has more than 100000 entries (use_multithreaded_reduction,
// ResultSetReduction.cpp:43-45, :236-272): cpu_threads() workers over ranges of `that`'s entries,
// insertion into `this` by compare-and-swap of the key word with a write-pending sentinel
// (get_matching_group_value_reduction :697-737; fill_slots copies `that`'s slots into a claimed row).
// Every key occurs once in `that`, so a row of `this` is reduced by one thread only.
int32_t reduce_baseline_mt(const mi355q_qmd& q, int64_t* this_buf, const int64_t* that_buf, int n_threads) {
  const int rq = q.row_size / 8;
  const uint32_t ec = (uint32_t)q.entry_count;
  std::atomic<int32_t> err{0};
  auto worker = [&](int64_t lo, int64_t hi) {
    for (int64_t e = lo; e < hi; ++e) {
      const int64_t* that_row = that_buf + e * rq;
      const int64_t key = that_row[0];
      if (key == kEmptyKey64) continue;
      const uint32_t h = murmur3(&key, 8, 0) % ec;
      uint32_t hp = h;
      bool done = false;
      do {
        int64_t* row = this_buf + (size_t)hp * rq;
        int64_t expected = kEmptyKey64;
        if (__atomic_compare_exchange_n(row, &expected, kEmptyKey64 - 1, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
          for (int j = 1; j < rq; ++j) row[j] = that_row[j];  // fill_slots
          __atomic_store_n(row, key, __ATOMIC_SEQ_CST);
          done = true;
          break;
        }
        while (__atomic_load_n(row, __ATOMIC_SEQ_CST) == kEmptyKey64 - 1) {
        }
        if (__atomic_load_n(row, __ATOMIC_SEQ_CST) == key) {
          reduce_targets(q, row + 1, that_row + 1);
          done = true;
          break;
        }
        hp = (hp + 1) % ec;
      } while (hp != h);
      if (!done) {
        err.store(MI355Q_ERR_OUT_OF_SLOTS);
        return;
      }
    }
  };
  std::vector<std::thread> th;
  const int64_t per = (q.entry_count + n_threads - 1) / n_threads;
  for (int t = 0; t < n_threads; ++t) {
    const int64_t lo = t * per, hi = std::min<int64_t>(lo + per, q.entry_count);
    if (lo < hi) th.emplace_back(worker, lo, hi);
  }
  for (auto& t : th) t.join();
  return err.load();
}

inline double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

}  // namespace

struct orc_gen_spec {
  int32_t kind;        // MI355Q_GEN_*
  int32_t null_every;
  uint64_t seed;
  int64_t a, b, c;
  double a_f;
};

// timing[0] buffer initialisation (max over kernels), [1] row loops incl. generation (wall of the
// kernel phase minus [0]), [2] generation alone (max over threads of the time spent in the
// generator), [3] reduce of the per-kernel buffers.  reduce_threads: workers inside one reduce
// (the reference uses cpu_threads()); the buffers themselves are reduced in order.
ORC_EXPORT int32_t orc_execute_streamed(const mi355q_plan* plan, const orc_gen_spec* gens, int64_t total_rows,
                                        int64_t row_offset, int64_t frag_rows, int64_t block_rows,
                                        const void* const* inner_cols, int64_t inner_rows, const void* join,
                                        int32_t n_threads, int32_t reduce_threads, int64_t* out_buf,
                                        mi355q_qmd* out_qmd, double* timing) {
  ExecCtx c;
  c.stated = plan;
  if (int e = lower_plan(*plan, &c.lowered)) return e;
  c.plan = &c.lowered;
  if (int e = qmd_init(c.lowered, c.qmd)) return e;
  if (int e = build_targets(c.lowered, plan->n_group_cols > 0, c.ts)) return e;
  for (int i = 0; i < plan->n_targets; ++i) c.ts[i].slot = c.qmd.target_slot[i];
  c.join = static_cast<const OrcJoin*>(join);
  c.inner_cols = reinterpret_cast<const int8_t* const*>(inner_cols);
  c.inner_rows = inner_rows;
  if (plan->join_outer_col >= 0 && !c.join) return MI355Q_ERR_INVALID_PLAN;
  if (out_qmd) *out_qmd = c.qmd;
  if (frag_rows <= 0 || block_rows <= 0 || total_rows < 0) return MI355Q_ERR_INVALID_PLAN;
  const size_t quads = (size_t)(buffer_bytes(c.qmd) / 8);
  const int64_t n_frags = (total_rows + frag_rows - 1) / frag_rows;
  n_threads = (int32_t)std::max<int64_t>(1, std::min<int64_t>(n_threads, std::max<int64_t>(1, n_frags)));
  const int nc = plan->n_cols;
  auto width_of = [&](int col) -> size_t {
    switch (gens[col].kind) {
      case MI355Q_GEN_I32_UNIFORM31:
      case MI355Q_GEN_I32_MOD: return 4;
      default: return 8;
    }
  };

  std::vector<std::vector<int64_t>> bufs(n_threads);
  std::vector<int32_t> errs(n_threads, 0);
  std::vector<double> t_init(n_threads, 0.0), t_gen(n_threads, 0.0);
  const double t0 = now_s();
  auto worker = [&](int tid) {
    const double a0 = now_s();
    int64_t* buf = tid == 0 ? out_buf : (bufs[tid].resize(quads), bufs[tid].data());
    init_buffer(c.qmd, buf);
    t_init[tid] = now_s() - a0;
    std::vector<std::vector<int64_t>> block(nc);  // int64 storage keeps every column 8-byte aligned
    std::vector<const int8_t*> cols(nc);
    for (int col = 0; col < nc; ++col) {
      block[col].resize((size_t)((block_rows * (int64_t)width_of(col) + 7) / 8));
      cols[col] = reinterpret_cast<const int8_t*>(block[col].data());
    }
    for (int64_t f = tid; f < n_frags; f += n_threads) {
      const int64_t f_lo = f * frag_rows, f_hi = std::min(total_rows, f_lo + frag_rows);
      for (int64_t lo = f_lo; lo < f_hi; lo += block_rows) {
        const int64_t n = std::min(block_rows, f_hi - lo);
        const double g0 = now_s();
        for (int col = 0; col < nc; ++col) {
          const orc_gen_spec& g = gens[col];
          if (int e = orc_generate_column(block[col].data(), n, row_offset + lo, g.kind, g.seed, g.a, g.b, g.c, g.a_f,
                                          g.null_every)) {
            errs[tid] = e;
            return;
          }
        }
        t_gen[tid] += now_s() - g0;
        if (int32_t e = run_fragment(c, cols.data(), n, buf)) {
          errs[tid] = e;
          return;
        }
      }
    }
  };
  if (n_threads == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
    for (auto& t : th) t.join();
  }
  const double t1 = now_s();
  for (int t = 0; t < n_threads; ++t)
    if (errs[t]) return errs[t];
  const bool mt_reduce = reduce_threads > 1 && c.qmd.desc_type == MI355Q_GROUP_BY_BASELINE_HASH &&
                         !c.qmd.output_columnar && c.qmd.group_col_count == 1 && c.qmd.key_width == 8 &&
                         c.qmd.slot_width == 8 && c.qmd.entry_count > 100000;
  for (int t = 1; t < n_threads; ++t) {
    const int32_t e = mt_reduce ? reduce_baseline_mt(c.qmd, out_buf, bufs[t].data(), reduce_threads)
                                : reduce_buffers(c.qmd, out_buf, bufs[t].data());
    if (e) return e;
    std::vector<int64_t>().swap(bufs[t]);
  }
  const double t2 = now_s();
  if (timing) {
    timing[0] = *std::max_element(t_init.begin(), t_init.end());
    timing[1] = (t1 - t0) - timing[0];
    timing[2] = *std::max_element(t_gen.begin(), t_gen.end());
    timing[3] = t2 - t1;
  }
  return 0;
}

// ================================================================ ORDER BY
// ResultSet::sort (ResultSet.cpp:781-851) with ResultSetComparator::operator() (:1310-1470): the
// permutation of the non-empty entries ordered by the order entries in sequence — both NULL: next
// entry; one NULL: it goes first iff nulls_first (whatever the direction); integers and doubles by
// value, AVG as pair_to_double, `(lhs < rhs) != is_desc`; then top_n / offset.  out_perm receives
// entry indices; returns how many.  std::stable_sort: ties keep entry order (the reference's
// std::sort leaves them unspecified).
struct orc_order_entry {
  int32_t target_idx, descending, nulls_first, reserved;
};

ORC_EXPORT int64_t orc_sort(const mi355q_qmd* q, const int64_t* buf, const orc_order_entry* order, int32_t n_order,
                            int64_t limit, int64_t offset, int64_t* out_perm) {
  if (q->output_columnar) {
    const mi355q_qmd qr = rowwise_of(*q);
    const std::vector<int64_t> rows = col_to_rows(*q, buf);
    return orc_sort(&qr, rows.data(), order, n_order, limit, offset, out_perm);
  }
  const int rq = q->row_size / 8, kq = q->key_bytes / 8;
  struct Val {
    bool is_null, is_fp;
    int64_t i;
    double d;
  };
  auto value_of = [&](int64_t e, int t) -> Val {
    const int64_t* row = buf + e * rq;
    const int s = q->target_slot[t];
    Val v{false, false, 0, 0.0};
    if (q->target_agg[t] == MI355Q_PROJECT_KEY && s < 0) {
      const int ki = q->target_key_idx[t];
      v.i = q->key_width == 4 ? (int64_t) reinterpret_cast<const int32_t*>(row)[ki] : row[ki];
      v.is_null = v.i == q->target_null[t];
      return v;
    }
    const int64_t raw = q->slot_width == 4 ? (int64_t) reinterpret_cast<const int32_t*>(row + kq)[s] : row[kq + s];
    switch (q->target_agg[t]) {
      case MI355Q_AVG: {
        const int64_t cnt = row[kq + s + 1];
        v.is_fp = true;
        v.is_null = cnt == 0;  // isNull(pair): !i2
        if (!v.is_null)
          v.d = (q->target_arg_is_f32[t] ? (double)bits_flt((int32_t)raw) : q->target_arg_is_fp[t] ? bits_dbl(raw) : (double)raw) /
                (double)cnt;
        return v;
      }
      case MI355Q_COUNT:
      case MI355Q_COUNT_IF:
        v.i = raw;
        return v;
      default:
        if (q->target_arg_is_f32[t]) {
          v.is_fp = true;
          v.d = (double)bits_flt((int32_t)raw);
          v.is_null = q->target_skip_null[t] && (int32_t)raw == (int32_t)q->target_null[t];
        } else if (q->target_is_fp[t]) {
          v.is_fp = true;
          v.d = bits_dbl(raw);
          v.is_null = q->target_skip_null[t] && raw == q->target_null[t];
        } else {
          v.i = raw;
          v.is_null = (q->target_skip_null[t] || q->target_agg[t] == MI355Q_PROJECT_KEY) && raw == q->target_null[t];
        }
        return v;
    }
  };
  std::vector<int64_t> perm;
  for (int64_t e = 0; e < q->entry_count; ++e)
    if (!is_empty_entry(*q, buf, e)) perm.push_back(e);
  std::stable_sort(perm.begin(), perm.end(), [&](int64_t lhs, int64_t rhs) {
    for (int o = 0; o < n_order; ++o) {
      const orc_order_entry& oe = order[o];
      const Val a = value_of(lhs, oe.target_idx), b = value_of(rhs, oe.target_idx);
      if (a.is_null && b.is_null) continue;
      if (a.is_null) return oe.nulls_first != 0;
      if (b.is_null) return oe.nulls_first == 0;
      if (a.is_fp) {
        if (a.d == b.d) continue;
        return (a.d < b.d) != (oe.descending != 0);
      }
      if (a.i == b.i) continue;
      return (a.i < b.i) != (oe.descending != 0);
    }
    return false;
  });
  int64_t n = (int64_t)perm.size() - offset;
  if (n < 0) n = 0;
  if (limit > 0 && n > limit) n = limit;
  for (int64_t i = 0; i < n; ++i) out_perm[i] = perm[(size_t)(offset + i)];
  return n;
}
