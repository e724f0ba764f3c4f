// fast_common.h — pieces shared by the specialised kernel families (kernels_fast.hip,
// kernels_part.hip): the 16-byte-per-lane streaming skeleton, the normalised range filter, the
// slot programme, and the plan-time shape test.  gfx950 wave64 only.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels.h"
#include "rowfunc.h"

namespace mq {
namespace fast {

constexpr int kBlock = 256;

struct none_t {};

template <typename T>
struct is_none { static constexpr bool value = false; };
template <>
struct is_none<none_t> { static constexpr bool value = true; };

// ---- 4-row vector loads -------------------------------------------------------------
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef long long v2i64 __attribute__((ext_vector_type(2)));

template <typename T>
struct Quad {
  T v[4];
};
template <>
struct Quad<none_t> {};

// Column chunks are reached through a pointer table in memory, so the compiler only knows them as
// generic pointers and would emit FLAT loads: those count against lgkmcnt as well as vmcnt, i.e.
// every wait for an LDS result (the scatter's staging protocol, the LDS tables) also waits for the
// prefetched tile.  The chunks are device (global) memory by contract: say so, and the loads become
// global_load_dwordx4 ... nt with a scalar base.
#define MQ_GLOBAL __attribute__((address_space(1)))
template <typename T>
MQ_D void load_quad(const int8_t* base, int64_t quad, Quad<T>& q);
template <>
MQ_D void load_quad<int32_t>(const int8_t* base, int64_t quad, Quad<int32_t>& q) {
  const v4i32 x = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad);
  q.v[0] = x.x; q.v[1] = x.y; q.v[2] = x.z; q.v[3] = x.w;
}
template <>
MQ_D void load_quad<int64_t>(const int8_t* base, int64_t quad, Quad<int64_t>& q) {
  const v2i64 a = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + quad * 2);
  const v2i64 b = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + quad * 2 + 1);
  q.v[0] = a.x; q.v[1] = a.y; q.v[2] = b.x; q.v[3] = b.y;
}
template <>
MQ_D void load_quad<double>(const int8_t* base, int64_t quad, Quad<double>& q) {
  const v2i64 a = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + quad * 2);
  const v2i64 b = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + quad * 2 + 1);
  q.v[0] = bits_dbl(a.x); q.v[1] = bits_dbl(a.y);
  q.v[2] = bits_dbl(b.x); q.v[3] = bits_dbl(b.y);
}
// a 1-byte column (TINYINT / BOOLEAN; the row mask of a compiled filter, kernels_filter.hip): four rows = one 4-byte load
template <>
MQ_D void load_quad<int8_t>(const int8_t* base, int64_t quad, Quad<int8_t>& q) {
  const uint32_t x = __builtin_nontemporal_load((const MQ_GLOBAL uint32_t*)base + quad);
  q.v[0] = (int8_t)x; q.v[1] = (int8_t)(x >> 8); q.v[2] = (int8_t)(x >> 16); q.v[3] = (int8_t)(x >> 24);
}
template <>
MQ_D void load_quad<none_t>(const int8_t*, int64_t, Quad<none_t>&) {}

template <typename T>
MQ_D T load_one(const int8_t* base, int64_t pos) { return ((const MQ_GLOBAL T*)base)[pos]; }
template <>
MQ_D none_t load_one<none_t>(const int8_t*, int64_t) { return none_t{}; }

template <typename T>
MQ_D T quad_get(const Quad<T>& q, int i) { return q.v[i]; }
template <>
MQ_D none_t quad_get<none_t>(const Quad<none_t>&, int) { return none_t{}; }
template <typename T>
MQ_D void quad_set(Quad<T>& q, int i, T v) { q.v[i] = v; }
template <>
MQ_D void quad_set<none_t>(Quad<none_t>&, int, none_t) {}

// Filter normalised at plan time to  lo <= v <= hi  (optionally negated for <>), plus the
// NULL exclusion of DEF_CMP_NULLABLE (RuntimeFunctions.cpp:73-83).
struct RangeFilter {
  int64_t lo, hi;
  int32_t negate, nullable;
  int64_t null_val;
  int32_t col;
};
template <typename T>
MQ_D bool filter_pass(const RangeFilter& f, T v) {
  const int64_t x = (int64_t)v;
  bool in = x >= f.lo && x <= f.hi;
  if (f.negate) in = !in;
  if (f.nullable && x == f.null_val) in = false;
  return in;
}
template <>
MQ_D bool filter_pass<none_t>(const RangeFilter&, none_t) { return true; }

// Streaming skeleton.  fn(filter_value, key, val) is called for every row.
//
// Access pattern (measured on MI355X, tools/microbench/stream.hip): a workgroup walks
// contiguous tiles of blockDim x UQ quads, every lane keeps UQ independent 16-byte
// non-temporal loads per column in flight, and the grid is only ~2 workgroups of 256 lanes
// per CU — 7.0-7.1 TB/s read bandwidth versus 5.5-6.0 TB/s for a grid-strided walk with 8
// workgroups per CU (fewer, longer streams keep HBM pages open).
template <typename FT, typename KT, typename VT, int UQ = 2, typename Fn>
MQ_D void scan_fragments(const int8_t* const* __restrict__ cols, const int64_t* __restrict__ num_rows,
                         int n_frags, int n_cols, int fcol, int kcol, int vcol, Fn&& fn) {
  const int64_t tile_q = (int64_t)blockDim.x * UQ;  // quads per tile
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * n_cols;
    const int8_t* fb = is_none<FT>::value ? nullptr : fc[fcol];
    const int8_t* kb = is_none<KT>::value ? nullptr : fc[kcol];
    const int8_t* vb = is_none<VT>::value ? nullptr : fc[vcol];
    const int64_t n = num_rows[f];
    const int64_t nq = n >> 2;
    const int64_t n_tiles = nq / tile_q;
    // rotate the starting workgroup per fragment so short fragments still spread over the grid
    for (int64_t t = (blockIdx.x + (int64_t)f * 7) % gridDim.x; t < n_tiles; t += gridDim.x) {
      const int64_t q0 = t * tile_q + threadIdx.x;
      Quad<FT> fq[UQ];
      Quad<KT> kq[UQ];
      Quad<VT> vq[UQ];
#pragma unroll
      for (int u = 0; u < UQ; ++u) load_quad<FT>(fb, q0 + (int64_t)u * blockDim.x, fq[u]);
#pragma unroll
      for (int u = 0; u < UQ; ++u) load_quad<KT>(kb, q0 + (int64_t)u * blockDim.x, kq[u]);
#pragma unroll
      for (int u = 0; u < UQ; ++u) load_quad<VT>(vb, q0 + (int64_t)u * blockDim.x, vq[u]);
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fn(quad_get(fq[u], i), quad_get(kq[u], i), quad_get(vq[u], i));
      }
    }
    // ragged end of the fragment: whole quads past the last full tile, then < 4 rows
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t gsize = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = n_tiles * tile_q + gtid; q < nq; q += gsize) {
      Quad<FT> f0;
      Quad<KT> k0;
      Quad<VT> v0;
      load_quad<FT>(fb, q, f0);
      load_quad<KT>(kb, q, k0);
      load_quad<VT>(vb, q, v0);
#pragma unroll
      for (int i = 0; i < 4; ++i) fn(quad_get(f0, i), quad_get(k0, i), quad_get(v0, i));
    }
    const int64_t tail = (nq << 2) + gtid;
    if (tail < n) {
      fn(load_one<FT>(fb, tail), load_one<KT>(kb, tail), load_one<VT>(vb, tail));
    }
  }
}

MQ_D unsigned long long wave_sum_u64(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
MQ_D long long wave_sum_i64(long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Slot programme shared by the LDS and baseline families: which op each 8-byte slot takes.
enum SlotOp : int32_t { SO_COUNT = 0, SO_SUM_I = 1, SO_SUM_F = 2, SO_MIN_I = 3, SO_MAX_I = 4,
                        SO_MIN_F = 5, SO_MAX_F = 6, SO_KEY = 7,
                        SO_COUNT_NN = 8 };  // rows whose value is not NULL (COUNT(col), AVG's count)
struct SlotProg {
  int32_t n;                         // slots
  int32_t op[MI355Q_MAX_SLOTS];
  // nullable value column (the reference's *_skip_val aggregates, RuntimeFunctions.cpp:1313-1431,
  // :1558-1584): NULL inputs are skipped by every value op; slots flagged null_init start at
  // the NULL sentinel, the first non-NULL value overwrites it (SUM / MIN / MAX; AVG's sum
  // starts at 0, OutputBufferInitialization.cpp:132-289)
  int32_t val_nullable;
  int32_t null_init[MI355Q_MAX_SLOTS];
  int64_t null_bits;                 // bit pattern of the value column's NULL as loaded
  // the sentinel each null_init slot starts at (its init value): SUM(int) starts at NULL_BIGINT, MIN / MAX at the
  // ARGUMENT type's NULL — different values for a nullable INT column aggregated both ways (the reference's own
  // benchmark: count / sum / max / min / avg of one nullable INT column, PerfectHashSingleCol/PHS001.sql)
  int64_t slot_null[MI355Q_MAX_SLOTS];
};

// one row's update of a group's slots in the output table.  fval / ival: the value as double /
// int64; vbits: its bit pattern (NULL test)
MQ_D void apply_slots_global(const SlotProg& sp, int64_t* slots, double fval, int64_t ival, int64_t vbits) {
  const bool is_null = sp.val_nullable && vbits == sp.null_bits;
  for (int j = 0; j < sp.n; ++j) {
    int64_t* s = slots + j;
    if (sp.op[j] == SO_COUNT) {
      atomicAdd((unsigned long long*)s, 1ull);
      continue;
    }
    if (is_null) continue;  // every other op reads the value
    if (sp.null_init[j]) {
      switch (sp.op[j]) {
        case SO_SUM_I: a_sum_i64_skip<true>(s, ival, sp.slot_null[j]); break;
        case SO_SUM_F: a_sum_f64_skip<true>(s, fval, bits_dbl(sp.slot_null[j])); break;
        case SO_MIN_I: a_min_i64_skip<true>(s, ival, sp.slot_null[j]); break;
        case SO_MAX_I: a_max_i64_skip<true>(s, ival, sp.slot_null[j]); break;
        case SO_MIN_F: a_minmax_f64<true, false, true>(s, fval, bits_dbl(sp.slot_null[j])); break;
        case SO_MAX_F: a_minmax_f64<true, true, true>(s, fval, bits_dbl(sp.slot_null[j])); break;
        default: break;
      }
      continue;
    }
    switch (sp.op[j]) {
      case SO_COUNT_NN: atomicAdd((unsigned long long*)s, 1ull); break;
      case SO_SUM_I: atomicAdd((unsigned long long*)s, (unsigned long long)ival); break;
      case SO_SUM_F: atomicAdd((double*)s, fval); break;
      case SO_MIN_I: atomicMin((long long*)s, (long long)ival); break;
      case SO_MAX_I: atomicMax((long long*)s, (long long)ival); break;
      case SO_MIN_F: a_minmax_f64<true, false, false>(s, fval, 0.0); break;
      case SO_MAX_F: a_minmax_f64<true, true, false>(s, fval, 0.0); break;
      default: break;
    }
  }
}

template <typename VT>
MQ_D int64_t as_bits(VT v) { return (int64_t)v; }
template <>
MQ_D int64_t as_bits<double>(double v) { return dbl_bits(v); }
template <>
MQ_D int64_t as_bits<none_t>(none_t) { return 0; }
template <typename VT>
MQ_D double as_f64(VT v) { return (double)v; }
template <>
MQ_D double as_f64<none_t>(none_t) { return 0.0; }
template <typename VT>
MQ_D int64_t as_i64(VT v) { return (int64_t)v; }
template <>
MQ_D int64_t as_i64<none_t>(none_t) { return 0; }

// ---------------------------------------------------------------------------- host side
inline bool all_aligned16(const FragView& fv, int col) {
  for (int f = 0; f < fv.n_frags; ++f) {
    if (((uintptr_t)fv.h_cols[(size_t)f * fv.n_cols + col]) & 15) return false;
  }
  return true;
}

// Turn `col <op> literal` on an integer column into an inclusive range (+ negate for <>).
// allow_int8: the caller's kernel also filters on a plain 1-byte column (TINYINT / BOOLEAN — and the row mask a compiled
// filter's pre-pass leaves, kernels_filter.hip): four rows are one 4-byte load
inline bool make_range_filter(const DevQual& q, RangeFilter* f, bool allow_int8 = false) {
  if (q.or_group != 0) return false;  // a member of a disjunction is not a conjunct (those plans take the row kernel)
  if (q.type != MI355Q_INT32 && q.type != MI355Q_INT64 && !(allow_int8 && q.type == MI355Q_INT8)) return false;
  f->col = q.col;
  f->negate = 0;
  f->nullable = q.nullable;
  f->null_val = int_null_of(q.type);
  const int64_t tmin = q.type == MI355Q_INT8 ? (int64_t)INT8_MIN : q.type == MI355Q_INT32 ? (int64_t)INT32_MIN : INT64_MIN;
  const int64_t tmax = q.type == MI355Q_INT8 ? (int64_t)INT8_MAX : q.type == MI355Q_INT32 ? (int64_t)INT32_MAX : INT64_MAX;
  const int64_t x = q.ival;
  switch (q.op) {
    case MI355Q_EQ: f->lo = x; f->hi = x; break;
    case MI355Q_NE: f->lo = x; f->hi = x; f->negate = 1; break;
    case MI355Q_LT:
      if (x == INT64_MIN) { f->lo = 1; f->hi = 0; } else { f->lo = tmin; f->hi = x - 1; }
      break;
    case MI355Q_LE: f->lo = tmin; f->hi = x; break;
    case MI355Q_GT:
      if (x == INT64_MAX) { f->lo = 1; f->hi = 0; } else { f->lo = x + 1; f->hi = tmax; }
      break;
    case MI355Q_GE: f->lo = x; f->hi = tmax; break;
    // x IS NOT NULL: every value but the sentinel (all of them on a NOT NULL column);
    // x IS NULL: the sentinel alone — and nothing on a NOT NULL column (codegenIsNull)
    case MI355Q_IS_NOT_NULL: f->lo = tmin; f->hi = tmax; break;
    case MI355Q_IS_NULL:
      if (q.nullable) { f->lo = f->null_val; f->hi = f->null_val; f->nullable = 0; }
      else { f->lo = 1; f->hi = 0; }
      break;
    default: return false;
  }
  // bounds inside the column's type (members that compare an INT32 column in 32 bits truncate them): a literal outside
  // the type — `i32 = 5000000000`, `i32 <= 5000000000` — leaves the empty range or the type's own bound, never a wrapped
  // one (a negated empty range passes every non-NULL row, as `i32 <> 5000000000` does)
  if (f->lo > f->hi || f->lo > tmax || f->hi < tmin) {
    f->lo = 1;
    f->hi = 0;
  } else {
    if (f->lo < tmin) f->lo = tmin;
    if (f->hi > tmax) f->hi = tmax;
  }
  return true;
}

// Two range quals on ONE column (x > 6 AND x < 8; the two bounds a dense join leaves on its key) are one range: the
// families load a filter column once per RangeFilter.  Returns the new count.
inline int merge_range_filters(RangeFilter* f, int32_t* type, int n) {
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n;) {
      if (f[i].col == f[j].col && type[i] == type[j] && !f[i].negate && !f[j].negate && f[i].null_val == f[j].null_val) {
        f[i].lo = f[j].lo > f[i].lo ? f[j].lo : f[i].lo;
        f[i].hi = f[j].hi < f[i].hi ? f[j].hi : f[i].hi;
        f[i].nullable = f[i].nullable || f[j].nullable;
        for (int k = j; k + 1 < n; ++k) {
          f[k] = f[k + 1];
          type[k] = type[k + 1];
        }
        --n;
      } else {
        ++j;
      }
    }
  return n;
}

inline RangeFilter no_filter() {
  RangeFilter f{};
  f.lo = INT64_MIN;
  f.hi = INT64_MAX;
  f.col = 0;
  return f;
}

// Common shape test for the grouped fast families: at most one integer qual, every value
// aggregate on ONE NOT NULL column of type int64/double, COUNT(*) and key projections free.
struct FastShape {
  int fil_type = 0;   // 0 none / MI355Q_INT32 / MI355Q_INT64 / MI355Q_INT8 (every dispatch on it names all four)
  RangeFilter flt;
  int vcol = -1, vtype = 0;
  SlotProg sp;
};

// Value columns the fast families load natively: plain int64 / double / int32 chunks, and
// kENCODING_FIXED(32) columns — the decoder of those is the same sign-extending 4-byte load, and
// their NULL (the storage sentinel) is recognised before widening.
inline int fast_value_type(int code) {
  if (code == MI355Q_INT64 || code == MI355Q_DOUBLE || code == MI355Q_INT32) return code;
  if (tc_enc(code) == MI355Q_ENC_FIXED && tc_storage(code) == MI355Q_INT32) return MI355Q_INT32;
  return 0;
}

inline bool grouped_fast_shape(const DevPlan& p, const FragView& fv, FastShape* s) {
  if (p.join_col >= 0 || p.n_quals > 1 || p.n_group > 1 || p.col0_key_quirk) return false;
  // a NULL group key: the baseline layout keeps the sentinel as an ordinary key; the perfect
  // layout translates it (max + 1) and is left to the generic family
  if (p.group_nullable && p.desc_type != MI355Q_GROUP_BY_BASELINE_HASH) return false;
  s->flt = no_filter();
  if (p.n_quals == 1) {
    if (!make_range_filter(p.quals[0], &s->flt, true)) return false;  // (INT32 / INT64, or a 1-byte column: the filter mask)
    s->fil_type = p.quals[0].type;
    if (!all_aligned16(fv, p.quals[0].col)) return false;
  }
  s->sp.n = p.slot_count;
  s->sp.val_nullable = 0;
  s->sp.null_bits = 0;
  for (int i = 0; i < MI355Q_MAX_SLOTS; ++i) {
    s->sp.op[i] = SO_COUNT;
    s->sp.null_init[i] = 0;
    s->sp.slot_null[i] = 0;
  }
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    if (t.table != 0) return false;
    if (t.agg == MI355Q_PROJECT_KEY) {
      if (t.slot >= 0) s->sp.op[t.slot] = SO_KEY;
      continue;
    }
    if (t.agg == MI355Q_COUNT && (t.col < 0 || !t.skip_null)) {
      s->sp.op[t.slot] = SO_COUNT;  // COUNT(*) / COUNT(NOT NULL col)
      continue;
    }
    if (t.col < 0) return false;
    // every value aggregate (and COUNT(nullable col)) reads ONE int64 / int32 / double column
    const int vt = fast_value_type(t.arg_type);
    if (!vt) return false;
    if (s->vcol >= 0 && s->vcol != t.col) return false;
    s->vcol = t.col;
    s->vtype = vt;
    if (t.skip_null) {
      s->sp.val_nullable = 1;
      // the sentinel as the kernel loads it: the STORAGE type's (an int32 chunk is widened by
      // sign extension, so NULL_INT arrives as (int64)INT32_MIN)
      s->sp.null_bits = vt == MI355Q_DOUBLE ? kNullDoubleBits : vt == MI355Q_INT32 ? (int64_t)INT32_MIN : INT64_MIN;
    }
    const bool fp = t.arg_fp;
    switch (t.agg) {
      case MI355Q_COUNT: s->sp.op[t.slot] = SO_COUNT_NN; break;
      case MI355Q_SUM:
        s->sp.op[t.slot] = fp ? SO_SUM_F : SO_SUM_I;
        s->sp.null_init[t.slot] = t.skip_null;
        break;
      case MI355Q_AVG:
        s->sp.op[t.slot] = fp ? SO_SUM_F : SO_SUM_I;
        s->sp.op[t.slot + 1] = t.skip_null ? SO_COUNT_NN : SO_COUNT;
        break;
      case MI355Q_MIN:
        s->sp.op[t.slot] = fp ? SO_MIN_F : SO_MIN_I;
        s->sp.null_init[t.slot] = t.skip_null;
        break;
      case MI355Q_MAX:
        s->sp.op[t.slot] = fp ? SO_MAX_F : SO_MAX_I;
        s->sp.null_init[t.slot] = t.skip_null;
        break;
      default: return false;
    }
  }
  // every slot that starts at a NULL sentinel carries its own (SUM(int): NULL_BIGINT; MIN / MAX: the argument
  // type's NULL)
  for (int j = 0; j < p.slot_count; ++j)
    s->sp.slot_null[j] = s->sp.null_init[j] ? p.init_vals[j] : s->sp.null_bits;
  // one column cannot be nullable for one target and NOT NULL for another
  if (s->sp.val_nullable) {
    for (int i = 0; i < p.n_targets; ++i) {
      const DevTarget& t = p.targets[i];
      if (t.table == 0 && t.col == s->vcol && t.agg != MI355Q_PROJECT_KEY && !t.skip_null) return false;
    }
  }
  if (s->vcol >= 0 && !all_aligned16(fv, s->vcol)) return false;
  if (!all_aligned16(fv, p.group_col)) return false;
  return true;
}


inline void rec(hipEvent_t e, hipStream_t s) {
  if (e) (void)hipEventRecord(e, s);
}

}  // namespace fast
}  // namespace mq
