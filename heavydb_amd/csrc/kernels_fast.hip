// kernels_fast.hip — the specialised members of the kernel family, selected at plan time.
//
// All of them share one streaming skeleton written for gfx950: every lane reads FOUR
// consecutive rows of each column it needs with 16-byte loads (int32 -> one dwordx4, 8-byte
// types -> two), a wave covers 256 consecutive rows per step, and the grid strides over
// 4-row "quads" of each fragment (the fragment loop of multifrag_query_hoisted_literals,
// RuntimeFunctions.cpp:2434-2471, lives inside the kernel).  Work is integer/hash/reduction:
// HBM-bound, no MFMA.
//
//   scan_count     COUNT(*) WHERE col <op> k         (SURVEY cfg1)  per-lane popcount ->
//                  wave shuffle reduce -> one atomic per block
//   perfect_lds    GROUP BY small-range int key      (SURVEY cfg2)  per-block LDS table
//                  (ds atomics), flushed once per block with global atomics — the reference's
//                  shared-memory group-by idea (GpuSharedMemoryUtils.cpp) without the JIT
//   baseline       GROUP BY high-cardinality int64   (SURVEY cfg3)  direct: global CAS insert
//                  + atomics; partitioned: hash-partition rows into LDS-sized key ranges,
//                  then aggregate each range in LDS and emit each group once
//   join_sum       fact JOIN dim ... SUM/COUNT       (SURVEY cfg4)  probe fused into the scan
#include <cstring>
#include <type_traits>

#include "boolfilter.h"
#include "fast_common.h"

namespace mq {

using namespace fast;

namespace {

// =========================================================================== scan_count
template <typename FT>
__global__ __launch_bounds__(kBlock) void k_scan_count(const int8_t* const* __restrict__ cols,
                                                        const int64_t* __restrict__ num_rows,
                                                        int n_frags, int n_cols, RangeFilter flt,
                                                        int64_t* __restrict__ out) {
  unsigned long long cnt = 0;
  scan_fragments<FT, none_t, none_t, 8>(cols, num_rows, n_frags, n_cols, flt.col, 0, 0,
                                     [&](FT fv, none_t, none_t) { cnt += filter_pass<FT>(flt, fv); });
  cnt = wave_sum_u64(cnt);
  __shared__ unsigned long long s_part[kBlock / 64];
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int i = 0; i < kBlock / 64; ++i) t += s_part[i];
    if (t) atomicAdd((unsigned long long*)out, t);
  }
}


// =========================================================================== scan_agg
// Non-grouped aggregates over SEVERAL columns with up to four range quals — the shape of the reference's own
// NonGroupedAgg benchmark (Benchmarks/synthetic_benchmark/queries/NonGroupedAgg/NGA01-05.sql: six columns, one
// aggregate kind each; query_template's register accumulators, QueryTemplateGenerator.cpp:265-549).  Same
// streaming skeleton as k_scan_count (a workgroup walks contiguous tiles, every lane keeps UQ independent
// 16-byte loads per column in flight), one set of typed register accumulators per COLUMN — rows with a value,
// sum, min, max — whatever the targets over that column need; a wave shuffle reduce, one LDS fold per
// workgroup, and the workgroup's partial row is merged into the output with the reduce rule (reduce_target: the
// _skip_val forms non-grouped aggregates always take, TargetExprBuilder.cpp:684-690).
constexpr int kScanAggCols = 8;
struct ScanAggArgs {
  int32_t n_used, n_flt;
  int32_t col[kScanAggCols], type[kScanAggCols], nullable[kScanAggCols];  // type: MI355Q_INT32 / _INT64 / _DOUBLE (plain)
  RangeFilter flt[MI355Q_MAX_QUALS];
  int32_t flt_type[MI355Q_MAX_QUALS];
  int32_t target_cslot[MI355Q_MAX_TARGETS];  // index into col[] of each target's argument, -1 = COUNT(*)
  // a filter compiled at plan time (boolfilter.h) instead of range quals: flt[k].col / flt_type[k] name its columns
  int32_t bf_on, pad_bf_;
  const BoolFilter* bf;  // DEVICE memory
};

struct ColAcc {
  unsigned long long cnt;  // rows whose value is not NULL (and passed the quals)
  int64_t sum_i, min_i, max_i;
  double sum_f, min_f, max_f;
};

// a quad of any column as loaded: 16 bytes (4-byte types) or 32 bytes (8-byte types); the loads of a step are
// all issued before the first value is looked at (wave-uniform branches do not wait for memory)
struct RawQuad {
  v4i32 lo, hi;
};
MQ_D void load_raw(const int8_t* base, int64_t quad, bool w8, RawQuad& r) {
  if (w8) {
    r.lo = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2);
    r.hi = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2 + 1);
  } else {
    r.lo = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad);
  }
}
MQ_D int64_t raw_i64(const RawQuad& r, int i) {
  const v4i32& h = i < 2 ? r.lo : r.hi;
  const int j = (i & 1) * 2;
  return (int64_t)(((uint64_t)(uint32_t)(j ? h.w : h.y) << 32) | (uint64_t)(uint32_t)(j ? h.z : h.x));
}
MQ_D int32_t raw_i32(const RawQuad& r, int i) { return i == 0 ? r.lo.x : i == 1 ? r.lo.y : i == 2 ? r.lo.z : r.lo.w; }

MQ_D void scan_agg_apply(ColAcc& a, const RawQuad& r, int type, uint32_t pass, bool nullable) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bool ok = (pass >> i) & 1u;
    if (type == MI355Q_DOUBLE) {
      const double v = bits_dbl(raw_i64(r, i));
      ok = ok && !(nullable && v == kNullDouble);
      if (ok) {
        a.cnt += 1;
        a.sum_f += v;
        a.min_f = v < a.min_f ? v : a.min_f;
        a.max_f = a.max_f < v ? v : a.max_f;
      }
    } else {
      const int64_t x = type == MI355Q_INT32 ? (int64_t)raw_i32(r, i) : raw_i64(r, i);
      ok = ok && !(nullable && x == (type == MI355Q_INT32 ? (int64_t)INT32_MIN : INT64_MIN));
      if (ok) {
        a.cnt += 1;
        a.sum_i += x;
        a.min_i = x < a.min_i ? x : a.min_i;
        a.max_i = a.max_i < x ? x : a.max_i;
      }
    }
  }
}
MQ_D uint32_t scan_agg_filter(const RangeFilter& f, int type, const RawQuad& r) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool p = type == MI355Q_INT32 ? filter_pass<int32_t>(f, raw_i32(r, i))
                   : type == MI355Q_INT8 ? filter_pass<int32_t>(f, (int32_t)(int8_t)((uint32_t)r.lo.x >> (8 * i)))  // (four 1-byte rows in lo.x)
                                         : filter_pass<int64_t>(f, raw_i64(r, i));
    m |= (p ? 1u : 0u) << i;
  }
  return m;
}

template <typename T>
MQ_D void scan_agg_one(ColAcc& a, const int8_t* base, int64_t pos, bool nullable) {
  const T v = load_one<T>(base, pos);
  if constexpr (std::is_same<T, double>::value) {
    if (nullable && v == kNullDouble) return;
    a.cnt += 1;
    a.sum_f += v;
    a.min_f = v < a.min_f ? v : a.min_f;
    a.max_f = a.max_f < v ? v : a.max_f;
  } else {
    const int64_t x = (int64_t)v;
    if (nullable && x == (std::is_same<T, int32_t>::value ? (int64_t)INT32_MIN : INT64_MIN)) return;
    a.cnt += 1;
    a.sum_i += x;
    a.min_i = x < a.min_i ? x : a.min_i;
    a.max_i = a.max_i < x ? x : a.max_i;
  }
}

MQ_D double wave_sum_f64(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
MQ_D long long wave_min_i64(long long v) {
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_down(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
MQ_D long long wave_max_i64(long long v) {
  for (int off = 32; off > 0; off >>= 1) {
    const long long o = __shfl_down(v, off, 64);
    v = v < o ? o : v;
  }
  return v;
}
MQ_D double wave_min_f64(double v) {
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}
MQ_D double wave_max_f64(double v) {
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(v, off, 64);
    v = v < o ? o : v;
  }
  return v;
}

// NC / NF: columns / quals this member holds registers for (the step's register footprint is UQ x (NC + NF) raw quads)
template <int UQ, int NC, int NF>
__global__ __launch_bounds__(kBlock) void k_scan_agg(const int8_t* const* __restrict__ cols,
                                                      const int64_t* __restrict__ num_rows, int n_frags, int n_cols,
                                                      ScanAggArgs a, DevPlan p, int64_t* __restrict__ out) {
  __shared__ BoolFilter s_bf;  // the compiled filter: atoms + truth table (a.bf_on)
  if (a.bf_on) {
    bf_load(a.bf, &s_bf, threadIdx.x, kBlock);
    __syncthreads();
  }
  ColAcc acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    acc[c].cnt = 0;
    acc[c].sum_i = 0;
    acc[c].min_i = INT64_MAX;
    acc[c].max_i = INT64_MIN;
    acc[c].sum_f = 0.0;
    acc[c].min_f = 1.7976931348623157e308;
    acc[c].max_f = -1.7976931348623157e308;
  }
  unsigned long long rows_passing = 0;
  const int64_t tile_q = (int64_t)kBlock * UQ;
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * n_cols;
    const int64_t n = num_rows[f];
    const int64_t nq = n >> 2;
    const int64_t n_tiles = nq / tile_q;
    // the fragment's chunk of every stream, fetched once
    const int8_t *fbase[NF], *cbase[NC];
#pragma unroll
    for (int k = 0; k < NF; ++k) fbase[k] = k < a.n_flt ? fc[a.flt[k].col] : nullptr;
#pragma unroll
    for (int c = 0; c < NC; ++c) cbase[c] = c < a.n_used ? fc[a.col[c]] : nullptr;
    // one step = UQ quads of every column: phase 1 issues every load, phase 2 filters and accumulates
    auto do_step = [&](int64_t q0, int n_quads, int64_t stride) {
      RawQuad fr[NF][UQ];
      RawQuad cr[NC][UQ];
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        if (u >= n_quads) break;
        const int64_t quad = q0 + (int64_t)u * stride;
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          if (k >= a.n_flt) break;
          if (a.flt_type[k] == MI355Q_INT8) fr[k][u].lo.x = (int)__builtin_nontemporal_load((const MQ_GLOBAL uint32_t*)fbase[k] + quad);
          else load_raw(fbase[k], quad, a.flt_type[k] != MI355Q_INT32, fr[k][u]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (c >= a.n_used) break;
          load_raw(cbase[c], quad, a.type[c] != MI355Q_INT32, cr[c][u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        if (u >= n_quads) break;
        uint32_t pass = 15u;
        if (a.bf_on) {  // atoms on the filter columns' values + one bit of the truth table per row
          pass = 0u;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int64_t fval[NF];
#pragma unroll
            for (int k = 0; k < NF; ++k)
              fval[k] = k >= a.n_flt ? 0 : a.flt_type[k] == MI355Q_INT32 ? (int64_t)raw_i32(fr[k][u], i) : raw_i64(fr[k][u], i);
            pass |= (bf_row_passes<NF>(s_bf, fval) ? 1u : 0u) << i;
          }
        } else {
#pragma unroll
          for (int k = 0; k < NF; ++k) {
            if (k >= a.n_flt) break;
            pass &= scan_agg_filter(a.flt[k], a.flt_type[k], fr[k][u]);
          }
        }
        rows_passing += __popc(pass);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (c >= a.n_used) break;
          scan_agg_apply(acc[c], cr[c][u], a.type[c], pass, a.nullable[c] != 0);
        }
      }
    };
    for (int64_t t = (blockIdx.x + (int64_t)f * 7) % gridDim.x; t < n_tiles; t += gridDim.x)
      do_step(t * tile_q + threadIdx.x, UQ, kBlock);
    for (int64_t q = n_tiles * tile_q + gtid; q < nq; q += gsize) do_step(q, 1, 0);
    const int64_t tail = (nq << 2) + gtid;
    if (tail < n) {
      bool pass = true;
      if (a.bf_on) {
        int64_t fval[NF];
#pragma unroll
        for (int k = 0; k < NF; ++k)
          fval[k] = k >= a.n_flt ? 0 : a.flt_type[k] == MI355Q_INT32 ? (int64_t)load_one<int32_t>(fbase[k], tail) : load_one<int64_t>(fbase[k], tail);
        pass = bf_row_passes<NF>(s_bf, fval);
      } else {
        for (int k = 0; k < a.n_flt; ++k) {
          pass = pass && (a.flt_type[k] == MI355Q_INT32 ? filter_pass<int32_t>(a.flt[k], load_one<int32_t>(fc[a.flt[k].col], tail))
                          : a.flt_type[k] == MI355Q_INT8 ? filter_pass<int32_t>(a.flt[k], (int32_t)load_one<int8_t>(fc[a.flt[k].col], tail))
                                                         : filter_pass<int64_t>(a.flt[k], load_one<int64_t>(fc[a.flt[k].col], tail)));
        }
      }
      if (pass) {
        rows_passing += 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (c >= a.n_used) break;
          const int8_t* base = fc[a.col[c]];
          if (a.type[c] == MI355Q_INT32) scan_agg_one<int32_t>(acc[c], base, tail, a.nullable[c] != 0);
          else if (a.type[c] == MI355Q_INT64) scan_agg_one<int64_t>(acc[c], base, tail, a.nullable[c] != 0);
          else scan_agg_one<double>(acc[c], base, tail, a.nullable[c] != 0);
        }
      }
    }
  }
  // wave reduce, then one fold per workgroup in LDS
  __shared__ unsigned long long s_rows[kBlock / 64];
  __shared__ ColAcc s_acc[kBlock / 64][kScanAggCols];
  rows_passing = wave_sum_u64(rows_passing);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c >= a.n_used) break;
    ColAcc r;
    r.cnt = wave_sum_u64(acc[c].cnt);
    r.sum_i = wave_sum_i64(acc[c].sum_i);
    r.min_i = wave_min_i64(acc[c].min_i);
    r.max_i = wave_max_i64(acc[c].max_i);
    r.sum_f = wave_sum_f64(acc[c].sum_f);
    r.min_f = wave_min_f64(acc[c].min_f);
    r.max_f = wave_max_f64(acc[c].max_f);
    if (lane == 0) s_acc[wave][c] = r;
  }
  if (lane == 0) s_rows[wave] = rows_passing;
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned long long rows = 0;
  for (int w = 0; w < kBlock / 64; ++w) rows += s_rows[w];
  if (!rows) return;  // nothing passed in this workgroup: the slots keep what they have
  // the workgroup's partial row, as the row function would have left it in a private buffer
  int64_t part[MI355Q_MAX_SLOTS];
  for (int j = 0; j < p.slot_count; ++j) part[j] = p.init_vals[j];
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    const int cs = a.target_cslot[i];
    if (cs < 0) {
      part[t.slot] = (int64_t)rows;
      continue;
    }
    ColAcc r = s_acc[0][cs];
    for (int w = 1; w < kBlock / 64; ++w) {
      const ColAcc& o = s_acc[w][cs];
      r.cnt += o.cnt;
      r.sum_i += o.sum_i;
      r.min_i = o.min_i < r.min_i ? o.min_i : r.min_i;
      r.max_i = r.max_i < o.max_i ? o.max_i : r.max_i;
      r.sum_f += o.sum_f;
      r.min_f = o.min_f < r.min_f ? o.min_f : r.min_f;
      r.max_f = r.max_f < o.max_f ? o.max_f : r.max_f;
    }
    const bool fp = a.type[cs] == MI355Q_DOUBLE;
    switch (t.agg) {
      case MI355Q_COUNT: part[t.slot] = (int64_t)r.cnt; break;
      case MI355Q_AVG:
        part[t.slot + 1] = (int64_t)r.cnt;
        [[fallthrough]];
      case MI355Q_SUM:
        if (r.cnt) part[t.slot] = fp ? dbl_bits(r.sum_f) : r.sum_i;
        break;
      case MI355Q_MIN:
        if (r.cnt) part[t.slot] = fp ? dbl_bits(r.min_f) : r.min_i;
        break;
      default:
        if (r.cnt) part[t.slot] = fp ? dbl_bits(r.max_f) : r.max_i;
    }
  }
  for (int i = 0; i < p.n_targets; ++i) reduce_target<true>(p.targets[i], p.init_vals, out, part);
}

// ---- typed members of the scan-aggregate family: every argument column a plain INT32, no quals, ONE set of aggregate
// kinds for all columns (the reference benchmark's NonGroupedAgg/NGA01-05.sql: six INT columns, one kind each) — the
// accumulators a kind does not need do not exist, MIN / MAX compare in 32 bits, the NULL test is one compare, and two
// quads per column are in flight.  OPS: 1 = rows with a value, 2 = sum, 4 = min, 8 = max.
template <int NC, int OPS, bool NUL, int UQ>
__global__ __launch_bounds__(kBlock) void k_scan_agg_i32(const int8_t* const* __restrict__ cols, const int64_t* __restrict__ num_rows,
                                                          int n_frags, int n_cols, ScanAggArgs a, DevPlan p, int64_t* __restrict__ out) {
  uint32_t cnt[NC];
  int64_t sum[NC];
  int32_t mn[NC], mx[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    cnt[c] = 0;
    sum[c] = 0;
    mn[c] = INT32_MAX;
    mx[c] = INT32_MIN;
  }
  unsigned long long rows = 0;
  auto one = [&](int c, int32_t x) {
    const bool ok = !NUL || x != INT32_MIN;
    if (OPS & 1) cnt[c] += ok ? 1u : 0u;
    if (OPS & 2) sum[c] += ok ? (int64_t)x : 0;
    if (OPS & 4) mn[c] = (ok && x < mn[c]) ? x : mn[c];
    if (OPS & 8) mx[c] = (ok && x > mx[c]) ? x : mx[c];   // (a NULL is INT32_MIN: it never raises a maximum anyway)
  };
  const int64_t tile_q = (int64_t)kBlock * UQ;
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * n_cols;
    const int64_t n = num_rows[f];
    const int64_t nq = n >> 2;
    const int64_t n_tiles = nq / tile_q;
    const int8_t* cbase[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) cbase[c] = c < a.n_used ? fc[a.col[c]] : nullptr;
    // software pipeline: the next tile's loads are issued before this tile's values are looked at (the members with a 64-bit
    // sum spend ~300 VALU instructions per tile: without it a wave had nothing in flight meanwhile — NGA02 / NGA05 0.56)
    v4i32 x[NC][UQ], y[NC][UQ];
    auto load_tile = [&](v4i32 (&d)[NC][UQ], int64_t t) {
      const int64_t q0 = t * tile_q + threadIdx.x;
#pragma unroll
      for (int u = 0; u < UQ; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          if (c < a.n_used) d[c][u] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)cbase[c] + q0 + (int64_t)u * kBlock);
    };
    int64_t t = (blockIdx.x + (int64_t)f * 7) % gridDim.x;
    if (t < n_tiles) load_tile(x, t);
    for (; t < n_tiles; t += gridDim.x) {
      if (t + gridDim.x < n_tiles) load_tile(y, t + gridDim.x);
#pragma unroll
      for (int u = 0; u < UQ; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          if (c >= a.n_used) break;
          one(c, x[c][u].x);
          one(c, x[c][u].y);
          one(c, x[c][u].z);
          one(c, x[c][u].w);
        }
      rows += 4 * UQ;
#pragma unroll
      for (int u = 0; u < UQ; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c) x[c][u] = y[c][u];
    }
    for (int64_t q = n_tiles * tile_q + gtid; q < nq; q += gsize) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c >= a.n_used) break;
        const v4i32 v = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)cbase[c] + q);
        one(c, v.x);
        one(c, v.y);
        one(c, v.z);
        one(c, v.w);
      }
      rows += 4;
    }
    const int64_t tail = (nq << 2) + gtid;
    if (tail < n) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (c >= a.n_used) break;
        one(c, load_one<int32_t>(cbase[c], tail));
      }
      rows += 1;
    }
  }
  // wave reduce, then one fold per workgroup in LDS (the epilogue of k_scan_agg on the accumulators this member keeps)
  __shared__ unsigned long long s_rows[kBlock / 64];
  __shared__ ColAcc s_acc[kBlock / 64][kScanAggCols];
  rows = wave_sum_u64(rows);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    if (c >= a.n_used) break;
    ColAcc r{};
    r.cnt = (OPS & 1) ? wave_sum_u64((unsigned long long)cnt[c]) : rows;   // (without NULLs every row has a value)
    r.sum_i = (OPS & 2) ? wave_sum_i64(sum[c]) : 0;
    r.min_i = (OPS & 4) ? wave_min_i64((long long)mn[c]) : INT64_MAX;
    r.max_i = (OPS & 8) ? wave_max_i64((long long)mx[c]) : INT64_MIN;
    if (lane == 0) s_acc[wave][c] = r;
  }
  if (lane == 0) s_rows[wave] = rows;
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned long long all_rows = 0;
  for (int w = 0; w < kBlock / 64; ++w) all_rows += s_rows[w];
  if (!all_rows) return;
  int64_t part[MI355Q_MAX_SLOTS];
  for (int j = 0; j < p.slot_count; ++j) part[j] = p.init_vals[j];
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    const int cs = a.target_cslot[i];
    if (cs < 0) {
      part[t.slot] = (int64_t)all_rows;
      continue;
    }
    ColAcc r = s_acc[0][cs];
    for (int w = 1; w < kBlock / 64; ++w) {
      const ColAcc& o = s_acc[w][cs];
      r.cnt += o.cnt;
      r.sum_i += o.sum_i;
      r.min_i = o.min_i < r.min_i ? o.min_i : r.min_i;
      r.max_i = r.max_i < o.max_i ? o.max_i : r.max_i;
    }
    switch (t.agg) {
      case MI355Q_COUNT: part[t.slot] = (int64_t)r.cnt; break;
      case MI355Q_AVG:
        part[t.slot + 1] = (int64_t)r.cnt;
        [[fallthrough]];
      case MI355Q_SUM:
        if (r.cnt) part[t.slot] = r.sum_i;
        break;
      case MI355Q_MIN:
        if (r.cnt) part[t.slot] = r.min_i;
        break;
      default:
        if (r.cnt) part[t.slot] = r.max_i;
    }
  }
  for (int i = 0; i < p.n_targets; ++i) reduce_target<true>(p.targets[i], p.init_vals, out, part);
}

// =========================================================================== perfect_lds
template <typename VT>
MQ_D void slot_apply_lds(int op, int64_t* s, int64_t key, VT val);

MQ_D void lds_minmax_f64(int64_t* s, double v, bool is_max) {
  int64_t old = *(volatile int64_t*)s;
  for (;;) {
    const double o = bits_dbl(old);
    const double r = is_max ? (o < v ? v : o) : (v < o ? v : o);
    const int64_t nv = dbl_bits(r);
    if (nv == old) return;
    const int64_t seen = (int64_t)atomicCAS((unsigned long long*)s, (unsigned long long)old,
                                            (unsigned long long)nv);
    if (seen == old) return;
    old = seen;
  }
}

template <typename VT>
MQ_D void apply_slot(int op, int64_t* s, int64_t key, VT val) {
  switch (op) {
    case SO_COUNT: atomicAdd((unsigned long long*)s, 1ull); break;
    case SO_KEY: *(volatile int64_t*)s = key; break;
    default:
      if constexpr (!is_none<VT>::value) {
        switch (op) {
          case SO_SUM_I: atomicAdd((unsigned long long*)s, (unsigned long long)(int64_t)val); break;
          case SO_SUM_F: atomicAdd((double*)s, (double)val); break;
          case SO_MIN_I: atomicMin((long long*)s, (long long)val); break;
          case SO_MAX_I: atomicMax((long long*)s, (long long)val); break;
          case SO_MIN_F: lds_minmax_f64(s, (double)val, false); break;
          case SO_MAX_F: lds_minmax_f64(s, (double)val, true); break;
        }
      }
  }
}
template <typename VT>
MQ_D void apply_slots(const SlotProg& sp, int64_t* slots, int64_t key, VT val) {
  for (int j = 0; j < sp.n; ++j) apply_slot<VT>(sp.op[j], slots + j, key, val);
}

// Merge one partial slot value into a global slot (NOT NULL semantics: init is the identity).
MQ_D void flush_slot(int op, int64_t* g, int64_t v, int64_t init) {
  if (v == init) return;
  switch (op) {
    case SO_COUNT:
    case SO_SUM_I: atomicAdd((unsigned long long*)g, (unsigned long long)v); break;
    case SO_SUM_F: atomicAdd((double*)g, bits_dbl(v)); break;
    case SO_MIN_I: atomicMin((long long*)g, (long long)v); break;
    case SO_MAX_I: atomicMax((long long*)g, (long long)v); break;
    case SO_MIN_F: a_minmax_f64<true, false, false>(g, bits_dbl(v), 0.0); break;
    case SO_MAX_F: a_minmax_f64<true, true, false>(g, bits_dbl(v), 0.0); break;
    default: MQ_STORE64(g, v);
  }
}

struct PerfectArgs {
  int64_t min_val, entry_count;
  int32_t row_quad, key_quad;
  int32_t kcol, vcol;
  SlotProg sp;
  int64_t init[MI355Q_MAX_SLOTS];
};

template <typename FT, typename KT, typename VT>
__global__ __launch_bounds__(kBlock) void k_perfect_lds(const int8_t* const* __restrict__ cols,
                                                         const int64_t* __restrict__ num_rows,
                                                         int n_frags, int n_cols, RangeFilter flt,
                                                         PerfectArgs a, int64_t* __restrict__ out,
                                                         int32_t* __restrict__ d_err) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int64_t* tab = (int64_t*)smem_raw;
  const int rq = a.row_quad, kq = a.key_quad;
  const int64_t ne = a.entry_count;
  const int64_t quads = ne * rq;
  // The LDS copy is COLUMN-major: quad j of entry e at tab[j * ne + e].  In the row-major image a
  // 16-byte row puts every SUM slot on an odd bank pair, i.e. the 64-bit ds_add of a wave only ever
  // uses half of the LDS banks; one contiguous array per quad spreads random entries over all of them.
  for (int64_t i = threadIdx.x; i < quads; i += kBlock) {
    const int j = (int)(i / ne);
    tab[i] = j < kq ? kEmptyKey64 : a.init[j - kq];
  }
  __syncthreads();
  bool bad = false;
  const SlotProg& sp = a.sp;
  scan_fragments<FT, KT, VT>(cols, num_rows, n_frags, n_cols, flt.col, a.kcol, a.vcol,
                             [&](FT fv, KT key, VT val) {
    if (!filter_pass<FT>(flt, fv)) return;
    const int64_t idx = (int64_t)key - a.min_val;
    if (idx < 0 || idx >= ne) {
      bad = true;
      return;
    }
    // every row of a group stores the same key: a plain store, nothing to read back or wait for
    // (the read-compare-store this replaces stalled the wave on an LDS round trip per row)
    if (kq) *(volatile int64_t*)(tab + idx) = (int64_t)key;
    for (int j = 0; j < sp.n; ++j) apply_slot<VT>(sp.op[j], tab + (int64_t)(kq + j) * ne + idx, (int64_t)key, val);
  });
  if (bad) atomicCAS(d_err, 0, MI355Q_ERR_OUT_OF_SLOTS);
  __syncthreads();
  // flush: one pass over the block's table (output rows stay row-major), only touched slots reach HBM
  for (int64_t i = threadIdx.x; i < quads; i += kBlock) {
    const int64_t e = i / rq;
    const int j = (int)(i % rq);
    const int64_t v = tab[(int64_t)j * ne + e];
    if (j < kq) {
      if (v != kEmptyKey64) MQ_STORE64(out + i, v);
    } else {
      flush_slot(a.sp.op[j - kq], out + i, v, a.init[j - kq]);
    }
  }
}

// The same kernel with the slot program known at compile time (PROG: one nibble per slot, op + 1) and
// 1024-lane workgroups.  The generic member above spends ~100 instructions per row walking the op switch
// of every slot (scalar branches, a waitcnt per LDS op): at 12 bytes per row that, not HBM, is what bounds
// it (cfg2: 0.62 of the roofline where the bare read pattern with one ds_add per row reaches 0.79 - 0.83,
// tools/microbench/groupby_small.hip).  Here a row is the index arithmetic in 32 bits, one key store and
// one LDS atomic per slot; one workgroup per CU keeps 16 waves streaming (measured: 1024 lanes per CU in one
// workgroup stream 4 - 5 % faster than in four, 2048 lanes per CU are slower).
constexpr int kPerfectWide = 1024;
template <uint32_t PROG, int J>
constexpr int prog_op() { return (int)((PROG >> (4 * J)) & 15u) - 1; }
template <uint32_t PROG>
constexpr int prog_n() { return PROG >= 0x1000u ? 4 : PROG >= 0x100u ? 3 : PROG >= 0x10u ? 2 : 1; }

template <typename FT, typename KT, typename VT, uint32_t PROG>
__global__ __launch_bounds__(kPerfectWide) void k_perfect_lds_prog(const int8_t* const* __restrict__ cols,
                                                                    const int64_t* __restrict__ num_rows,
                                                                    int n_frags, int n_cols, RangeFilter flt,
                                                                    PerfectArgs a, int64_t* __restrict__ out,
                                                                    int32_t* __restrict__ d_err) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int64_t* tab = (int64_t*)smem_raw;
  constexpr int NS = prog_n<PROG>();
  const int kq = a.key_quad, rq = a.row_quad;
  const uint32_t ne = (uint32_t)a.entry_count;
  const uint32_t quads = ne * (uint32_t)rq;
  // column-major LDS copy as in k_perfect_lds: quad j of entry e at tab[j * ne + e]
  for (uint32_t i = threadIdx.x; i < quads; i += kPerfectWide) {
    const int j = (int)(i / ne);
    tab[i] = j < kq ? kEmptyKey64 : a.init[j - kq];
  }
  __syncthreads();
  bool bad = false;
  int64_t* const slots = tab + (kq ? ne : 0u);
  scan_fragments<FT, KT, VT, 4>(cols, num_rows, n_frags, n_cols, flt.col, a.kcol, a.vcol,
                             [&](FT fv, KT key, VT val) {
    if (!filter_pass<FT>(flt, fv)) return;
    const uint64_t d = (uint64_t)((int64_t)key - a.min_val);
    if (d >= (uint64_t)ne) {
      bad = true;
      return;
    }
    const uint32_t idx = (uint32_t)d;
    // a keyed row stores the key once, in the key quad; a projected-key slot (SO_KEY, always slot 0 in
    // the compiled programs) is filled from it at the flush
    if (kq) *(volatile int64_t*)(tab + idx) = (int64_t)key;
    if (prog_op<PROG, 0>() != SO_KEY || !kq) apply_slot<VT>(prog_op<PROG, 0>(), slots + idx, (int64_t)key, val);
    if constexpr (NS > 1) apply_slot<VT>(prog_op<PROG, 1>(), slots + ne + idx, (int64_t)key, val);
    if constexpr (NS > 2) apply_slot<VT>(prog_op<PROG, 2>(), slots + 2 * ne + idx, (int64_t)key, val);
    if constexpr (NS > 3) apply_slot<VT>(prog_op<PROG, 3>(), slots + 3 * ne + idx, (int64_t)key, val);
  });
  if (bad) atomicCAS(d_err, 0, MI355Q_ERR_OUT_OF_SLOTS);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < quads; i += kPerfectWide) {
    const uint32_t e = i / (uint32_t)rq;
    const int j = (int)(i % (uint32_t)rq);
    const bool key_slot = kq && j == kq && prog_op<PROG, 0>() == SO_KEY;
    const int64_t v = tab[(key_slot ? 0u : (uint32_t)j * ne) + e];
    if (j < kq || key_slot) {
      if (v != kEmptyKey64) MQ_STORE64(out + i, v);
    } else {
      flush_slot(a.sp.op[j - kq], out + i, v, a.init[j - kq]);
    }
  }
}

// =========================================================================== baseline direct
struct BaselineArgs {
  int64_t entry_count;
  int32_t row_quad;
  int32_t kcol, vcol;
  SlotProg sp;  // slots after the key quad
};

template <typename FT, typename VT>
__global__ __launch_bounds__(kBlock) void k_baseline_direct(const int8_t* const* __restrict__ cols,
                                                             const int64_t* __restrict__ num_rows,
                                                             int n_frags, int n_cols,
                                                             RangeFilter flt, BaselineArgs a,
                                                             int64_t* __restrict__ out,
                                                             int32_t* __restrict__ d_err) {
  const uint32_t ne = (uint32_t)a.entry_count;
  bool full = false;
  scan_fragments<FT, int64_t, VT>(cols, num_rows, n_frags, n_cols, flt.col, a.kcol, a.vcol,
                                  [&](FT fv, int64_t key, VT val) {
    if (!filter_pass<FT>(flt, fv)) return;
    int64_t* slots = baseline_find_or_insert(out, ne, a.row_quad, 8, key);
    if (!slots) {
      full = true;
      return;
    }
    apply_slots_global(a.sp, slots, as_f64<VT>(val), as_i64<VT>(val), as_bits<VT>(val));
  });
  if (full) atomicCAS(d_err, 0, -1);
}

// =========================================================================== join + sum
struct JoinSumArgs {
  int32_t kcol, vcol;        // outer key column, outer value column (or -1)
  int32_t n_slots;
  int32_t op[4];             // 0 COUNT(*), 1 SUM(outer v), 2 SUM(inner w)
  int32_t hash_type;
  const void* table;
  int64_t min_key, max_key, entries;
  const int64_t* inner_w;    // inner int64 payload column (or null)
  const uint32_t* bitmap;    // perfect table's presence bitmap when no slot reads the inner table
  int64_t null_sum;          // NULL_BIGINT: the non-grouped SUM starts NULL
  int32_t k2col;             // second component of a composite (int64, int64) key, or -1
  uint32_t entries_rcp;      // floor(2^32 / entries): hash % entries without a division
};

// K2T = int64_t: composite (int64, int64) key on a keyed one-to-one table, slots (k0, k1, row id) — the hash
// is MurmurHash1 over the 16 key bytes, a slot is free when its first component is EMPTY_KEY_64
// (get_composite_key_index_impl, JoinHashTableQueryRuntime.cpp:140-163)
template <typename K2T, typename VT>
__global__ __launch_bounds__(kBlock) void k_join_sum(const int8_t* const* __restrict__ cols,
                                                      const int64_t* __restrict__ num_rows,
                                                      int n_frags, int n_cols, JoinSumArgs a,
                                                      int64_t* __restrict__ out) {
  long long acc[4] = {0, 0, 0, 0};
  unsigned long long contrib[4] = {0, 0, 0, 0};  // non-NULL contributions per slot
  auto slot_of = [&](uint32_t h) -> uint32_t {
    const uint32_t n = (uint32_t)a.entries;
    uint32_t r = h - __umulhi(h, a.entries_rcp) * n;
    if (r >= n) r -= n;
    return r;
  };
  scan_fragments<K2T, int64_t, VT>(cols, num_rows, n_frags, n_cols, a.k2col < 0 ? 0 : a.k2col, a.kcol, a.vcol,
                                   [&](K2T key2, int64_t key, VT val) {
    int64_t idx;
    if constexpr (!is_none<K2T>::value) {
      const int64_t* tab = (const int64_t*)a.table;
      const uint32_t n = (uint32_t)a.entries;
      const uint32_t w[4] = {(uint32_t)(uint64_t)key, (uint32_t)((uint64_t)key >> 32), (uint32_t)(uint64_t)key2,
                             (uint32_t)((uint64_t)key2 >> 32)};
      const uint32_t h = slot_of(murmur1_words(w, 4));
      idx = -1;
      uint32_t hp = h;
      do {
        const int64_t* e = tab + (size_t)hp * 3;
        const int64_t e0 = e[0], e1 = e[1];
        if (e0 == key && e1 == (int64_t)key2) {
          idx = e[2];
          break;
        }
        if (e0 == kEmptyKey64) break;
        hp = hp + 1 == n ? 0 : hp + 1;
      } while (hp != h);
    } else
    if (a.bitmap) {
      // semi-join: only WHETHER the key matches is needed, so probe the 1-bit-per-slot view of
      // the perfect table (32x smaller: 12.5 MB for 100 M dim rows, L2 / Infinity-Cache resident)
      const uint64_t off = (uint64_t)key - (uint64_t)a.min_key;
      idx = (key >= a.min_key && key <= a.max_key && ((a.bitmap[off >> 5] >> (off & 31)) & 1u)) ? 0 : -1;
    } else if (a.hash_type == 0) {
      idx = (key >= a.min_key && key <= a.max_key) ? ((const int32_t*)a.table)[key - a.min_key] : -1;
    } else {
      const int64_t* tab = (const int64_t*)a.table;
      const uint32_t n = (uint32_t)a.entries;
      idx = -1;
      const uint32_t h = slot_of(murmur1_u64((uint64_t)key));
      uint32_t hp = h;
      do {
        const int64_t k = tab[(size_t)hp * 2];
        if (k == key) {
          idx = tab[(size_t)hp * 2 + 1];
          break;
        }
        if (k == kEmptyKey64) break;
        hp = hp + 1 == n ? 0 : hp + 1;
      } while (hp != h);
    }
    if (idx < 0) return;
    for (int j = 0; j < a.n_slots; ++j) {
      if (a.op[j] == 0) {
        ++contrib[j];
      } else {
        // non-grouped SUM skips the NULL sentinel even on NOT NULL columns
        // (skip_null_val forced, TargetExprBuilder.cpp:684-690)
        const int64_t v = a.op[j] == 1 ? as_i64<VT>(val) : a.inner_w[idx];
        if (v != a.null_sum) {
          acc[j] += v;
          ++contrib[j];
        }
      }
    }
  });
  __shared__ long long s_acc[kBlock / 64][8];
  for (int j = 0; j < 4; ++j) {
    acc[j] = wave_sum_i64(acc[j]);
    contrib[j] = wave_sum_u64(contrib[j]);
  }
  if ((threadIdx.x & 63) == 0) {
    for (int j = 0; j < 4; ++j) {
      s_acc[threadIdx.x >> 6][j] = acc[j];
      s_acc[threadIdx.x >> 6][4 + j] = (long long)contrib[j];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int w = 0; w < kBlock / 64; ++w)
      for (int j = 0; j < 8; ++j) t[j] += s_acc[w][j];
    for (int j = 0; j < a.n_slots; ++j) {
      if (!t[4 + j]) continue;
      if (a.op[j] == 0) {
        atomicAdd((unsigned long long*)(out + j), (unsigned long long)t[4 + j]);
      } else {
        // slot is NULL until the first non-NULL contribution; a partial sum that happens to
        // equal the sentinel bit pattern must still be added, so CAS explicitly
        int64_t old = MQ_LOAD64(out + j);
        for (;;) {
          const int64_t nv = old == a.null_sum ? t[j] : old + t[j];
          const int64_t seen = (int64_t)MQ_CAS64(out + j, old, nv);
          if (seen == old) break;
          old = seen;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------- host side
inline int stream_grid(int n_cus, int blocks_per_cu, int64_t total_rows) {
  const int dbg = tune_knobs().blocks_per_cu;
  if (dbg > 0) blocks_per_cu = dbg;
  int64_t want = (total_rows / 4 + kBlock - 1) / kBlock;
  int64_t cap = (int64_t)n_cus * blocks_per_cu;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

constexpr int64_t kPerfectLdsMaxBytes = 64 * 1024;

}  // namespace

// ------------------------------------------------------------------------ scan_count
bool scan_count_eligible(const DevPlan& p, const FragView& fv) {
  if (p.desc_type != MI355Q_NON_GROUPED_AGGREGATE || p.join_col >= 0) return false;
  if (p.n_targets != 1 || p.targets[0].agg != MI355Q_COUNT || p.targets[0].col >= 0) return false;
  if (p.n_quals != 1) return false;
  RangeFilter f;
  if (!make_range_filter(p.quals[0], &f)) return false;
  return all_aligned16(fv, p.quals[0].col);
}

hipError_t launch_scan_count(const DevPlan& p, const FragView& fv, int64_t* out, int n_cus,
                             hipStream_t s, LaunchStats* st) {
  RangeFilter f;
  make_range_filter(p.quals[0], &f);
  const int grid = stream_grid(n_cus, 2, fv.total_rows);
  st->kernel_name = "k_scan_count";
  st->n_launches = 1;
  rec(st->k_start, s);
  if (p.quals[0].type == MI355Q_INT32) {
    hipLaunchKernelGGL(k_scan_count<int32_t>, dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, f, out);
  } else {
    hipLaunchKernelGGL(k_scan_count<int64_t>, dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, f, out);
  }
  rec(st->k_stop, s);
  return hipGetLastError();
}


// ------------------------------------------------------------------------ scan_agg
static bool scan_agg_args(const DevPlan& p, const FragView& fv, ScanAggArgs* a) {
  if (p.desc_type != MI355Q_NON_GROUPED_AGGREGATE || p.join_col >= 0 || p.n_quals > MI355Q_MAX_QUALS) return false;
  std::memset(a, 0, sizeof(*a));
  for (int i = 0; i < p.n_quals; ++i) {
    if (!make_range_filter(p.quals[i], &a->flt[i], true)) return false;  // (1-byte filter columns too)
    a->flt_type[i] = p.quals[i].type;
    if (!all_aligned16(fv, p.quals[i].col)) return false;
  }
  a->n_flt = merge_range_filters(a->flt, a->flt_type, p.n_quals);
  if (p.bf_active) {  // the compiled filter's columns take the filter slots
    const BoolFilter* bf = step_bool_filter();
    if (!bf || p.n_quals != 0 || bf->n_cols > 4 || bf->n_progs != 0) return false;  // (program atoms: the row-mask pre-pass)
    for (int k = 0; k < bf->n_cols; ++k) {
      if (!all_aligned16(fv, bf->col[k])) return false;
      a->flt[k] = no_filter();
      a->flt[k].col = bf->col[k];
      a->flt_type[k] = bf->col_type[k];
    }
    a->n_flt = bf->n_cols;
    a->bf_on = 1;
    a->bf = step_bool_filter_dev();
  }
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    a->target_cslot[i] = -1;
    if (t.table != 0 || t.arg_f32) return false;
    if (t.agg == MI355Q_COUNT && t.col < 0) continue;
    if (t.agg != MI355Q_COUNT && t.agg != MI355Q_SUM && t.agg != MI355Q_MIN && t.agg != MI355Q_MAX && t.agg != MI355Q_AVG)
      return false;
    if (t.col < 0) return false;
    if (t.arg_type != MI355Q_INT32 && t.arg_type != MI355Q_INT64 && t.arg_type != MI355Q_DOUBLE) return false;
    // SUM / AVG of an integer column accumulate 64-bit; an AVG over integers keeps an integer sum slot
    int cs = -1;
    for (int c = 0; c < a->n_used; ++c)
      if (a->col[c] == t.col) cs = c;
    if (cs < 0) {
      if (a->n_used >= kScanAggCols) return false;
      cs = a->n_used++;
      a->col[cs] = t.col;
      a->type[cs] = t.arg_type;
      a->nullable[cs] = t.skip_null;
      if (!all_aligned16(fv, t.col)) return false;
    } else if (a->nullable[cs] != t.skip_null) {
      return false;  // one column, two NULL conventions: the row kernel
    }
    a->target_cslot[i] = cs;
  }
  return true;
}

bool scan_agg_eligible(const DevPlan& p, const FragView& fv) {
  ScanAggArgs a;
  return scan_agg_args(p, fv, &a);
}

hipError_t launch_scan_agg(const DevPlan& p, const FragView& fv, int64_t* out, int n_cus, hipStream_t s, LaunchStats* st) {
  ScanAggArgs a;
  if (!scan_agg_args(p, fv, &a)) return hipErrorInvalidValue;
  const int grid = stream_grid(n_cus, 2, fv.total_rows);
  st->kernel_name = "k_scan_agg";
  st->n_launches = 1;
  rec(st->k_start, s);
  // the typed members: plain INT32 argument columns, no quals, one set of aggregate kinds
  {
    bool typed = a.n_flt == 0 && a.n_used >= 1 && !(tune_knobs().flags & MI355Q_OPT_LDS_GENERIC_MEMBER);
    bool nul = false;
    for (int c = 0; c < a.n_used; ++c) {
      typed = typed && a.type[c] == MI355Q_INT32;
      nul = nul || a.nullable[c] != 0;
    }
    int ops = 0;
    for (int i = 0; i < p.n_targets; ++i) {
      if (a.target_cslot[i] < 0) continue;
      switch (p.targets[i].agg) {
        case MI355Q_COUNT: ops |= 1; break;
        case MI355Q_SUM: ops |= 2; break;
        case MI355Q_AVG: ops |= 2; break;
        case MI355Q_MIN: ops |= 4; break;
        default: ops |= 8;
      }
    }
    // (with NULLs every kind needs to know whether a column had a value at all; without them that is the row count)
    if (nul) ops |= 1;
    // the instantiated kind sets: count, sum, min, max (each with the value count where the columns are nullable), or all
    const int set = ops == 1 ? 1 : (ops & ~1) == 2 ? 2 : (ops & ~1) == 4 ? 4 : (ops & ~1) == 8 ? 8 : 15;
    if (typed && ops != 0) {
      st->variant = 5;
      // workgroups per CU, measured at 1 B rows x six columns with the software pipeline (profiles/
      // r06_nga_1b_pipelined_sweep.jsonl): the members without a 64-bit sum are best at three (3.51 - 3.58 ms = 0.84 - 0.86 of
      // 8 TB/s; four: 3.72 - 3.77), the ones with one at two (3.94 ms = 0.76; three: 4.20)
      const int grid = stream_grid(n_cus, (set == 2 || set == 15) ? 2 : 3, fv.total_rows);
#define MQ_SA(NC, OPS, NUL, UQ) \
  hipLaunchKernelGGL((k_scan_agg_i32<NC, OPS, NUL, UQ>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows, fv.n_frags, fv.n_cols, a, p, out)
#define MQ_SA_OPS(NC, UQ)                                   \
  do {                                                      \
    if (nul) {                                              \
      if (set == 1) MQ_SA(NC, 1, true, UQ);                 \
      else if (set == 2) MQ_SA(NC, 3, true, UQ);            \
      else if (set == 4) MQ_SA(NC, 5, true, UQ);            \
      else if (set == 8) MQ_SA(NC, 9, true, UQ);            \
      else MQ_SA(NC, 15, true, UQ);                         \
    } else {                                                \
      if (set == 1) MQ_SA(NC, 1, false, UQ);                \
      else if (set == 2) MQ_SA(NC, 2, false, UQ);           \
      else if (set == 4) MQ_SA(NC, 4, false, UQ);           \
      else if (set == 8) MQ_SA(NC, 8, false, UQ);           \
      else MQ_SA(NC, 15, false, UQ);                        \
    }                                                       \
  } while (0)
      if (a.n_used <= 2) MQ_SA_OPS(2, 4);
      else if (a.n_used <= 4) MQ_SA_OPS(4, 2);
      else if (a.n_used <= 6) MQ_SA_OPS(6, 2);
      else MQ_SA_OPS(8, 1);
#undef MQ_SA_OPS
#undef MQ_SA
      rec(st->k_stop, s);
      return hipGetLastError();
    }
  }
  // every lane keeps about eight 16-byte loads in flight: quads per column per step by the number of columns
  const int streams = a.n_used + a.n_flt;
  if (streams <= 2)
    hipLaunchKernelGGL((k_scan_agg<4, 2, 2>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows, fv.n_frags, fv.n_cols, a, p, out);
  else if (streams <= 4)
    hipLaunchKernelGGL((k_scan_agg<2, 4, 4>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows, fv.n_frags, fv.n_cols, a, p, out);
  else if (a.n_flt <= 4)
    hipLaunchKernelGGL((k_scan_agg<1, kScanAggCols, 4>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows, fv.n_frags, fv.n_cols, a, p, out);
  else
    hipLaunchKernelGGL((k_scan_agg<1, kScanAggCols, MI355Q_MAX_QUALS>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows, fv.n_frags, fv.n_cols, a, p, out);
  rec(st->k_stop, s);
  return hipGetLastError();
}

// ------------------------------------------------------------------------ perfect_lds
static int perfect_key_storage(const DevPlan& p) {
  if (p.group_type == MI355Q_INT32 || p.group_type == MI355Q_INT64) return p.group_type;
  if (tc_enc(p.group_type) == MI355Q_ENC_FIXED && tc_storage(p.group_type) == MI355Q_INT32 && !p.group_nullable)
    return MI355Q_INT32;
  return 0;
}

bool perfect_lds_eligible(const DevPlan& p, const FragView& fv) {
  if (p.desc_type != MI355Q_GROUP_BY_PERFECT_HASH) return false;
  // plain INT / BIGINT keys, or a NOT NULL kENCODING_FIXED(32) key (the same 4-byte load)
  if (perfect_key_storage(p) == 0) return false;
  // a bucketed range indexes by (key - min) / bucket (get_group_value_fast); this kernel's index
  // is key - min, so bucketed keys stay with the row kernel
  if (p.group_bucket[0] != 0) return false;
  if (p.entry_count * p.row_quad * 8 > kPerfectLdsMaxBytes) return false;
  FastShape s;
  return grouped_fast_shape(p, fv, &s) && !s.sp.val_nullable;
}

// The compile-time slot programs (nibble = SlotOp + 1, slot 0 in the lowest nibble) and the value type
// each is instantiated for; anything else runs the generic member.
constexpr uint32_t prog_of(int o0, int o1 = -1, int o2 = -1, int o3 = -1) {
  return (uint32_t)(o0 + 1) | (uint32_t)(o1 + 1) << 4 | (uint32_t)(o2 + 1) << 8 | (uint32_t)(o3 + 1) << 12;
}
template <typename FT, typename KT>
static bool launch_perfect_prog_v(const FastShape& fs, const PerfectArgs& a, const FragView& fv, int64_t* out,
                                  int32_t* d_err, int grid, size_t lds, hipStream_t s) {
  if (a.key_quad > 1 || fs.sp.n < 1 || fs.sp.n > 4 || a.row_quad != a.key_quad + fs.sp.n) return false;
  const uint32_t prog = prog_of(fs.sp.op[0], fs.sp.n > 1 ? fs.sp.op[1] : -1, fs.sp.n > 2 ? fs.sp.op[2] : -1,
                                fs.sp.n > 3 ? fs.sp.op[3] : -1);
  const int vt = fs.vcol < 0 ? 0 : fs.vtype;
#define MQ_PROG(VT_CODE, VT, ...)                                                                              \
  if (vt == (VT_CODE) && prog == prog_of(__VA_ARGS__)) {                                                       \
    auto k = k_perfect_lds_prog<FT, KT, VT, prog_of(__VA_ARGS__)>;                                             \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kPerfectWide), lds, s, fv.d_cols, fv.d_num_rows, fv.n_frags, fv.n_cols, \
                       fs.flt, a, out, d_err);                                                                 \
    return true;                                                                                               \
  }
  // (a keyed table projects the key into slot 0 as well: SO_KEY)
  MQ_PROG(0, none_t, SO_COUNT)
  MQ_PROG(0, none_t, SO_KEY, SO_COUNT)
  MQ_PROG(MI355Q_INT64, int64_t, SO_SUM_I)
  MQ_PROG(MI355Q_INT64, int64_t, SO_KEY, SO_SUM_I)
  MQ_PROG(MI355Q_INT64, int64_t, SO_COUNT, SO_SUM_I)
  MQ_PROG(MI355Q_INT64, int64_t, SO_SUM_I, SO_COUNT)
  MQ_PROG(MI355Q_INT64, int64_t, SO_KEY, SO_COUNT, SO_SUM_I)
  MQ_PROG(MI355Q_INT64, int64_t, SO_KEY, SO_SUM_I, SO_COUNT)
  MQ_PROG(MI355Q_INT64, int64_t, SO_KEY, SO_COUNT, SO_SUM_I, SO_COUNT)
  MQ_PROG(MI355Q_INT32, int32_t, SO_SUM_I)
  MQ_PROG(MI355Q_INT32, int32_t, SO_KEY, SO_SUM_I)
  MQ_PROG(MI355Q_INT32, int32_t, SO_KEY, SO_SUM_I, SO_COUNT)
  MQ_PROG(MI355Q_DOUBLE, double, SO_SUM_F)
  MQ_PROG(MI355Q_DOUBLE, double, SO_KEY, SO_SUM_F)
  MQ_PROG(MI355Q_DOUBLE, double, SO_COUNT, SO_SUM_F)
  MQ_PROG(MI355Q_DOUBLE, double, SO_SUM_F, SO_COUNT)
  MQ_PROG(MI355Q_DOUBLE, double, SO_KEY, SO_COUNT, SO_SUM_F)
  MQ_PROG(MI355Q_DOUBLE, double, SO_KEY, SO_SUM_F, SO_COUNT)
  MQ_PROG(MI355Q_DOUBLE, double, SO_KEY, SO_COUNT, SO_SUM_F, SO_COUNT)
#undef MQ_PROG
  return false;
}

template <typename FT, typename KT>
static hipError_t launch_perfect_lds_v(const FastShape& fs, const PerfectArgs& a, const FragView& fv,
                                       int64_t* out, int32_t* d_err, int grid, size_t lds,
                                       hipStream_t s, int n_cus) {
  {  // compile-time slot program, one 1024-lane workgroup per CU
    int64_t want = (fv.total_rows / 4 + kPerfectWide - 1) / kPerfectWide;
    if (want < 1) want = 1;
    const int grid_w = (int)(want < n_cus ? want : n_cus);
    if (launch_perfect_prog_v<FT, KT>(fs, a, fv, out, d_err, grid_w, lds, s)) return hipGetLastError();
  }
  if (fs.vcol < 0) {
    hipLaunchKernelGGL((k_perfect_lds<FT, KT, none_t>), dim3(grid), dim3(kBlock), lds, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  } else if (fs.vtype == MI355Q_INT64) {
    hipLaunchKernelGGL((k_perfect_lds<FT, KT, int64_t>), dim3(grid), dim3(kBlock), lds, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  } else if (fs.vtype == MI355Q_INT32) {
    hipLaunchKernelGGL((k_perfect_lds<FT, KT, int32_t>), dim3(grid), dim3(kBlock), lds, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  } else {
    hipLaunchKernelGGL((k_perfect_lds<FT, KT, double>), dim3(grid), dim3(kBlock), lds, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  }
  return hipGetLastError();
}

hipError_t launch_perfect_lds(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err,
                              int n_cus, hipStream_t s, LaunchStats* st) {
  FastShape fs;
  grouped_fast_shape(p, fv, &fs);
  PerfectArgs a{};
  a.min_val = p.min_val;
  a.entry_count = p.entry_count;
  a.row_quad = p.row_quad;
  a.key_quad = p.key_quad;
  a.kcol = p.group_col;
  a.vcol = fs.vcol < 0 ? 0 : fs.vcol;
  a.sp = fs.sp;
  for (int i = 0; i < MI355Q_MAX_SLOTS; ++i) a.init[i] = p.init_vals[i];
  const size_t lds = (size_t)(p.entry_count * p.row_quad * 8);
  int bpc = (int)((160 * 1024) / (lds + 512));
  if (bpc > 4) bpc = 4;
  if (bpc < 1) bpc = 1;
  const int grid = stream_grid(n_cus, bpc, fv.total_rows);
  st->kernel_name = "k_perfect_lds";
  st->n_launches = 1;
  rec(st->k_start, s);
  hipError_t e;
  const bool k32 = perfect_key_storage(p) == MI355Q_INT32;
  if (fs.fil_type == 0) {
    e = k32 ? launch_perfect_lds_v<none_t, int32_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus)
            : launch_perfect_lds_v<none_t, int64_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus);
  } else if (fs.fil_type == MI355Q_INT32) {
    e = k32 ? launch_perfect_lds_v<int32_t, int32_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus)
            : launch_perfect_lds_v<int32_t, int64_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus);
  } else if (fs.fil_type == MI355Q_INT8) {
    e = k32 ? launch_perfect_lds_v<int8_t, int32_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus)
            : launch_perfect_lds_v<int8_t, int64_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus);
  } else {
    e = k32 ? launch_perfect_lds_v<int64_t, int32_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus)
            : launch_perfect_lds_v<int64_t, int64_t>(fs, a, fv, out, d_err, grid, lds, s, n_cus);
  }
  rec(st->k_stop, s);
  return e;
}

// ------------------------------------------------------------------------ baseline
bool baseline_fast_eligible(const DevPlan& p, const FragView& fv) {
  if (p.desc_type != MI355Q_GROUP_BY_BASELINE_HASH || p.key_width != 8) return false;
  // BIGINT keys, or DOUBLE keys as their bit pattern (groupByColumnCodegen bit-casts a floating-point key to i64,
  // IRCodegen.cpp:1505-1507: -0.0 and +0.0 are different groups, NULL_DOUBLE is an ordinary key) — the shape of the
  // reference's BaselineHash/BH001-006 benchmark queries, GROUP BY cast(x AS DOUBLE)
  if (p.group_type != MI355Q_INT64 && p.group_type != MI355Q_DOUBLE) return false;
  FastShape s;
  return grouped_fast_shape(p, fv, &s);
}

// Plan-time choice inside the baseline family: tiny inputs or tiny tables are fastest with
// direct atomics; everything else partitions.
int baseline_fast_variant(const DevPlan& p, const FragView& fv, int requested, int n_cus) {
  if (requested == 1) return 1;
  const bool can_part = part_supported(p, fv, n_cus);
  if (requested == 2) return can_part ? 2 : 1;
  // direct atomics for small inputs and for small TABLES: measured on 1 B rows with 10 K groups (refbench PHS004 /
  // PHM003 / BH004), the partitioned family takes 2.2 s — ten keys per partition, each a tenth of its rows, all
  // through the heavy-hitter path — against 0.58 s of contended direct atomics (profiles/r03_refbench_*.jsonl); from
  // 100 K groups on it wins (18 ms).  Tables of at most a few thousand groups never get here (the LDS group-by
  // takes them); the band in between is the open cliff of DESIGN section 9.
  if (!can_part || fv.total_rows < (int64_t)8 << 20 || p.entry_count < 65536) return 1;
  return 2;
}

int64_t baseline_fast_scratch_bytes(const DevPlan& p, const FragView& fv, int variant,
                                    int64_t cap_bytes, int n_cus) {
  const int v = baseline_fast_variant(p, fv, variant, n_cus);
  if (v == 1) return 0;
  return part_scratch_bytes(p, fv, n_cus, cap_bytes);
}

template <typename FT>
static hipError_t launch_baseline_direct_v(const FastShape& fs, const BaselineArgs& a,
                                           const FragView& fv, int64_t* out, int32_t* d_err,
                                           int grid, hipStream_t s) {
  if (fs.vcol < 0) {
    hipLaunchKernelGGL((k_baseline_direct<FT, none_t>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  } else if (fs.vtype == MI355Q_INT64) {
    hipLaunchKernelGGL((k_baseline_direct<FT, int64_t>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  } else if (fs.vtype == MI355Q_INT32) {
    hipLaunchKernelGGL((k_baseline_direct<FT, int32_t>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  } else {
    hipLaunchKernelGGL((k_baseline_direct<FT, double>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, fs.flt, a, out, d_err);
  }
  return hipGetLastError();
}

hipError_t launch_baseline_fast(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err,
                                void* scratch, int64_t scratch_bytes, int64_t cap_bytes,
                                int variant, int n_cus, hipStream_t s, LaunchStats* st) {
  const int v = baseline_fast_variant(p, fv, variant, n_cus);
  if (v != 1) {
    return launch_baseline_partitioned(p, fv, out, d_err, scratch, scratch_bytes, cap_bytes, n_cus, s,
                                       st);
  }
  FastShape fs;
  grouped_fast_shape(p, fv, &fs);
  BaselineArgs a{};
  a.entry_count = p.entry_count;
  a.row_quad = p.row_quad;
  a.kcol = p.group_col;
  a.vcol = fs.vcol < 0 ? 0 : fs.vcol;
  a.sp = fs.sp;
  const int grid = stream_grid(n_cus, 8, fv.total_rows);
  st->kernel_name = "k_baseline_direct";
  st->n_launches = 1;
  st->variant = 1;
  rec(st->k_start, s);
  hipError_t e;
  if (fs.fil_type == 0) e = launch_baseline_direct_v<none_t>(fs, a, fv, out, d_err, grid, s);
  else if (fs.fil_type == MI355Q_INT32) e = launch_baseline_direct_v<int32_t>(fs, a, fv, out, d_err, grid, s);
  else if (fs.fil_type == MI355Q_INT8) e = launch_baseline_direct_v<int8_t>(fs, a, fv, out, d_err, grid, s);
  else e = launch_baseline_direct_v<int64_t>(fs, a, fv, out, d_err, grid, s);
  rec(st->k_stop, s);
  return e;
}

// ------------------------------------------------------------------------ join_sum
static bool join_sum_shape(const DevPlan& p, const FragView& fv, JoinSumArgs* a) {
  if (p.desc_type != MI355Q_NON_GROUPED_AGGREGATE || p.join_col < 0 || p.n_quals != 0) return false;
  if (p.join_type != MI355Q_INT64 || p.join_nullable) return false;
  // one-to-one tables with one 8-byte key component — or, keyed, two of them — INNER joins
  if (p.join_hash_type > 1 || p.join_width != 8 || p.join_kind != MI355Q_JOIN_INNER) return false;
  a->k2col = -1;
  if (p.join_n_keys == 2) {
    if (p.join_hash_type != 1 || p.join_types[1] != MI355Q_INT64 || p.join_nullables[1]) return false;
    a->k2col = p.join_cols[1];
    if (!all_aligned16(fv, a->k2col)) return false;
  } else if (p.join_n_keys != 1) {
    return false;
  }
  if (p.n_targets > 4) return false;
  a->kcol = p.join_col;
  a->vcol = -1;
  a->inner_w = nullptr;
  a->n_slots = p.n_targets;
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    if (t.slot != i) return false;
    if (t.agg == MI355Q_COUNT && t.col < 0) {
      a->op[i] = 0;
    } else if (t.agg == MI355Q_SUM && t.arg_type == MI355Q_INT64 && !t.arg_nullable) {
      if (t.table == 0) {
        if (a->vcol >= 0 && a->vcol != t.col) return false;
        a->vcol = t.col;
        a->op[i] = 1;
      } else {
        const int64_t* w = (const int64_t*)p.inner_cols[t.col];
        if (a->inner_w && a->inner_w != w) return false;
        a->inner_w = w;
        a->op[i] = 2;
      }
    } else {
      return false;
    }
  }
  if (!all_aligned16(fv, p.join_col)) return false;
  if (a->vcol >= 0 && !all_aligned16(fv, a->vcol)) return false;
  a->hash_type = p.join_hash_type;
  a->bitmap = (p.join_hash_type == 0 && !a->inner_w) ? p.join_bitmap : nullptr;
  a->table = p.join_buf;
  a->min_key = p.join_min;
  a->max_key = p.join_max;
  a->entries = p.join_entries;
  a->entries_rcp = 0;
  if (p.join_hash_type == 1) {
    if (p.join_entries < 2 || p.join_entries >= ((int64_t)1 << 32)) return false;
    a->entries_rcp = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)p.join_entries);
  }
  a->null_sum = INT64_MIN;
  return true;
}

bool join_sum_eligible(const DevPlan& p, const FragView& fv) {
  JoinSumArgs a;
  return join_sum_shape(p, fv, &a);
}

hipError_t launch_join_sum(const DevPlan& p, const FragView& fv, int64_t* out, int n_cus,
                           hipStream_t s, LaunchStats* st) {
  JoinSumArgs a;
  join_sum_shape(p, fv, &a);
  const int grid = stream_grid(n_cus, 4, fv.total_rows);
  st->kernel_name = "k_join_sum";
  st->n_launches = 1;
  rec(st->k_start, s);
  const bool no_val = a.vcol < 0;
  if (no_val) a.vcol = 0;
  if (a.k2col >= 0) {
    if (no_val)
      hipLaunchKernelGGL((k_join_sum<int64_t, none_t>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows,
                         fv.n_frags, fv.n_cols, a, out);
    else
      hipLaunchKernelGGL((k_join_sum<int64_t, int64_t>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows,
                         fv.n_frags, fv.n_cols, a, out);
  } else if (no_val) {
    hipLaunchKernelGGL((k_join_sum<none_t, none_t>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, a, out);
  } else {
    hipLaunchKernelGGL((k_join_sum<none_t, int64_t>), dim3(grid), dim3(kBlock), 0, s, fv.d_cols,
                       fv.d_num_rows, fv.n_frags, fv.n_cols, a, out);
  }
  rec(st->k_stop, s);
  return hipGetLastError();
}

}  // namespace mq
