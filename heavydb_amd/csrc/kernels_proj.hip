// kernels_proj.hip — the PROJECTION family: `SELECT cols / expressions FROM t WHERE quals [LIMIT n]`, one output entry
// per row that passes the filter (QueryDescriptionType::Projection).
//
// What the reference does per matching row (heavyai/heavydb): old = total_matched++ — an atomic on the GPU, a plain
// counter in the CPU kernel of one fragment (GroupByAndAggregate.cpp:1080-1101) — then get_scan_output_slot /
// get_columnar_scan_output_offset (GroupByRuntime.cpp:242-269) writes the row's offset in its fragment into entry
// `old` and the targets are stored behind it with agg_id (TargetExprBuilder.cpp:330-560); a row that finds the buffer
// full makes the row function answer -pos (GroupByAndAggregate.cpp:1151-1156), and with a scan limit the loop stops
// at max_matched (QueryTemplateGenerator.cpp:751-780).
//
// On gfx950 this is a stream compaction that keeps the CPU executor's order (fragment, then row) — a result
// identical to the reference's CPU path, not merely the same set:
//   * a workgroup takes TILES of 16 384 rows off one ticket counter (one returning atomic per 16 K rows);
//   * pass A streams the filter columns of the tile with 16-byte loads (4 rows per lane and load), evaluates the
//     quals and keeps the tile's match bits in one 64-bit register per lane;
//   * the tile's match count is published in a 64-bit {state, value} descriptor and the tile's first output entry
//     is the sum of its predecessors' counts, read back by one wave ("decoupled look-back": a tile adds the
//     aggregates of the tiles before it until it meets one whose inclusive prefix is known) — relaxed agent-scope
//     8-byte atomics on both sides, the hand-off form that needs no fence (MI355X_MICROARCH.md, "8-B agent atomics
//     both sides");
//   * pass B walks the tile in sub-tiles of 1 024 / 2 048 rows: ranks from a block scan of the match bits, the
//     projected columns are loaded ONLY where a quad has a matching row, the entries are assembled in LDS in their
//     final layout (row-wise rows, or one run per column) and leave as contiguous, fully coalesced stores.
// Expressions (projected or in a qual) are evaluated in registers by the micro-op evaluator (expr.h): no
// temporary column, no second pass — the reference compiles them into the row function the same way
// (Executor::compileBody, NativeCodegen.cpp:3455: filters first, then the body).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "expr.h"
#include "regprog.h"
#include "fast_common.h"
#include "kernels.h"
#include "rowfunc.h"

namespace mq {

namespace {

using fast::v2i64;
using fast::v4i32;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kIters = 16;                          // quads per lane and tile
constexpr int kIterRows = kBlock * 4;               // rows one iteration of the block covers
constexpr int64_t kTileRows = (int64_t)kIterRows * kIters;  // 16 384
constexpr int kCntBits = 15;                        // a tile's matches (<= 16 384) in the low bits of its count word, its fragment above
constexpr uint32_t kSparseTile = 1024;              // fast member: tiles with at most this many matches (6 %) emit entry by entry
constexpr int64_t kSplitMinTiles = 4096;            // the fast member's split route from 67 M rows (below: one fused launch)

constexpr uint64_t kStateShift = 62;
constexpr uint64_t kStateAggregate = 1ull << kStateShift;  // value = the tile's own count
constexpr uint64_t kStateInclusive = 2ull << kStateShift;  // value = count of the tile and everything before it
constexpr uint64_t kValueMask = (1ull << kStateShift) - 1;

struct ProjArgs {
  ProjSpec ps;
  const DevExprSet* xs;          // device memory; null when the plan has no expressions
  uint32_t qual_expr_mask;       // expressions a qual reads (evaluated for every row)
  int32_t n_frags;
  const int8_t* const* cols;     // [frag][ps.n_cols_table]
  const int64_t* num_rows;       // [frag]
  const int64_t* tile_start;     // [frag + 1] first tile of each fragment
  int64_t n_tiles;
  unsigned long long* desc;      // [n_tiles], zeroed
  unsigned long long* counters;  // [0] ticket, [1] total_matched
  int8_t* out;
  int32_t* d_err;
  int32_t sub_iters;             // iterations per sub-tile (1 or 2): what the LDS image holds
  int32_t vec_mask;              // bit c: column c of every fragment is 16-byte aligned (vector loads allowed)
  int32_t row_quals;             // the quals hold a disjunction (or_group): evaluated by quals_pass, row by row
  // quals over plain INT32 / INT64 columns, normalised at plan time to lo <= v <= hi (+ negation, + NULL exclusion):
  // qmode 1 = INT32 (compared in 32 bits: lo / hi clamped to the type), 2 = INT64, 0 = the general comparison
  int32_t qmode[MI355Q_MAX_QUALS];
  fast::RangeFilter qf[MI355Q_MAX_QUALS];
  int32_t out16;                 // row-wise buffer on a 16-byte boundary (paired stores for odd target counts)
  int32_t fast_quals;            // every qual compares an aligned physical column with a literal, no disjunction
  int32_t fast_targets;          // every target is an aligned physical column
  // expressions: the typed-handler evaluator's area in the workgroup's dynamic LDS, behind the output image — the
  // programs as XNodes, x_below stack levels of 4 x kBlock values, then the values of the expressions (n x 4 x kBlock);
  // x_lds_off < 0: no room (very deep programs beside a wide image) — the row-at-a-time evaluator with its private stack
  int32_t x_lds_off, x_below;
};

// ---- four rows of one column ------------------------------------------------------------------
struct RawQuad {
  v2i64 a, b;
};
// storage width w; `quad` = index of the 4-row group in the chunk
MQ_D void load_raw_quad(const int8_t* base, int w, int64_t quad, RawQuad& r) {
  if (w == 8) {
    r.a = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + quad * 2);
    r.b = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + quad * 2 + 1);
  } else if (w == 4) {
    r.a = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + quad);
  } else if (w == 2) {
    r.a.x = __builtin_nontemporal_load((const MQ_GLOBAL long long*)base + quad);
  } else {
    r.a.x = (long long)(unsigned)__builtin_nontemporal_load((const MQ_GLOBAL int*)base + quad);
  }
}
// element i as the sign-extended storage integer (FLOAT: its 32 bits, DOUBLE: its 64 bits)
MQ_D int64_t raw_elem(const RawQuad& r, int w, int i) {
  if (w == 8) return i == 0 ? r.a.x : i == 1 ? r.a.y : i == 2 ? r.b.x : r.b.y;
  if (w == 4) {
    const uint64_t q = (uint64_t)(i < 2 ? r.a.x : r.a.y);
    return (int64_t)(int32_t)(uint32_t)(q >> ((i & 1) * 32));
  }
  if (w == 2) return (int64_t)(int16_t)(uint16_t)((uint64_t)r.a.x >> (i * 16));
  return (int64_t)(int8_t)(uint8_t)((uint64_t)r.a.x >> (i * 8));
}
// the value col_value_bits would return for that element
MQ_D int64_t value_of_raw(int code, int64_t raw) {
  const int st = tc_storage(code);
  if (st == MI355Q_FLOAT) return (int64_t)(uint32_t)raw;
  if (st == MI355Q_DOUBLE) return raw;
  return decode_loaded(code, raw);
}

// all expressions of `mask`, in order, into xv (an expression may read the earlier ones)
MQ_D void eval_exprs(const DevExprSet& xs, uint32_t mask, const int8_t* const* fc, int64_t pos, int64_t* xv, int32_t* err) {
  for (int k = 0; k < xs.n; ++k)
    if ((mask >> k) & 1u) xv[k] = eval_expr(xs.e[k], fc, pos, err, xv, xs.n_cols);
}

// ---- the expressions in LDS form (expr.h eval_expr_rows: typed handlers, the stack below its top and the expressions'
// values in LDS — nothing in scratch memory)
constexpr int kXJ = 4;  // rows evaluated together: the four rows of a lane's quad
struct ProjExprLds {
  const XNode* prog;  // MI355Q_MAX_EXPR_NODES per expression
  ExLdsStack stk;
  int64_t* xv;        // value of expression k, row j of the quad: xv[(k * kXJ + j) * kBlock + tid]
};
// the expressions of `mask`, in order, for the rows pos[0..3] (out-of-fragment rows clamped by the caller); err[j] = the first
// error row j met (its value is then unspecified)
MQ_D void eval_exprs_lds(const DevExprSet& xs, const ProjExprLds& L, uint32_t mask, const int8_t* const* fc, const int64_t (&pos)[kXJ],
                         int32_t (&err)[kXJ]) {
  const int64_t raw[kExPre][kXJ] = {};
#pragma unroll 1
  for (int k = 0; k < xs.n; ++k) {
    if (!((mask >> k) & 1u)) continue;
    int64_t v[kXJ];
    int32_t e[kXJ];
    if ((xs.noerr_mask >> k) & 1) eval_expr_rows<kXJ, kBlock, false>(xs.e[k], L.prog + k * MI355Q_MAX_EXPR_NODES, fc, pos, raw, L.stk, v, e, L.xv, xs.n_cols);
    else eval_expr_rows<kXJ, kBlock, true>(xs.e[k], L.prog + k * MI355Q_MAX_EXPR_NODES, fc, pos, raw, L.stk, v, e, L.xv, xs.n_cols);
#pragma unroll
    for (int j = 0; j < kXJ; ++j) {
      L.xv[(size_t)(k * kXJ + j) * kBlock + L.stk.tid] = v[j];
      if (e[j] && !err[j]) err[j] = e[j];
    }
  }
}

MQ_D unsigned long long wave_excl_scan_u32x2(unsigned long long v, unsigned long long* total) {
  // inclusive scan over the wave of two packed 32-bit counters
  const int lane = threadIdx.x & 63;
  unsigned long long inc = v;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl(inc, lane - d < 0 ? lane : lane - d, 64);
    if (lane >= d) inc += o;
  }
  *total = __shfl(inc, 63, 64);
  return inc - v;
}


// the inner row an outer row joins through a ONE-TO-ONE table (process_row's probe, rowfunc.h): >= 0, or -1 = no match
// (a NULL key matches nothing: hash_join_idx_nullable)
MQ_D int64_t proj_join_row(const DevPlan& p, const int8_t* const* fc, int64_t pos) {
  int64_t jk[MI355Q_MAX_GROUP_COLS];
  bool null_key = false;
  for (int i = 0; i < p.join_n_keys; ++i) {
    jk[i] = decode_int(fc[p.join_cols[i]], p.join_types[i], pos);
    null_key = null_key || (p.join_nullables[i] && jk[i] == int_null_of(p.join_types[i]));
  }
  if (null_key) return -1;
  const JoinMatch jm = join_lookup(p, jk);
  return jm.count > 0 ? jm.single : -1;
}
// the value an unmatched row of a LEFT join shows for an inner column: the type's NULL (codegenOuterJoinNullPlaceholder)
MQ_D int64_t proj_null_bits(int code) {
  const int st = tc_storage(code);
  if (st == MI355Q_DOUBLE) return kNullDoubleBits;
  if (st == MI355Q_FLOAT) return (int64_t)(uint32_t)kNullFloatBits;
  return int_null_of(code);
}

// the whole filter of one row — every kind of qual (any column type and encoding, disjunctions, quals on expressions).  ONE
// call site per kernel (the general path of pass A), so that the row function is instantiated once
// HJ: the step joins — an INNER join keeps the rows that find a match
// L (HX): the LDS form of the evaluator, or L.xv == nullptr.  xslot >= 0: the quals' expressions have been evaluated for
// this row's quad (tile_filter): the row's values are slot xslot of L.xv
template <bool HX, bool HJ = false>
MQ_D bool row_passes(const DevPlan& p, const ProjArgs& a, const ProjExprLds& L, const int8_t* const* fc, int64_t pos, int32_t* err, int xslot) {
  const int n_phys = a.ps.n_phys_cols;
  int64_t xv[HX ? MI355Q_MAX_EXPRS : 1];
  const bool lds_x = HX && xslot >= 0;
  if (HX && a.qual_expr_mask && !lds_x) eval_exprs(*a.xs, a.qual_expr_mask, fc, pos, xv, err);
  uint32_t seen = 0, any = 0;
#pragma unroll 1
  for (int k = 0; k < p.n_quals; ++k) {
    const DevQual& q = p.quals[k];
    const bool t = (HX && q.col >= n_phys)
                       ? qual_on_value(q, lds_x ? L.xv[(size_t)((q.col - n_phys) * kXJ + xslot) * kBlock + L.stk.tid] : xv[HX ? q.col - n_phys : 0])
                       : eval_qual(q, fc[q.col < n_phys ? q.col : 0], pos);
    if (q.or_group == 0) {
      if (!t) return false;
    } else {
      seen |= 1u << q.or_group;
      if (t) any |= 1u << q.or_group;
    }
  }
  if (seen != any) return false;
  if (HJ && p.join_kind != MI355Q_JOIN_LEFT) return proj_join_row(p, fc, pos) >= 0;
  return true;
}

// ---- pass A: the filter over one tile.  Bit (4 u + i) of the result = row i of the lane's quad in iteration u.
// Four iterations at a time.  Where all four lie inside the fragment and the plan's quals are comparisons of aligned
// physical columns with literals (a.fast_quals), every qual is one vector load per quad and lane — INT32 / INT64 range
// forms compare in registers, the other types go through qual_on_value; everything else (a fragment's ragged end,
// unaligned chunks, disjunctions, quals on expressions) takes row_passes, row by row.
template <bool HX, bool HJ = false>
MQ_D uint64_t tile_filter(const DevPlan& p, const ProjArgs& a, const ProjExprLds& L, const int8_t* const* fc, int64_t n, int64_t row0, int32_t* err) {
  const int tid = threadIdx.x;
  uint64_t m = 0;
#pragma unroll 1
  for (int u0 = 0; u0 < kIters; u0 += 4) {
    uint32_t bits[4];
    int64_t r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = row0 + (int64_t)(u0 + j) * kIterRows + tid * 4;
      const int64_t left = n - r[j];
      bits[j] = left >= 4 ? 0xfu : left <= 0 ? 0u : ((1u << left) - 1u);
    }
    const bool inside = row0 + (int64_t)(u0 + 4) * kIterRows <= n;  // (uniform)
    if (inside && a.fast_quals) {
#pragma unroll 1
      for (int k = 0; k < p.n_quals; ++k) {
        const DevQual& q = p.quals[k];
        const int8_t* base = fc[q.col];
        const int mode = a.qmode[k];
        if (mode == 1) {  // plain INT32 column, range form, compared in 32 bits
          const fast::RangeFilter& f = a.qf[k];
          const int32_t lo = (int32_t)f.lo, hi = (int32_t)f.hi, nul = (int32_t)f.null_val;
          v4i32 x[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) x[j] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + (r[j] >> 2));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint32_t pass = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int32_t v = x[j][i];
              bool in = v >= lo && v <= hi;
              if (f.negate) in = !in;
              if (f.nullable && v == nul) in = false;
              pass |= (uint32_t)in << i;
            }
            bits[j] &= pass;
          }
        } else if (mode == 2) {  // plain INT64 column, range form
          const fast::RangeFilter& f = a.qf[k];
          v2i64 xa[4], xb[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            xa[j] = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + (r[j] >> 2) * 2);
            xb[j] = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + (r[j] >> 2) * 2 + 1);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint32_t pass = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int64_t v = i == 0 ? xa[j].x : i == 1 ? xa[j].y : i == 2 ? xb[j].x : xb[j].y;
              pass |= (uint32_t)fast::filter_pass<int64_t>(f, v) << i;
            }
            bits[j] &= pass;
          }
        } else {  // any other column type / encoding: one quad at a time through the general comparison
          const int w = type_width(q.type);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            RawQuad raw;
            load_raw_quad(base, w, r[j] >> 2, raw);
            uint32_t pass = 0;
#pragma unroll 1
            for (int i = 0; i < 4; ++i) pass |= (uint32_t)qual_on_value(q, value_of_raw(q.type, raw_elem(raw, w, i))) << i;
            bits[j] &= pass;
          }
        }
      }
    } else {
#pragma unroll 1
      for (int ji = 0; ji < 16; ++ji) {
        const int j = ji >> 2, i = ji & 3;
        const uint32_t bj = j == 0 ? bits[0] : j == 1 ? bits[1] : j == 2 ? bits[2] : bits[3];
        const int64_t rj = j == 0 ? r[0] : j == 1 ? r[1] : j == 2 ? r[2] : r[3];
        const bool quad_x = HX && a.qual_expr_mask && L.xv != nullptr;
        if (quad_x && i == 0 && bj) {  // the quals' expressions for the quad's four rows together (rows past the end repeat the last)
          int64_t p4[kXJ];
          int32_t e4[kXJ] = {0, 0, 0, 0};
#pragma unroll
          for (int x = 0; x < kXJ; ++x) p4[x] = rj + x < n ? rj + x : n - 1;
          eval_exprs_lds(*a.xs, L, a.qual_expr_mask, fc, p4, e4);
#pragma unroll
          for (int x = 0; x < kXJ; ++x)
            if (((bj >> x) & 1u) && e4[x] && !*err) *err = e4[x];
        }
        if (!((bj >> i) & 1u)) continue;
        if (!row_passes<HX, HJ>(p, a, L, fc, rj + i, err, quad_x ? i : -1)) {
          const uint32_t clr = ~(1u << i);
          if (j == 0) bits[0] &= clr;
          if (j == 1) bits[1] &= clr;
          if (j == 2) bits[2] &= clr;
          if (j == 3) bits[3] &= clr;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= (uint64_t)bits[j] << (4 * (u0 + j));
  }
  return m;
}

// the tile's predecessors: publishes the tile's count, returns the number of matches before the tile (wave 0 only)
MQ_D unsigned long long tile_lookback(unsigned long long* desc, unsigned long long* counters, int64_t n_tiles, int64_t tile,
                                      unsigned long long tile_count) {
  const int lane = threadIdx.x & 63;
  if (lane == 0)
    __hip_atomic_store(&desc[tile], (tile == 0 ? kStateInclusive : kStateAggregate) | tile_count, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long excl = 0;
  int64_t idx = tile - 1;  // nearest predecessor this round looks at
  while (idx >= 0) {
    const int64_t mine = idx - lane;
    const bool valid = mine >= 0;
    unsigned long long d = kStateInclusive;  // (before tile 0: an inclusive prefix of 0)
    if (valid) d = __hip_atomic_load(&desc[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__any(valid && (d >> kStateShift) == 0)) {  // a predecessor still in pass A: look again
      __builtin_amdgcn_s_sleep(8);
      if (valid && (d >> kStateShift) == 0) d = __hip_atomic_load(&desc[mine], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long incl_lanes = __ballot((d >> kStateShift) == 2);
    const int stop = incl_lanes ? __builtin_ctzll(incl_lanes) : 63;  // nearest tile whose inclusive prefix is known
    unsigned long long v = lane <= stop ? (d & kValueMask) : 0ull;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    excl += __shfl(v, 0, 64);
    if (incl_lanes) break;
    idx -= 64;
  }
  if (lane == 0) {
    if (tile > 0)
      __hip_atomic_store(&desc[tile], kStateInclusive | (excl + tile_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile == n_tiles - 1) counters[1] = excl + tile_count;
  }
  return excl;
}

// ---- the FAST row-wise member: quals are range forms over plain INT32 / INT64 columns, targets are plain INT64 /
// DOUBLE / INT32 / FLOAT columns, every chunk 16-byte aligned (anything else: the general member below).  No LDS image:
// the rank of every matching row of the tile is known after ONE exchange — per iteration a lane's rank inside its wave
// is a sum of mbcnt over the four bit planes of the wave's match masks (ballots), the (iteration, wave) totals are
// scanned by one wave — and every lane then stores its rows whole: 8 (1 + NT) contiguous bytes per row, adjacent ranks
// adjacent in memory, so the stores of a tile fill whole lines in the L2.  The projected columns of one iteration are
// loaded together (NT x 16 / 32 bytes in flight per lane), only for quads with a match.
struct FastArgs {
  int32_t n_quals, n_targets, n_frags, n_cols_table;
  int32_t qcol[MI355Q_MAX_QUALS], qmode[MI355Q_MAX_QUALS];  // mode 1: INT32 in 32 bits, 2: INT64, 3: INT8 (four rows = one 4-byte load)
  fast::RangeFilter qf[MI355Q_MAX_QUALS];
  int32_t tcol[MI355Q_MAX_TARGETS], tkind[MI355Q_MAX_TARGETS];  // kind 0: 8 raw bytes, 1: INT32 sign-extended, 2: FLOAT -> double
  int64_t entry_count;
  const int8_t* const* cols;
  const int64_t* num_rows;
  const int64_t* tile_start;
  int64_t n_tiles;
  unsigned long long* desc;
  unsigned long long* counters;
  int64_t* out;
  int32_t out16, columnar;
  int32_t tw[MI355Q_MAX_TARGETS];        // columnar: bytes of the target's slot (8 or 4)
  int64_t tcol_off[MI355Q_MAX_TARGETS];  // columnar: byte offset of the slot column
  // targets that are forms (kernels.h ProjForm): tcol / tkind describe the column the form READS (kind 0 or 1)
  int32_t any_form, pad2_;
  int32_t f_on[MI355Q_MAX_TARGETS], f_src_type[MI355Q_MAX_TARGETS], f_cast_to[MI355Q_MAX_TARGETS], f_cast_flags[MI355Q_MAX_TARGETS];
  int32_t f_op[MI355Q_MAX_TARGETS], f_type[MI355Q_MAX_TARGETS], f_flags[MI355Q_MAX_TARGETS], f_lit_first[MI355Q_MAX_TARGETS];
  int64_t f_lit[MI355Q_MAX_TARGETS];
  int32_t* d_err;
  // the SPLIT route (large inputs): pass A ran in k_proj_mask — pre_mask[tile * kBlock + tid] is the lane's 64 match bits,
  // desc[tile] the number of matches BEFORE the tile (k_proj_scan_tiles); no look-back
  unsigned long long* pre_mask;
  uint32_t* pre_cnt;  // [tile] matches of the tile (low 15 bits; <= kSparseTile -> k_proj_sparse emits it, else k_proj_fast) | fragment << 15
  uint32_t* dense;    // [counters[2]] the tiles with more matches than that, in order: k_proj_fast's tickets walk this list
};
// one row of a form target: the column's value (ex_col: integers sign-extended, DOUBLE as its bits) -> the target's value
MQ_D int64_t form_eval(int cast_from, int cast_to, int cast_flags, int op, int type, int flags, bool lit_first, int64_t lit, int64_t v,
                       int32_t& ev) {
  if (cast_to) {
    DevExprNode n{};
    n.op = MI355Q_EX_CAST;
    n.arg = cast_from;
    n.type = cast_to;
    n.flags = cast_flags;
    v = ex_cast(n, v, ev);
  }
  if (op) {
    DevExprNode n{};
    n.op = op;
    n.type = type;
    n.flags = flags;
    v = lit_first ? ex_arith(n, lit, v, ev) : ex_arith(n, v, lit, ev);
  }
  return ex_wrap_int(op ? type : cast_to, v);
}

template <typename T>
MQ_D bool range_pass(const fast::RangeFilter& f, T v);
template <>
MQ_D bool range_pass<int32_t>(const fast::RangeFilter& f, int32_t v) {
  bool in = v >= (int32_t)f.lo && v <= (int32_t)f.hi;
  if (f.negate) in = !in;
  if (f.nullable && v == (int32_t)f.null_val) in = false;
  return in;
}
template <>
MQ_D bool range_pass<int64_t>(const fast::RangeFilter& f, int64_t v) { return fast::filter_pass<int64_t>(f, v); }
template <>
MQ_D bool range_pass<int8_t>(const fast::RangeFilter& f, int8_t v) { return fast::filter_pass<int8_t>(f, v); }  // (a 1-byte column: the row mask)

// one qual over four iterations of the tile: bits[j] &= pass mask of the lane's quad of iteration u0 + j
template <typename T>
MQ_D void fast_qual(const fast::RangeFilter& f, const int8_t* base, const int64_t (&r)[4], bool inside, int64_t n, uint32_t (&bits)[4]) {
  if (inside) {
    fast::Quad<T> x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fast::load_quad<T>(base, r[j] >> 2, x[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t pass = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) pass |= (uint32_t)range_pass<T>(f, x[j].v[i]) << i;
      bits[j] &= pass;
    }
  } else {  // the fragment's ragged end: row by row, only inside the fragment
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t pass = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((bits[j] >> i) & 1u) pass |= (uint32_t)range_pass<T>(f, fast::load_one<T>(base, r[j] + i)) << i;
      bits[j] = pass;
    }
  }
}

// pass A of the fast member for one tile: bit 4 u + i = row i of the lane's quad in iteration u passes every qual
MQ_D uint64_t fast_pass_a(const FastArgs& a, const int8_t* const* fc, int64_t n, int64_t row0, int tid) {
  uint64_t m = 0;
#pragma unroll 1
  for (int u0 = 0; u0 < kIters; u0 += 4) {
    uint32_t bits[4];
    int64_t r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[j] = row0 + (int64_t)(u0 + j) * kIterRows + tid * 4;
      const int64_t left = n - r[j];
      bits[j] = left >= 4 ? 0xfu : left <= 0 ? 0u : ((1u << left) - 1u);
    }
    const bool inside = row0 + (int64_t)(u0 + 4) * kIterRows <= n;  // (uniform)
#pragma unroll 1
    for (int k = 0; k < a.n_quals; ++k) {
      if (a.qmode[k] == 1) fast_qual<int32_t>(a.qf[k], fc[a.qcol[k]], r, inside, n, bits);
      else if (a.qmode[k] == 3) fast_qual<int8_t>(a.qf[k], fc[a.qcol[k]], r, inside, n, bits);
      else fast_qual<int64_t>(a.qf[k], fc[a.qcol[k]], r, inside, n, bits);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= (uint64_t)bits[j] << (4 * (u0 + j));
  }
  return m;
}
// the fragment a tile belongs to
MQ_D int fast_tile_frag(const FastArgs& a, int64_t tile) {
  int lo = 0, hi = a.n_frags;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.tile_start[mid] <= tile) lo = mid;
    else hi = mid;
  }
  return lo;
}

// ---- the SPLIT route of the fast member (round 6).  The fused kernel's tile is a chain — ticket, pass A, the look-back,
// pass B — and at four workgroups per CU the chain, not the memory system, sets the pace when few rows match: 1 B rows with a
// 4-byte filter column took 2.0 ms at 0.1 % selectivity where the column streams in 0.65 ms.  Here pass A is a kernel of its own
// with nothing to wait for (k_proj_mask: the lane's 64 match bits = 1 bit per row, and the tile's count), one workgroup scans
// the counts (k_proj_scan_tiles), and k_proj_fast runs as pass B alone.
__global__ __launch_bounds__(kBlock) void k_proj_mask(FastArgs a) {
  __shared__ uint32_t s_w[kWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int f = fast_tile_frag(a, tile);
    const int8_t* const* fc = a.cols + (size_t)f * a.n_cols_table;
    const int64_t n = a.num_rows[f];
    const int64_t row0 = (tile - a.tile_start[f]) * kTileRows;
    const uint64_t m = fast_pass_a(a, fc, n, row0, tid);
    __builtin_nontemporal_store((unsigned long long)m, a.pre_mask + tile * kBlock + tid);
    uint32_t c = (uint32_t)__popcll(m);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (lane == 0) s_w[wave] = c;
    __syncthreads();
    if (tid == 0) {
      uint32_t t = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) t += s_w[w];
      a.pre_cnt[tile] = t | ((uint32_t)f << kCntBits);
    }
    __syncthreads();
  }
}
// desc[tile] = matches before the tile; counters[1] = matches of all tiles; dense[0 .. counters[2]) = the tiles with more than
// kSparseTile matches, in order.  ONE workgroup of 1 024 lanes, a run of tiles each.
__global__ __launch_bounds__(1024) void k_proj_scan_tiles(const uint32_t* cnt, unsigned long long* desc, int64_t n_tiles,
                                                          unsigned long long* counters, uint32_t* dense) {
  __shared__ unsigned long long s_sum[1024];
  __shared__ uint32_t s_dense[1024];
  constexpr uint32_t kCntMask = (1u << kCntBits) - 1u;
  const int tid = threadIdx.x;
  const int64_t per = (((n_tiles + 1023) / 1024) + 3) & ~(int64_t)3;  // (runs start on 16-byte boundaries of the counts)
  const int64_t lo = (int64_t)tid * per < n_tiles ? (int64_t)tid * per : n_tiles, hi = lo + per < n_tiles ? lo + per : n_tiles;
  unsigned long long sum = 0;
  uint32_t nd = 0;
  {  // (four 16-byte loads in flight per lane: a run is read in a few latencies, not one per tile)
    int64_t i = lo;
    for (; i + 16 <= hi; i += 16) {
      v4i32 c[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) c[x] = *(const v4i32*)(cnt + i + 4 * x);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const uint32_t w[4] = {(uint32_t)c[x].x & kCntMask, (uint32_t)c[x].y & kCntMask, (uint32_t)c[x].z & kCntMask, (uint32_t)c[x].w & kCntMask};
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          sum += w[y];
          nd += w[y] > kSparseTile;
        }
      }
    }
    for (; i < hi; ++i) {
      const uint32_t w = cnt[i] & kCntMask;
      sum += w;
      nd += w > kSparseTile;
    }
  }
  s_sum[tid] = sum;
  s_dense[tid] = nd;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned long long o = tid >= d ? s_sum[tid - d] : 0ull;
    const uint32_t od = tid >= d ? s_dense[tid - d] : 0u;
    __syncthreads();
    s_sum[tid] += o;
    s_dense[tid] += od;
    __syncthreads();
  }
  unsigned long long run = s_sum[tid] - sum;
  uint32_t di = s_dense[tid] - nd;
  {
    int64_t i = lo;
    for (; i + 8 <= hi; i += 8) {
      v4i32 c[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) c[x] = *(const v4i32*)(cnt + i + 4 * x);
      const uint32_t cc[8] = {(uint32_t)c[0].x, (uint32_t)c[0].y, (uint32_t)c[0].z, (uint32_t)c[0].w,
                              (uint32_t)c[1].x, (uint32_t)c[1].y, (uint32_t)c[1].z, (uint32_t)c[1].w};
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const uint32_t w = cc[x] & kCntMask;
        desc[i + x] = run;
        run += w;
        if (w > kSparseTile) dense[di++] = (uint32_t)(i + x);
      }
    }
    for (; i < hi; ++i) {
      const uint32_t w = cnt[i] & kCntMask;
      desc[i] = run;
      run += w;
      if (w > kSparseTile) dense[di++] = (uint32_t)i;
    }
  }
  if (tid == 1023) {
    counters[1] = s_sum[1023];
    counters[2] = s_dense[1023];
  }
}

// pass B of the SPARSE tiles of the split route (at most kSparseTile matches): the tile's matches are listed in LDS in rank
// order (their row inside the tile, 14 bits each), then entry j is the work of lane j — it loads its row's columns one value each
// and writes the entry: adjacent lanes write adjacent entries, and every load of the tile is in flight at once.  (The quad path
// of k_proj_fast walks the 16 iterations one after the other, each a load -> stage -> store round trip: at 1 % selectivity nearly
// every (iteration, wave) holds a match or two and a tile cost sixteen exposed latencies — 0.75 ms per 1 B rows at 0.1 %.)  One
// member for every shape: the target count, the layout and the forms are run-time values here.
__global__ __launch_bounds__(kBlock) void k_proj_sparse(FastArgs a) {
  __shared__ uint16_t s_list[kSparseTile];
  __shared__ uint32_t s_cnt[kIters * kWaves];
  constexpr uint32_t kCntMask = (1u << kCntBits) - 1u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = a.n_targets, rq = 1 + nt;
  int32_t err = 0;
  // the NEXT tile's count word, base and match bits are loaded while this tile is emitted (a tile is a chain of a few dependent
  // loads otherwise)
  int64_t tile = blockIdx.x;
  uint32_t w_next = 0;
  unsigned long long base_next = 0, m_next = 0;
  if (tile < a.n_tiles) {
    w_next = a.pre_cnt[tile];
    base_next = a.desc[tile];
    m_next = __builtin_nontemporal_load(a.pre_mask + tile * kBlock + tid);
  }
  for (; tile < a.n_tiles; tile += gridDim.x) {
    const uint32_t w = w_next;
    const int64_t tile_base = (int64_t)base_next;
    const uint64_t m = m_next;
    if (tile + gridDim.x < a.n_tiles) {
      w_next = a.pre_cnt[tile + gridDim.x];
      base_next = a.desc[tile + gridDim.x];
      m_next = __builtin_nontemporal_load(a.pre_mask + (tile + gridDim.x) * kBlock + tid);
    }
    const uint32_t tile_matches = w & kCntMask;
    if (tile_matches == 0 || tile_matches > kSparseTile) continue;  // (uniform)
    if (tile_base >= a.entry_count) continue;
    const int f = (int)(w >> kCntBits);
    const int8_t* const* fc = a.cols + (size_t)f * a.n_cols_table;
    const int64_t row0 = (tile - a.tile_start[f]) * kTileRows;
    // matches of every (iteration, wave), their exclusive prefix in that order (as k_proj_fast)
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < kIters; ++u) {
      uint32_t c = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) c += (uint32_t)__popcll(__ballot((m >> (4 * u + b)) & 1ull));
      if (lane == u) mine = c;
    }
    if (lane < kIters) s_cnt[lane * kWaves + wave] = mine;
    __syncthreads();
    if (wave == 0) {
      const uint32_t v = s_cnt[lane];
      uint32_t inc = v;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl(inc, lane - d < 0 ? lane : lane - d, 64);
        if (lane >= d) inc += o;
      }
      s_cnt[lane] = inc - v;
    }
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < kIters; ++u) {
      const uint32_t mm = (uint32_t)(m >> (4 * u)) & 0xfu;
      if (!__any(mm != 0)) continue;  // (uniform)
      uint32_t k = s_cnt[u * kWaves + wave];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned long long bal = __ballot((mm >> b) & 1u);
        k += __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((mm >> i) & 1u) s_list[k++] = (uint16_t)(u * kIterRows + tid * 4 + i);
    }
    __syncthreads();
    // entries j and j + kBlock of a lane together (their loads, then their stores), four targets at a time
    int64_t n_emit = (int64_t)tile_matches;
    if (tile_base + n_emit > a.entry_count) n_emit = a.entry_count - tile_base;  // the scan limit / the buffer's end
#pragma unroll 1
    for (int64_t j0 = tid; j0 < n_emit; j0 += 2 * kBlock) {
      const bool two = j0 + kBlock < n_emit;
      const int64_t rank[2] = {tile_base + j0, tile_base + j0 + kBlock};
      const int64_t rr[2] = {row0 + (int64_t)s_list[j0], row0 + (int64_t)s_list[two ? j0 + kBlock : j0]};
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        if (y == 1 && !two) break;
        if (!a.columnar) a.out[rank[y] * rq] = rr[y];
        else a.out[rank[y]] = rr[y];
      }
#pragma unroll 1
      for (int t0 = 0; t0 < nt; t0 += 4) {
        int64_t v[2][4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int t = t0 + x < nt ? t0 + x : nt - 1;
          const int8_t* base = fc[a.tcol[t]];
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            if (a.tkind[t] == 0) v[y][x] = __builtin_nontemporal_load((const MQ_GLOBAL long long*)base + rr[y]);
            else v[y][x] = (int64_t)__builtin_nontemporal_load((const MQ_GLOBAL int*)base + rr[y]);
          }
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int t = t0 + x;
          if (t >= nt) break;
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            if (y == 1 && !two) break;
            int64_t val = v[y][x];
            if (a.tkind[t] == 2) val = dbl_bits((double)bits_flt((int32_t)val));
            if (a.any_form && a.f_on[t]) {
              int32_t e = 0;
              val = form_eval(a.f_src_type[t], a.f_cast_to[t], a.f_cast_flags[t], a.f_op[t], a.f_type[t], a.f_flags[t], a.f_lit_first[t] != 0,
                              a.f_lit[t], val, e);
              if (e && !err) err = e;
            }
            if (!a.columnar) a.out[rank[y] * rq + 1 + t] = val;
            else {
              char* col = (char*)a.out + a.tcol_off[t];
              if (a.tw[t] == 8) ((int64_t*)col)[rank[y]] = val;
              else ((int32_t*)col)[rank[y]] = (int32_t)val;
            }
          }
        }
      }
    }
    __syncthreads();  // (s_list / s_cnt are the next tile's)
  }
  if (err) atomicCAS(a.d_err, 0, err);
}

// NT targets; COL: a columnar buffer (one run per column) instead of whole rows.  A wave stages the quads of LH lanes at a
// time (ranks are lane-major: a run of lanes is a run of entries): all 64 while 256 staged entries fit the LDS budget of
// four workgroups per CU, else 32.
// XF: some target is a form (kernels.h ProjForm) — members without forms carry none of that code
// (an empty statement the optimiser cannot move: it keeps SimplifyCFG from merging the LAST stores of two branches into one store
// through a phi of two offsets of `vals` — an address no longer constant, which leaves the whole array in scratch)
#define MQ_NO_STORE_SINK() asm volatile("")
template <int NT, bool COL, bool XF = false>
__global__ __launch_bounds__(kBlock, (XF || COL) ? 1 : NT <= 3 ? 4 : 3) void k_proj_fast(FastArgs a, int) {
  extern __shared__ __attribute__((aligned(16))) char s_stage[];  // per wave: the entries of one step, assembled before they leave
  __shared__ uint32_t s_cnt[kIters * kWaves];   // matches of (iteration, wave); then their exclusive prefix in that order
  __shared__ long long s_bcast[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int rq = 1 + NT;
  constexpr int LH = NT > 3 ? 32 : 64;
  constexpr int kStageRows = LH * 4;
  char* const stage = s_stage + (size_t)wave * kStageRows * rq * 8;
  int32_t err = 0;  // the error a form target raised for an emitted row
  // columnar: the stage holds the key run, then one run per target of its slot width
  int32_t run_off[NT + 1];
  run_off[0] = kStageRows * 8;
#pragma unroll
  for (int t = 0; t < NT; ++t) run_off[t + 1] = run_off[t] + kStageRows * (COL ? a.tw[t] : 8);

  // (the split route: the tickets walk the list of dense tiles — IN ORDER, as the look-back keeps the fused launch: workgroups
  // that take tiles by position drift apart and the entries they write no longer fill the buffer front to back; measured 20 – 30 %
  // slower on wide rows)
  const int64_t n_tickets = a.pre_mask ? (int64_t)a.counters[2] : a.n_tiles;
  for (;;) {
    if (tid == 0) s_bcast[0] = (long long)atomicAdd(&a.counters[0], 1ull);
    __syncthreads();
    const int64_t ticket = s_bcast[0];
    if (ticket >= n_tickets) break;
    const int64_t tile = a.pre_mask ? (int64_t)a.dense[ticket] : ticket;
    const int f = fast_tile_frag(a, tile);
    const int8_t* const* fc = a.cols + (size_t)f * a.n_cols_table;
    const int64_t n = a.num_rows[f];
    const int64_t row0 = (tile - a.tile_start[f]) * kTileRows;

    // ---- pass A (or its result, on the split route)
    const uint64_t m = a.pre_mask ? (uint64_t)__builtin_nontemporal_load(a.pre_mask + tile * kBlock + tid) : fast_pass_a(a, fc, n, row0, tid);

    // ---- matches of every (iteration, wave): lane u of the wave keeps iteration u's
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < kIters; ++u) {
      uint32_t c = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) c += (uint32_t)__popcll(__ballot((m >> (4 * u + b)) & 1ull));
      if (lane == u) mine = c;
    }
    if (lane < kIters) s_cnt[lane * kWaves + wave] = mine;
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the 64 counts in (iteration, wave) order; the tile's total; its predecessors
      const uint32_t v = s_cnt[lane];
      uint32_t inc = v;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl(inc, lane - d < 0 ? lane : lane - d, 64);
        if (lane >= d) inc += o;
      }
      const unsigned long long tile_count = (unsigned long long)__shfl(inc, 63, 64);
      s_cnt[lane] = inc - v;
      const unsigned long long excl = a.pre_mask ? a.desc[tile] : tile_lookback(a.desc, a.counters, a.n_tiles, tile, tile_count);
      if (lane == 0) s_bcast[1] = (long long)excl;
    }
    __syncthreads();
    const int64_t tile_base = s_bcast[1];
    if (tile_base >= a.entry_count) continue;  // past a scan limit (or a full buffer): nothing of this tile is kept

    // ---- pass B.  The projected columns of G iterations are loaded TOGETHER (only the quads with a match), then the
    // iterations are emitted in order: one exposed memory latency per G iterations instead of one per iteration (round 6: at
    // 1 % selectivity nearly every (iteration, wave) holds a match, so a tile's pass B was 16 dependent load -> store
    // round trips: 3.0 ms per 1 B rows and one column where the filter column streams in 0.65 ms).  G is what the member's
    // registers allow at four waves per SIMD.
    constexpr int G = (XF || COL) ? 1 : NT == 1 ? 2 : 1;
    auto load_iter = [&](int u, int64_t (&vals)[NT][4]) __attribute__((always_inline)) {
      const uint32_t mm = (uint32_t)(m >> (4 * u)) & 0xfu;
      const int64_t r = row0 + (int64_t)u * kIterRows + tid * 4;
      const bool inside = row0 + (int64_t)(u + 1) * kIterRows <= n;  // (uniform)
      if (mm) {
        if (inside) {  // every projected column of the quad first: independent 16-byte loads (a 4-byte column's quad waits,
                       // packed, in the first two value registers)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int8_t* base = fc[a.tcol[t]];
            if (a.tkind[t] == 0) {
              const v2i64 x = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + (r >> 2) * 2);
              const v2i64 y = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + (r >> 2) * 2 + 1);
              vals[t][0] = x.x; vals[t][1] = x.y; vals[t][2] = y.x; vals[t][3] = y.y;
            } else {
              const v2i64 x = __builtin_nontemporal_load((const MQ_GLOBAL v2i64*)base + (r >> 2));
              vals[t][0] = x.x; vals[t][1] = x.y;
              MQ_NO_STORE_SINK();
            }
          }
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int8_t* base = fc[a.tcol[t]];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              int64_t v = 0;
              if ((mm >> i) & 1u) {
                if (a.tkind[t] == 0) v = fast::load_one<int64_t>(base, r + i);
                else {
                  const int32_t x = fast::load_one<int32_t>(base, r + i);
                  v = a.tkind[t] != 2 ? (int64_t)x : dbl_bits((double)bits_flt(x));
                }
              }
              vals[t][i] = v;
            }
          }
          MQ_NO_STORE_SINK();
        }
      }
    };
    auto emit_iter = [&](int u, int64_t (&vals)[NT][4]) __attribute__((always_inline)) {
      const uint32_t mm = (uint32_t)(m >> (4 * u)) & 0xfu;
      // per bit plane (row i of every lane's quad): matches in lower lanes, matches in the wave (ballots: every lane takes part)
      unsigned long long bal[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) bal[b] = __ballot((mm >> b) & 1u);
      if ((bal[0] | bal[1] | bal[2] | bal[3]) == 0) return;  // (uniform)
      const int64_t r = row0 + (int64_t)u * kIterRows + tid * 4;
      const bool inside = row0 + (int64_t)(u + 1) * kIterRows <= n;  // (uniform)
      if (mm && inside) {  // (a 4-byte column's quad waited, packed, in the first two value registers)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (a.tkind[t] == 0) continue;
          const int32_t x0 = (int32_t)(uint32_t)(uint64_t)vals[t][0], x1 = (int32_t)(uint32_t)((uint64_t)vals[t][0] >> 32);
          const int32_t x2 = (int32_t)(uint32_t)(uint64_t)vals[t][1], x3 = (int32_t)(uint32_t)((uint64_t)vals[t][1] >> 32);
          if (a.tkind[t] != 2) {  // INT32 sign-extended (or, in a columnar buffer, the 4 bytes as they are)
            vals[t][0] = x0; vals[t][1] = x1; vals[t][2] = x2; vals[t][3] = x3;
          } else {
            vals[t][0] = dbl_bits((double)bits_flt(x0)); vals[t][1] = dbl_bits((double)bits_flt(x1));
            vals[t][2] = dbl_bits((double)bits_flt(x2)); vals[t][3] = dbl_bits((double)bits_flt(x3));
          }
        }
      }
      // targets that are forms: [CAST](column) <op> literal on the quad just loaded (wave-uniform branches on the form)
      int32_t ferr[4] = {0, 0, 0, 0};
      if (XF && mm) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (!a.f_on[t]) continue;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int32_t e = 0;
            vals[t][i] = form_eval(a.f_src_type[t], a.f_cast_to[t], a.f_cast_flags[t], a.f_op[t], a.f_type[t], a.f_flags[t],
                                   a.f_lit_first[t] != 0, a.f_lit[t], vals[t][i], e);
            if (e && !ferr[i]) ferr[i] = e;
          }
        }
      }
      uint32_t step_off = 0;  // matches of the wave in the earlier steps of this iteration
#pragma unroll
      for (int h = 0; h < 64 / LH; ++h) {
        const unsigned long long hmask = LH == 64 ? ~0ull : (h == 0 ? 0xffffffffull : 0xffffffff00000000ull);
        uint32_t step_cnt = 0, k = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const unsigned long long hb = bal[b] & hmask;
          step_cnt += (uint32_t)__popcll(hb);
          k += __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u));
        }
        if (step_cnt == 0) continue;  // (uniform)
        const bool in_step = LH == 64 || (lane >> 5) == h;
        const int64_t wfirst = tile_base + s_cnt[u * kWaves + wave] + step_off;  // rank of the wave's first entry of this step
        // the wave's entries of this step, in rank order, into its stage ...
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (!in_step || !((mm >> i) & 1u)) continue;
          // (the error of a form counts for a row that is EMITTED: one past the scan limit / the buffer's end is not)
          if (XF && ferr[i] && !err && wfirst + (int64_t)k < a.entry_count) err = ferr[i];
          if (!COL) {
            int64_t* row = (int64_t*)stage + (size_t)k * rq;
            row[0] = r + i;
#pragma unroll
            for (int t = 0; t < NT; ++t) row[1 + t] = vals[t][i];
          } else {
            ((int64_t*)stage)[k] = r + i;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              if (a.tw[t] == 8) ((int64_t*)(stage + run_off[t]))[k] = vals[t][i];
              else ((int32_t*)(stage + run_off[t]))[k] = (int32_t)vals[t][i];
            }
          }
          ++k;
        }
        // ... and out of it as contiguous runs: the wave's ranks are adjacent, so are its bytes (64 lanes x 8 / 16 bytes
        // per store instead of one small piece per line)
        __builtin_amdgcn_wave_barrier();
        int64_t n_rows = (int64_t)step_cnt;
        if (wfirst + n_rows > a.entry_count) n_rows = a.entry_count > wfirst ? a.entry_count - wfirst : 0;
        if (!COL) {
          const int64_t nq = n_rows * rq;
          int64_t* dst = a.out + wfirst * rq;
          if ((((uintptr_t)dst) & 15) == 0) {
            const int64_t n2 = nq >> 1;
            for (int64_t q = lane; q < n2; q += 64) ((v2i64*)dst)[q] = ((const v2i64*)stage)[q];
            if ((nq & 1) && lane == 0) dst[nq - 1] = ((const int64_t*)stage)[nq - 1];
          } else {
            for (int64_t q = lane; q < nq; q += 64) dst[q] = ((const int64_t*)stage)[q];
          }
        } else {
          for (int64_t q = lane; q < n_rows; q += 64) a.out[wfirst + q] = ((const int64_t*)stage)[q];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            char* col = (char*)a.out + a.tcol_off[t];
            if (a.tw[t] == 8) {
              for (int64_t q = lane; q < n_rows; q += 64) ((int64_t*)col)[wfirst + q] = ((const int64_t*)(stage + run_off[t]))[q];
            } else {
              for (int64_t q = lane; q < n_rows; q += 64) ((int32_t*)col)[wfirst + q] = ((const int32_t*)(stage + run_off[t]))[q];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        step_off += step_cnt;
      }
    };
#pragma unroll 1
    for (int u0 = 0; u0 < kIters; u0 += G) {
      int64_t vals[G][NT][4];
#pragma unroll
      for (int g = 0; g < G; ++g) load_iter(u0 + g, vals[g]);
#pragma unroll
      for (int g = 0; g < G; ++g) emit_iter(u0 + g, vals[g]);
    }
  }
  if (XF && err) atomicCAS(a.d_err, 0, err);
}

// HJ: the step joins a one-to-one hash table (fast_quals is off: every row takes row_passes, which probes for an INNER
// join; pass B probes again for the rows it writes and reads the inner columns through the matched row)
template <bool HJ>
__global__ __launch_bounds__(kBlock) void k_proj_compact_lds(DevPlan p, ProjArgs a) {
  extern __shared__ __attribute__((aligned(16))) char s_img[];
  __shared__ unsigned long long s_wave[kWaves];
  __shared__ long long s_bcast[2];

  const ProjSpec& ps = a.ps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_phys = ps.n_phys_cols;
  int32_t err = 0;
  // the expressions' LDS area: programs (copied once), stack, values
  ProjExprLds L{};
  L.stk.tid = tid;
  if (a.xs && a.x_lds_off >= 0) {
    XNode* const s_prog = (XNode*)(s_img + a.x_lds_off);
    for (int w = tid; w < a.xs->n * MI355Q_MAX_EXPR_NODES; w += kBlock) {
      const DevExprNode& nd = a.xs->e[w / MI355Q_MAX_EXPR_NODES].nodes[w % MI355Q_MAX_EXPR_NODES];
      XNode x;
      x.h = nd.flags >> kExHandlerShift;
      x.pad_ = 0;
      x.lit = nd.ilit;
      s_prog[w] = x;
    }
    L.prog = s_prog;
    L.stk.st = (int64_t*)(s_prog + MI355Q_MAX_EXPRS * MI355Q_MAX_EXPR_NODES);
    L.xv = L.stk.st + (size_t)a.x_below * kXJ * kBlock;
    __syncthreads();
  }

  for (;;) {
    // ---- the next tile, in ticket order (a tile's predecessors have all been taken by running workgroups)
    if (tid == 0) s_bcast[0] = (long long)atomicAdd(&a.counters[0], 1ull);
    __syncthreads();
    const int64_t tile = s_bcast[0];
    __syncthreads();
    if (tile >= a.n_tiles) break;
    int f = 0;
    {
      int lo = 0, hi = a.n_frags;  // tile_start[lo] <= tile < tile_start[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.tile_start[mid] <= tile) lo = mid;
        else hi = mid;
      }
      f = lo;
    }
    const int8_t* const* fc = a.cols + (size_t)f * ps.n_cols_table;
    const int64_t n = a.num_rows[f];
    const int64_t row0 = (tile - a.tile_start[f]) * kTileRows;  // first row of the tile in its fragment

    const uint64_t m = tile_filter<true, HJ>(p, a, L, fc, n, row0, &err);

    // ---- the tile's count, its descriptor, and the entries before it
    {
      unsigned long long c = (unsigned long long)__popcll(m);
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
      if (lane == 0) s_wave[wave] = c;
    }
    __syncthreads();
    unsigned long long tile_count = 0;
    for (int w = 0; w < kWaves; ++w) tile_count += s_wave[w];
    __syncthreads();
    if (wave == 0) {
      const unsigned long long excl = tile_lookback(a.desc, a.counters, a.n_tiles, tile, tile_count);
      if (lane == 0) s_bcast[1] = (long long)excl;
    }
    __syncthreads();
    const int64_t tile_base = s_bcast[1];
    if (tile_count == 0 || tile_base >= ps.entry_count) continue;  // nothing to write (an empty tile, or past a scan limit)

    // ---- pass B: sub-tile by sub-tile, assemble the entries in LDS and stream them out
    int64_t done = 0;  // entries of this tile already out
    const int sub = a.sub_iters;
    for (int u0 = 0; u0 < kIters; u0 += sub) {
      const uint32_t mA = (uint32_t)(m >> (4 * u0)) & 0xfu;
      const uint32_t mB = sub == 2 ? (uint32_t)(m >> (4 * (u0 + 1))) & 0xfu : 0u;
      const unsigned long long packed = (unsigned long long)__popc(mA) | ((unsigned long long)__popc(mB) << 32);
      unsigned long long wtot;
      const unsigned long long wex = wave_excl_scan_u32x2(packed, &wtot);
      if (lane == 0) s_wave[wave] = wtot;
      __syncthreads();
      unsigned long long before = 0, all = 0;
      for (int w = 0; w < kWaves; ++w) {
        if (w < wave) before += s_wave[w];
        all += s_wave[w];
      }
      const uint32_t totA = (uint32_t)all, totB = (uint32_t)(all >> 32);
      const uint32_t exA = (uint32_t)(wex + before), exB = (uint32_t)((wex + before) >> 32);
      const int64_t sub_total = (int64_t)totA + totB;
      const int64_t gbase = tile_base + done;  // output entry of the sub-tile's first matching row
      int64_t n_write = ps.entry_count - gbase;
      n_write = n_write < 0 ? 0 : n_write > sub_total ? sub_total : n_write;
      if (sub_total == 0 || n_write == 0) {
        __syncthreads();
        done += sub_total;
        if (gbase >= ps.entry_count) break;
        continue;
      }
      // columnar image: one run per column, each 8-byte aligned; row-wise image: whole rows
      int32_t img_off[MI355Q_MAX_TARGETS + 1];
      if (ps.columnar) {
        int32_t o = 0;
        img_off[0] = 0;
        o = (int32_t)sub_total * 8;
        for (int t = 0; t < ps.n_targets; ++t) {
          img_off[t + 1] = o;
          o += ((int32_t)sub_total * ps.t[t].width + 7) & ~7;
        }
      }
      for (int half = 0; half < sub; ++half) {
        const uint32_t mm = half ? mB : mA;
        if (!mm) continue;
        const int64_t r = row0 + (int64_t)(u0 + half) * kIterRows + tid * 4;
        const uint32_t first = half ? totA + exB : exA;  // rank of the quad's first matching row in the sub-tile
        // the entry's key: the row's offset in its fragment
        {
          uint32_t k = first;
          for (int i = 0; i < 4; ++i) {
            if (!((mm >> i) & 1u)) continue;
            if (ps.columnar) ((int64_t*)s_img)[k] = r + i;
            else ((int64_t*)s_img)[(size_t)k * ps.row_quad] = r + i;
            ++k;
          }
        }
        int64_t xv[4][MI355Q_MAX_EXPRS];
        if (a.xs && L.xv) {  // the quad's four rows together (a row past the fragment's end repeats the last one)
          int64_t p4[kXJ];
          int32_t e4[kXJ] = {0, 0, 0, 0};
#pragma unroll
          for (int i = 0; i < kXJ; ++i) p4[i] = r + i < n ? r + i : n - 1;
          eval_exprs_lds(*a.xs, L, (1u << a.xs->n) - 1u, fc, p4, e4);
          // only a row that is EMITTED counts: one whose rank lies past the scan limit / the buffer's end is never
          // written, and the reference's loop stops at max_matched before it would evaluate it
          uint32_t kk = first;
#pragma unroll
          for (int i = 0; i < kXJ; ++i) {
            if (!((mm >> i) & 1u)) continue;
            if (e4[i] && !err && (int64_t)kk < n_write) err = e4[i];
            ++kk;
          }
        } else if (a.xs) {
          uint32_t kk = first;
          for (int i = 0; i < 4; ++i) {
            if (!((mm >> i) & 1u)) continue;
            int32_t e1 = 0;
            eval_exprs(*a.xs, (1u << a.xs->n) - 1u, fc, r + i, xv[i], &e1);
            if (e1 && !err && (int64_t)kk < n_write) err = e1;
            ++kk;
          }
        }
        int64_t inner_pos[4] = {-1, -1, -1, -1};
        if (HJ) {
          for (int i = 0; i < 4; ++i)
            if ((mm >> i) & 1u) inner_pos[i] = proj_join_row(p, fc, r + i);
        }
        for (int t = 0; t < ps.n_targets; ++t) {
          const ProjTarget& pt = ps.t[t];
          int64_t vals[4];
          if (HJ && pt.col >= kProjInnerCol) {  // an inner column: the matched row's value, or NULL (LEFT join, no match)
            const int8_t* base = p.inner_cols[pt.col - kProjInnerCol];
            for (int i = 0; i < 4; ++i)
              vals[i] = !((mm >> i) & 1u) ? 0 : inner_pos[i] >= 0 ? col_value_bits(base, pt.code, inner_pos[i]) : proj_null_bits(pt.code);
          } else if (pt.col >= n_phys) {
            if (L.xv) {
              for (int i = 0; i < 4; ++i) vals[i] = L.xv[(size_t)((pt.col - n_phys) * kXJ + i) * kBlock + tid];
            } else {
              for (int i = 0; i < 4; ++i) vals[i] = xv[i][pt.col - n_phys];
            }
          } else {
            const int8_t* base = fc[pt.col];
            const int w = type_width(pt.code);
            if (((a.vec_mask >> pt.col) & 1) && r + 4 <= n) {
              RawQuad raw;
              load_raw_quad(base, w, r >> 2, raw);
#pragma unroll
              for (int i = 0; i < 4; ++i) vals[i] = value_of_raw(pt.code, raw_elem(raw, w, i));
            } else {
              for (int i = 0; i < 4; ++i) vals[i] = ((mm >> i) & 1u) ? col_value_bits(base, pt.code, r + i) : 0;
            }
          }
          uint32_t k = first;
          for (int i = 0; i < 4; ++i) {
            if (!((mm >> i) & 1u)) continue;
            int64_t v = vals[i];
            if (pt.kind == PROJ_F32_TO_F64) v = dbl_bits((double)bits_flt((int32_t)(uint32_t)v));
            if (!ps.columnar) {
              ((int64_t*)s_img)[(size_t)k * ps.row_quad + 1 + t] = v;
            } else {
              char* dst = s_img + img_off[t + 1];
              switch (pt.width) {
                case 1: ((int8_t*)dst)[k] = (int8_t)v; break;
                case 2: ((int16_t*)dst)[k] = (int16_t)v; break;
                case 4: ((int32_t*)dst)[k] = (int32_t)(uint32_t)v; break;
                default: ((int64_t*)dst)[k] = v;
              }
            }
            ++k;
          }
        }
      }
      __syncthreads();
      // ---- the image leaves as contiguous stores
      if (!ps.columnar) {
        const int64_t nq = n_write * ps.row_quad;
        int64_t* dst = (int64_t*)a.out + gbase * ps.row_quad;
        const int64_t* src = (const int64_t*)s_img;
        for (int64_t i = tid; i < nq; i += kBlock) dst[i] = src[i];
      } else {
        {
          int64_t* dst = (int64_t*)a.out + gbase;
          const int64_t* src = (const int64_t*)s_img;
          for (int64_t i = tid; i < n_write; i += kBlock) dst[i] = src[i];
        }
        for (int t = 0; t < ps.n_targets; ++t) {
          const ProjTarget& pt = ps.t[t];
          const char* src = s_img + img_off[t + 1];
          int8_t* dst = a.out + pt.col_off + gbase * pt.width;
          switch (pt.width) {
            case 1:
              for (int64_t i = tid; i < n_write; i += kBlock) dst[i] = ((const int8_t*)src)[i];
              break;
            case 2:
              for (int64_t i = tid; i < n_write; i += kBlock) ((int16_t*)dst)[i] = ((const int16_t*)src)[i];
              break;
            case 4:
              for (int64_t i = tid; i < n_write; i += kBlock) ((int32_t*)dst)[i] = ((const int32_t*)src)[i];
              break;
            default:
              for (int64_t i = tid; i < n_write; i += kBlock) ((int64_t*)dst)[i] = ((const int64_t*)src)[i];
          }
        }
      }
      __syncthreads();
      done += sub_total;
    }
  }
  if (err) atomicCAS(a.d_err, 0, err);
}

// ---- ONE-TO-MANY joins under a Projection (round 6): every matching inner row of an outer row is an entry — the matching SET
// of HashJoin::codegenMatchingSet (HashJoin.cpp:209) walked by the join loop around the row function's body (IRCodegen.cpp:635
// buildJoinLoops, :1233 codegenJoinLoops); the table is [offsets | counts | payload] (perfect) or [keys | offsets | counts |
// payload] (keyed), HashJoinRuntime.cpp:702 count_matches, :1027 fill_row_ids.  The compaction counts ENTRIES instead of
// rows: pass A probes every row that passes the quals for its match count (a LEFT join counts an unmatched row once, with
// the inner columns' NULLs), the (iteration, wave) totals are scanned, the tile's total goes through the same look-back;
// pass B probes again and every lane writes its rows' entries — a row's matches in payload order, rows in (fragment, row)
// order, as the reference's CPU executor emits them — straight to the buffer (no LDS image: a row may have any number of
// matches).  Plain column targets of either side; quals of any kind without expressions.
MQ_D JoinMatch proj_join_matches(const DevPlan& p, const int8_t* const* fc, int64_t pos) {
  int64_t jk[MI355Q_MAX_GROUP_COLS];
  bool null_key = false;
  for (int i = 0; i < p.join_n_keys; ++i) {
    jk[i] = decode_int(fc[p.join_cols[i]], p.join_types[i], pos);
    null_key = null_key || (p.join_nullables[i] && jk[i] == int_null_of(p.join_types[i]));
  }
  if (null_key) return JoinMatch{nullptr, -1, 0};
  return join_lookup(p, jk);
}
__global__ __launch_bounds__(kBlock) void k_proj_join_1n(DevPlan p, ProjArgs a) {
  __shared__ unsigned long long s_cnt[kIters * kWaves];  // entries of (iteration, wave); then their exclusive prefix in that order
  __shared__ long long s_bcast[2];
  const ProjSpec& ps = a.ps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool left = p.join_kind == MI355Q_JOIN_LEFT;
  int32_t err = 0;
  ProjExprLds L{};
  for (;;) {
    if (tid == 0) s_bcast[0] = (long long)atomicAdd(&a.counters[0], 1ull);
    __syncthreads();
    const int64_t tile = s_bcast[0];
    __syncthreads();
    if (tile >= a.n_tiles) break;
    int f = 0;
    {
      int lo = 0, hi = a.n_frags;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.tile_start[mid] <= tile) lo = mid;
        else hi = mid;
      }
      f = lo;
    }
    const int8_t* const* fc = a.cols + (size_t)f * ps.n_cols_table;
    const int64_t n = a.num_rows[f];
    const int64_t row0 = (tile - a.tile_start[f]) * kTileRows;
    const uint64_t m = tile_filter<false, false>(p, a, L, fc, n, row0, &err);  // the quals alone
    // ---- pass A: entries per (iteration, wave)
#pragma unroll 1
    for (int u = 0; u < kIters; ++u) {
      const uint32_t mm = (uint32_t)(m >> (4 * u)) & 0xfu;
      const int64_t r = row0 + (int64_t)u * kIterRows + tid * 4;
      unsigned long long c = 0;
      for (int i = 0; i < 4; ++i) {
        if (!((mm >> i) & 1u)) continue;
        const int32_t k = proj_join_matches(p, fc, r + i).count;
        c += (unsigned long long)(k > 0 ? k : left ? 1 : 0);
      }
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
      if (lane == 0) s_cnt[u * kWaves + wave] = c;
    }
    __syncthreads();
    if (wave == 0) {
      const unsigned long long v = s_cnt[lane];
      unsigned long long inc = v;
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl(inc, lane - d < 0 ? lane : lane - d, 64);
        if (lane >= d) inc += o;
      }
      const unsigned long long tile_count = __shfl(inc, 63, 64);
      s_cnt[lane] = inc - v;
      const unsigned long long excl = tile_lookback(a.desc, a.counters, a.n_tiles, tile, tile_count);
      if (lane == 0) s_bcast[1] = (long long)excl;
    }
    __syncthreads();
    const int64_t tile_base = s_bcast[1];
    if (tile_base >= ps.entry_count) continue;  // past a scan limit (or a full buffer)
    // ---- pass B: probe again, write the entries
#pragma unroll 1
    for (int u = 0; u < kIters; ++u) {
      const uint32_t mm = (uint32_t)(m >> (4 * u)) & 0xfu;
      const int64_t r = row0 + (int64_t)u * kIterRows + tid * 4;
      JoinMatch jm[4];
      unsigned long long c = 0;
      for (int i = 0; i < 4; ++i) {
        jm[i] = JoinMatch{nullptr, -1, 0};
        if (!((mm >> i) & 1u)) continue;
        jm[i] = proj_join_matches(p, fc, r + i);
        if (jm[i].count <= 0) {
          jm[i] = JoinMatch{nullptr, -1, left ? 1 : 0};   // LEFT join: once, with the inner columns' NULLs
        }
        c += (unsigned long long)jm[i].count;
      }
      unsigned long long inc = c;
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl(inc, lane - d < 0 ? lane : lane - d, 64);
        if (lane >= d) inc += o;
      }
      int64_t e = tile_base + (int64_t)s_cnt[u * kWaves + wave] + (int64_t)(inc - c);
      for (int i = 0; i < 4; ++i) {
        for (int32_t j = 0; j < jm[i].count; ++j, ++e) {
          if (e >= ps.entry_count) break;
          const int64_t inner_pos = jm[i].ids ? (int64_t)jm[i].ids[j] : jm[i].single;
          if (ps.columnar) ((int64_t*)a.out)[e] = r + i;
          else ((int64_t*)a.out)[e * ps.row_quad] = r + i;
          for (int t = 0; t < ps.n_targets; ++t) {
            const ProjTarget& pt = ps.t[t];
            int64_t v;
            if (pt.col >= kProjInnerCol) v = inner_pos >= 0 ? col_value_bits(p.inner_cols[pt.col - kProjInnerCol], pt.code, inner_pos) : proj_null_bits(pt.code);
            else v = col_value_bits(fc[pt.col], pt.code, r + i);
            if (pt.kind == PROJ_F32_TO_F64) v = dbl_bits((double)bits_flt((int32_t)(uint32_t)v));
            if (!ps.columnar) {
              ((int64_t*)a.out)[e * ps.row_quad + 1 + t] = v;
            } else {
              char* dst = (char*)a.out + pt.col_off;
              switch (pt.width) {
                case 1: ((int8_t*)dst)[e] = (int8_t)v; break;
                case 2: ((int16_t*)dst)[e] = (int16_t)v; break;
                case 4: ((int32_t*)dst)[e] = (int32_t)(uint32_t)v; break;
                default: ((int64_t*)dst)[e] = v;
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (err) atomicCAS(a.d_err, 0, err);
}

// entries [min(total_matched, entry_count), entry_count) of the buffer: the EMPTY_KEY_64 key — and, row-wise, the zero
// slots — of an initialised buffer (QueryMemoryInitializer::initRowGroups :617-698; initColumnarGroups :713-738 leaves
// the slot columns of a columnar projection alone)
__global__ __launch_bounds__(kBlock) void k_proj_init_tail(ProjSpec ps, const unsigned long long* counters, int64_t* out) {
  const int64_t live = (int64_t)counters[1] < ps.entry_count ? (int64_t)counters[1] : ps.entry_count;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  if (ps.columnar) {
    for (int64_t e = live + (int64_t)blockIdx.x * kBlock + threadIdx.x; e < ps.entry_count; e += stride) out[e] = kEmptyKey64;
    return;
  }
  const int64_t q0 = live * ps.row_quad, q1 = ps.entry_count * ps.row_quad;
  for (int64_t i = q0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < q1; i += stride)
    out[i] = (i % ps.row_quad) == 0 ? kEmptyKey64 : 0;
}

// ResultSet::rowCount of a projection buffer whose total is not known to the host (a wrapped buffer): the entries
// whose key is not EMPTY_KEY_64
__global__ __launch_bounds__(kBlock) void k_proj_count_live(const int64_t* keys, int64_t stride_quads, int64_t entries,
                                                            unsigned long long* count) {
  unsigned long long c = 0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < entries; e += stride) c += keys[e * stride_quads] != kEmptyKey64;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

}  // namespace

int64_t projection_tile_rows() { return kTileRows; }

void projection_forms(const DevExprSet& xs, const ProjSpec& ps, uint32_t qual_expr_mask, ProjForms* out) {
  std::memset(out, 0, sizeof(*out));
  if (qual_expr_mask != 0) return;  // (an expression inside a qual: the general member evaluates the filter's expressions)
  uint32_t used = 0;
  for (int t = 0; t < ps.n_targets; ++t) {
    const ProjTarget& pt = ps.t[t];
    if (pt.col < ps.n_phys_cols || pt.col >= kProjInnerCol) continue;
    const int k = pt.col - ps.n_phys_cols;
    if (k < 0 || k >= xs.n) return;
    used |= 1u << k;
    const DevExpr& e = xs.e[k];
    ProjForm& f = out->f[t];
    // [COL][CAST]? | [COL][CAST]?[LIT][op] | [LIT][COL][CAST]?[op]
    int i = 0;
    const bool lit_first = e.n_nodes >= 1 && e.nodes[0].op == MI355Q_EX_LIT;
    const DevExprNode* lit = nullptr;
    if (lit_first) lit = &e.nodes[i++];
    if (i >= e.n_nodes || e.nodes[i].op != MI355Q_EX_COL) return;
    const DevExprNode& col = e.nodes[i++];
    const int code = (int)col.ilit;
    if (col.arg < 0 || col.arg >= xs.n_cols || !rp_type_ok(code) || col.type != code) return;
    f.src_col = col.arg;
    f.src_code = code;
    int cur = code;
    if (i < e.n_nodes && e.nodes[i].op == MI355Q_EX_CAST) {
      const DevExprNode& c = e.nodes[i++];
      if (!rp_type_ok(c.type) || c.arg != cur) return;
      f.cast_to = c.type;
      f.cast_flags = c.flags & (EXF_NULLABLE | EXF_LHS_NULLABLE);
      if (ex_is_int(c.type) && (!ex_is_int(cur) || plain_width(c.type) < plain_width(cur))) f.can_raise = 1;
      cur = c.type;
    }
    if (i < e.n_nodes) {
      if (!lit_first) {
        if (e.nodes[i].op != MI355Q_EX_LIT) return;
        lit = &e.nodes[i++];
      }
      if (i != e.n_nodes - 1) return;
      const DevExprNode& op = e.nodes[i];
      if (op.op != MI355Q_EX_ADD && op.op != MI355Q_EX_SUB && op.op != MI355Q_EX_MUL) return;
      if (!lit || lit->arg != 0 || lit->type != op.type || cur != op.type || !rp_type_ok(op.type)) return;
      f.op = op.op;
      f.type = op.type;
      f.flags = op.flags & (EXF_NULLABLE | EXF_LHS_NULLABLE | EXF_RHS_NULLABLE);
      f.lit_first = lit_first;
      f.lit = ex_lit(*lit);
      if (ex_is_int(op.type)) f.can_raise = 1;
    } else if (lit_first || !f.cast_to) {
      return;  // (a literal alone, a bare column: not shapes a binding states as an expression)
    }
    if (e.type != (f.op ? f.type : f.cast_to)) return;
    f.on = 1;
  }
  if (used != (xs.n >= 32 ? ~0u : (1u << xs.n) - 1u)) return;  // (an expression no target reads would still be evaluated — and could raise)
  out->ok = 1;
}

int64_t projection_scratch_bytes(int n_frags, const int64_t* h_num_rows) {
  int64_t tiles = 0;
  for (int f = 0; f < n_frags; ++f) tiles += (h_num_rows[f] + kTileRows - 1) / kTileRows;
  // counters (64 B) | tile_start [n_frags + 1] | descriptors [tiles] | the split route's match bits [tiles][kBlock]
  return 64 + (((int64_t)(n_frags + 1) * 8 + 63) & ~(int64_t)63) + ((tiles * 8 + 63) & ~(int64_t)63) + 64 +
         (tiles >= kSplitMinTiles || tune_knobs().pass_rows == -3 ? tiles * kBlock * 8 + 2 * ((tiles * 4 + 127) & ~(int64_t)63) : 0);
}

hipError_t launch_projection(const DevPlan& p, const ProjSpec& ps, const DevExprSet* d_xs, uint32_t qual_expr_mask,
                             const FragView& fv, void* scratch, void* out, int32_t* d_err, unsigned long long** d_total,
                             int n_cus, hipStream_t s, LaunchStats* st, const ProjForms* forms) {
  char* sp = (char*)scratch;
  unsigned long long* counters = (unsigned long long*)sp;
  int64_t* tile_start = (int64_t*)(sp + 64);
  const int64_t ts_bytes = ((int64_t)(fv.n_frags + 1) * 8 + 63) & ~(int64_t)63;
  unsigned long long* desc = (unsigned long long*)(sp + 64 + ts_bytes);
  std::vector<int64_t> h_ts((size_t)fv.n_frags + 1);
  int64_t tiles = 0;
  for (int f = 0; f < fv.n_frags; ++f) {
    h_ts[f] = tiles;
    tiles += (fv.h_num_rows[f] + kTileRows - 1) / kTileRows;
  }
  h_ts[fv.n_frags] = tiles;
  *d_total = counters + 1;
  hipError_t e = hipMemsetAsync(sp, 0, (size_t)(64 + ts_bytes + tiles * 8), s);
  if (e != hipSuccess) return e;
  // (a pageable source: the copy is staged by the runtime before the call returns)
  e = hipMemcpyAsync(tile_start, h_ts.data(), h_ts.size() * 8, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(s);  // h_ts goes out of scope
  if (e != hipSuccess) return e;

  ProjArgs a{};
  a.ps = ps;
  a.xs = d_xs;
  a.qual_expr_mask = qual_expr_mask;
  a.n_frags = fv.n_frags;
  a.cols = fv.d_cols;
  a.num_rows = fv.d_num_rows;
  a.tile_start = tile_start;
  a.n_tiles = tiles;
  a.desc = desc;
  a.counters = counters;
  a.out = (int8_t*)out;
  a.d_err = d_err;
  int row_bytes = 8;
  for (int t = 0; t < ps.n_targets; ++t) row_bytes += ps.columnar ? ps.t[t].width : 8;
  a.sub_iters = row_bytes <= 36 ? 2 : 1;
  a.row_quals = 0;
  for (int k = 0; k < p.n_quals; ++k) a.row_quals |= p.quals[k].or_group != 0;
  a.vec_mask = 0;
  for (int c = 0; c < ps.n_phys_cols && c < 31; ++c)
    if (fast::all_aligned16(fv, c)) a.vec_mask |= 1 << c;
  for (int k = 0; k < p.n_quals && k < MI355Q_MAX_QUALS; ++k) {
    a.qmode[k] = 0;
    const DevQual& dq = p.quals[k];
    if (dq.col < ps.n_phys_cols && (dq.type == MI355Q_INT32 || dq.type == MI355Q_INT64) && fast::make_range_filter(dq, &a.qf[k])) {
      a.qmode[k] = dq.type == MI355Q_INT32 ? 1 : 2;
      if (dq.type == MI355Q_INT32) {  // compared in 32 bits: an empty range stays empty, everything else is inside the type
        if (a.qf[k].lo > a.qf[k].hi) {
          a.qf[k].lo = 1;
          a.qf[k].hi = 0;
        } else {
          a.qf[k].lo = std::max<int64_t>(a.qf[k].lo, INT32_MIN);
          a.qf[k].hi = std::min<int64_t>(a.qf[k].hi, INT32_MAX);
        }
      }
    }
  }
  a.out16 = (((uintptr_t)out) & 15) == 0;
  a.fast_quals = !a.row_quals && qual_expr_mask == 0 && p.join_col < 0;  // (a join: every row takes row_passes, which probes)
  for (int k = 0; k < p.n_quals; ++k)
    a.fast_quals = a.fast_quals && p.quals[k].col < ps.n_phys_cols && p.quals[k].col < 31 && ((a.vec_mask >> p.quals[k].col) & 1);
  a.fast_targets = 1;
  for (int t = 0; t < ps.n_targets; ++t)
    a.fast_targets = a.fast_targets && ps.t[t].col < ps.n_phys_cols && ps.t[t].col < 31 && ((a.vec_mask >> ps.t[t].col) & 1);
  // the fast row-wise member: range quals over plain INT32 / INT64 columns, plain 4- / 8-byte targets, aligned chunks
  const bool forms_ok = forms && forms->ok && !(tune_knobs().flags & MI355Q_OPT_LDS_GENERIC_MEMBER);
  bool fast_ok = (!d_xs || forms_ok) && !a.row_quals && p.join_col < 0 && ps.n_targets >= 1 && ps.n_targets <= 8 && tune_knobs().pass_rows != -1;
  FastArgs fa{};
  fa.d_err = d_err;
  for (int k = 0; k < p.n_quals && fast_ok; ++k) {
    const int c = p.quals[k].col;
    fa.qmode[k] = a.qmode[k];
    fa.qf[k] = a.qf[k];
    // a plain 1-byte column (TINYINT / BOOLEAN; the row mask of a compiled filter): the fast member's own mode
    if (a.qmode[k] == 0 && p.quals[k].type == MI355Q_INT8 && c < ps.n_phys_cols && fast::make_range_filter(p.quals[k], &fa.qf[k], true)) fa.qmode[k] = 3;
    fast_ok = fa.qmode[k] != 0 && c < ps.n_phys_cols && c < 31 && ((a.vec_mask >> c) & 1);
    fa.qcol[k] = c;
  }
  for (int t = 0; t < ps.n_targets && fast_ok; ++t) {
    const ProjTarget& pt = ps.t[t];
    if (pt.col >= ps.n_phys_cols && pt.col < kProjInnerCol) {  // an expression: its form reads ONE plain column
      const ProjForm& pf = forms->f[t];
      fast_ok = forms_ok && pf.on && ps.n_targets <= 4 && pf.src_col < ps.n_phys_cols && pf.src_col < 31 && ((a.vec_mask >> pf.src_col) & 1);
      fa.tcol[t] = pf.src_col;
      fa.tkind[t] = pf.src_code == MI355Q_INT32 ? 1 : 0;
      fa.tw[t] = ps.columnar ? pt.width : 8;
      fa.tcol_off[t] = pt.col_off;
      fa.any_form = 1;
      fa.f_on[t] = 1;
      fa.f_src_type[t] = pf.src_code;
      fa.f_cast_to[t] = pf.cast_to;
      fa.f_cast_flags[t] = pf.cast_flags;
      fa.f_op[t] = pf.op;
      fa.f_type[t] = pf.type;
      fa.f_flags[t] = pf.flags;
      fa.f_lit_first[t] = pf.lit_first;
      fa.f_lit[t] = pf.lit;
      continue;
    }
    fast_ok = pt.col < ps.n_phys_cols && pt.col < 31 && ((a.vec_mask >> pt.col) & 1) &&
              (pt.code == MI355Q_INT64 || pt.code == MI355Q_DOUBLE || pt.code == MI355Q_INT32 || pt.code == MI355Q_FLOAT);
    fa.tcol[t] = pt.col;
    // (a columnar buffer keeps a FLOAT as its 4 bytes: the same move as an INT32)
    fa.tkind[t] = (pt.code == MI355Q_INT64 || pt.code == MI355Q_DOUBLE) ? 0 : (pt.code == MI355Q_INT32 || ps.columnar) ? 1 : 2;
    fa.tw[t] = ps.columnar ? pt.width : 8;
    fa.tcol_off[t] = pt.col_off;
  }
  // the LDS image of one sub-tile (general member): every row of it may match; columnar runs are padded to 8 bytes each
  size_t lds = fast_ok ? 0 : (size_t)a.sub_iters * kIterRows * row_bytes + (ps.columnar ? 8 * (size_t)(ps.n_targets + 1) : 0);
  a.x_lds_off = -1;
  a.x_below = 0;
  const int x_deepest = ps.x_info & 0xff, x_n = ps.x_info >> 8;
  if (d_xs && !fast_ok && x_deepest > 0) {  // + the expressions' area (programs, stack below the top, values)
    const size_t img = (lds + 15) & ~(size_t)15;
    const size_t xarea = (size_t)MI355Q_MAX_EXPRS * MI355Q_MAX_EXPR_NODES * sizeof(XNode) +
                         (size_t)(x_deepest - 1 + x_n) * kXJ * kBlock * 8;
    if (img + xarea <= 156 * 1024) {
      a.x_lds_off = (int32_t)img;
      a.x_below = x_deepest - 1;
      lds = img + xarea;
    }
  }
  if (st) {
    st->kernel_name = "k_proj_compact";
    st->n_launches = 1;
    st->variant = fast_ok ? 0 : a.sub_iters;
  }
  if (tiles > 0) {
    // fast member: 256 staged rows per wave
    const size_t fast_lds = (size_t)kWaves * (ps.n_targets > 3 ? 32 : 64) * 4 * (1 + ps.n_targets) * 8;
    const int per_cu = fast_ok ? std::min<int>(4, (int)((158 * 1024) / (fast_lds + 512))) : lds <= 80 * 1024 ? 2 : 1;
    int64_t grid = (int64_t)n_cus * per_cu;
    if (tune_knobs().blocks_per_cu > 0) grid = (int64_t)n_cus * tune_knobs().blocks_per_cu;
    if (grid > tiles) grid = tiles;
    if (st && st->k_start) (void)hipEventRecord(st->k_start, s);
    if (!fast_ok) {
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_proj_compact_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        (void)hipFuncSetAttribute((const void*)k_proj_compact_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        attr_set = true;
      }
      if (p.join_col >= 0 && p.join_hash_type >= 2) hipLaunchKernelGGL(k_proj_join_1n, dim3((unsigned)std::min<int64_t>(tiles, (int64_t)n_cus * 4)), dim3(kBlock), 0, s, p, a);
      else if (p.join_col >= 0) hipLaunchKernelGGL(k_proj_compact_lds<true>, dim3((unsigned)grid), dim3(kBlock), lds, s, p, a);
      else hipLaunchKernelGGL(k_proj_compact_lds<false>, dim3((unsigned)grid), dim3(kBlock), lds, s, p, a);
    } else {
      fa.n_quals = p.n_quals;
      fa.n_targets = ps.n_targets;
      fa.n_frags = fv.n_frags;
      fa.n_cols_table = ps.n_cols_table;
      fa.entry_count = ps.entry_count;
      fa.cols = fv.d_cols;
      fa.num_rows = fv.d_num_rows;
      fa.tile_start = tile_start;
      fa.n_tiles = tiles;
      fa.desc = desc;
      fa.counters = counters;
      fa.out = (int64_t*)out;
      fa.out16 = a.out16;
      // the split route: pass A and the tiles' prefix in kernels of their own (pass_rows -2 / -3: tests and tools force one route)
      const bool split = fv.n_frags < (1 << (32 - kCntBits)) && (tune_knobs().pass_rows == -3 || (tiles >= kSplitMinTiles && tune_knobs().pass_rows != -2));
      if (split) {
        fa.pre_mask = (unsigned long long*)(sp + 64 + ts_bytes + ((tiles * 8 + 63) & ~(int64_t)63) + 64);
        fa.pre_cnt = (uint32_t*)(fa.pre_mask + tiles * kBlock);
        fa.dense = fa.pre_cnt + ((tiles + 15) & ~(int64_t)15);
        fa.columnar = ps.columnar;
        // (tiles by position: a grid beyond what is resident at once would run as a second, mostly idle round)
        static int occ_mask = 0, occ_sparse = 0;
        if (!occ_mask) {
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_mask, k_proj_mask, kBlock, 0) != hipSuccess || occ_mask < 1) occ_mask = 4;
          if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_sparse, k_proj_sparse, kBlock, 0) != hipSuccess || occ_sparse < 1) occ_sparse = 4;
        }
        hipLaunchKernelGGL(k_proj_mask, dim3((unsigned)std::min<int64_t>(tiles, (int64_t)n_cus * occ_mask)), dim3(kBlock), 0, s, fa);
        hipLaunchKernelGGL(k_proj_scan_tiles, dim3(1), dim3(1024), 0, s, fa.pre_cnt, desc, tiles, counters, fa.dense);
        hipLaunchKernelGGL(k_proj_sparse, dim3((unsigned)std::min<int64_t>(tiles, (int64_t)n_cus * occ_sparse)), dim3(kBlock), 0, s, fa);
        if (st) st->variant = 16;
      }
#define MQ_PROJ_FAST(N)                                                                                                   \
  case N:                                                                                                                  \
    if (ps.columnar) hipLaunchKernelGGL((k_proj_fast<N, true>), dim3((unsigned)grid), dim3(kBlock), fast_lds, s, fa, 0);   \
    else hipLaunchKernelGGL((k_proj_fast<N, false>), dim3((unsigned)grid), dim3(kBlock), fast_lds, s, fa, 0);              \
    break
#define MQ_PROJ_FAST_XF(N)                                                                                                      \
  case N:                                                                                                                       \
    if (ps.columnar) hipLaunchKernelGGL((k_proj_fast<N, true, true>), dim3((unsigned)grid), dim3(kBlock), fast_lds, s, fa, 0);  \
    else hipLaunchKernelGGL((k_proj_fast<N, false, true>), dim3((unsigned)grid), dim3(kBlock), fast_lds, s, fa, 0);             \
    break
      if (fa.any_form) {  // (≤ 4 targets: checked above)
        switch (ps.n_targets) {
          MQ_PROJ_FAST_XF(1);
          MQ_PROJ_FAST_XF(2);
          MQ_PROJ_FAST_XF(3);
          default:
            MQ_PROJ_FAST_XF(4);
        }
      } else
      switch (ps.n_targets) {
        MQ_PROJ_FAST(1);
        MQ_PROJ_FAST(2);
        MQ_PROJ_FAST(3);
        MQ_PROJ_FAST(4);
        MQ_PROJ_FAST(5);
        MQ_PROJ_FAST(6);
        MQ_PROJ_FAST(7);
        default:
          MQ_PROJ_FAST(8);
      }
#undef MQ_PROJ_FAST
#undef MQ_PROJ_FAST_XF
    }
    if (st && st->k_stop) (void)hipEventRecord(st->k_stop, s);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const int64_t tail_quads = ps.columnar ? ps.entry_count : ps.entry_count * ps.row_quad;
  int64_t g2 = (tail_quads + kBlock * 8 - 1) / (kBlock * 8);
  g2 = g2 < 1 ? 1 : g2 > (int64_t)n_cus * 8 ? (int64_t)n_cus * 8 : g2;
  hipLaunchKernelGGL(k_proj_init_tail, dim3((unsigned)g2), dim3(kBlock), 0, s, ps, counters, (int64_t*)out);
  return hipGetLastError();
}

hipError_t launch_projection_count_live(const int64_t* keys, int64_t stride_quads, int64_t entries, unsigned long long* d_count,
                                        hipStream_t s) {
  hipError_t e = hipMemsetAsync(d_count, 0, 8, s);
  if (e != hipSuccess) return e;
  int64_t g = (entries + kBlock * 8 - 1) / (kBlock * 8);
  g = g < 1 ? 1 : g > 2048 ? 2048 : g;
  hipLaunchKernelGGL(k_proj_count_live, dim3((unsigned)g), dim3(kBlock), 0, s, keys, stride_quads, entries, d_count);
  return hipGetLastError();
}

}  // namespace mq
