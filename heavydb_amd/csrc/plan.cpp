// plan.cpp — plan-time decisions: RelAlgExecutionUnit subset -> QueryMemoryDescriptor mirror
// -> device plan.  Host C++, no device code.
//
// Restates (heavyai/heavydb):
//   GroupByAndAggregate::getColRangeInfo            GroupByAndAggregate.cpp:232-365
//   is_column_range_too_big_for_perfect_hash        GroupByAndAggregate.cpp:130-139
//   getBucketedCardinality                          GroupByAndAggregate.cpp:367-375
//   get_keyless_info                                GroupByAndAggregate.cpp:489-648
//   QueryMemoryDescriptor::init                     Descriptors/QueryMemoryDescriptor.cpp:240-446
//   pick_baseline_key_width                         Descriptors/QueryMemoryDescriptor.cpp:113-146
//   multi-column getColRangeInfo / perfect key hash GroupByAndAggregate.cpp:241-283,1546-1598
//   get_col_decoder (encodings -> type codes)       ColumnIR.cpp (get_col_decoder), DecodersImpl.h
//   ColSlotContext (slots per target)               Descriptors/ColSlotContext.cpp:35-100
//   getRowSize                                      Descriptors/QueryMemoryDescriptor.cpp:848-860
//   init_agg_val_vec / get_agg_initial_val          OutputBufferInitialization.cpp:24-84,132-289
//   get_target_info_impl                            Shared/TargetInfo.cpp:20-82
//   skip_null_val for non-grouped aggregates        TargetExprBuilder.cpp:684-690
#include "plan.h"

#include <cfloat>
#include <climits>
#include <cstring>

namespace mq {

namespace {

constexpr int64_t kMaxBufferSize = int64_t(1) << 30;  // GroupByAndAggregate.cpp:57

struct ArgInfo {
  int type = 0;
  bool nullable = false;
  bool fp = false;   // DOUBLE or FLOAT
  const mi355q_range* range = nullptr;
  bool f32 = false;  // FLOAT: 4-byte init patterns, sign-extended (byte_width 4 cases)
};

bool valid_type(int t) { return t >= MI355Q_INT8 && t <= MI355Q_FLOAT; }
bool int_type(int t) { return t >= MI355Q_INT8 && t <= MI355Q_INT64; }

constexpr int64_t kBaselineGroupbyThreshold = 1000000;  // g_baseline_groupby_threshold, Execute.cpp:113
thread_local int64_t t_twin_max_entries = 0;            // > 0 inside a PerfectTwinScope

// two's-complement add: a caller-supplied range may end at INT64_MAX
int64_t wrap_add(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }

// getBucketedCardinality (GroupByAndAggregate.cpp:367-375); callers have bounded max - min
int64_t bucketed_cardinality(const mi355q_range& r) {
  int64_t c = r.max - r.min;
  if (r.bucket > 0) c /= r.bucket;
  return c + 1 + (r.has_nulls ? 1 : 0);
}

// Initial slot value for an aggregate whose init type has `notnull`.
int64_t initial_val(int agg, const ArgInfo& a, bool notnull) {
  if (a.f32) {  // get_agg_initial_val with byte_width 4 (OutputBufferInitialization.cpp:139-236)
    switch (agg) {
      case MI355Q_SUM:
      case MI355Q_SUM_IF:
        return notnull ? (int64_t)flt_bits(0.0f) : (int64_t)kNullFloatBits;
      case MI355Q_MIN:
        return notnull ? (int64_t)flt_bits(FLT_MAX) : (int64_t)kNullFloatBits;
      case MI355Q_MAX:
        return notnull ? (int64_t)flt_bits(-FLT_MAX) : (int64_t)kNullFloatBits;
      default:
        return 0;
    }
  }
  switch (agg) {
    case MI355Q_SUM:
    case MI355Q_SUM_IF:
      if (notnull) return a.fp ? dbl_bits(0.0) : 0;
      return a.fp ? kNullDoubleBits : INT64_MIN;  // NULL of DOUBLE / of the BIGINT sum
    case MI355Q_MIN:
      if (a.fp) return notnull ? dbl_bits(DBL_MAX) : kNullDoubleBits;
      return notnull ? INT64_MAX : int_null_of(a.type);
    case MI355Q_MAX:
      if (a.fp) return notnull ? dbl_bits(-DBL_MAX) : kNullDoubleBits;
      return notnull ? INT64_MIN : int_null_of(a.type);
    default:  // AVG, COUNT, projections
      return 0;
  }
}

}  // namespace

// mi355q_col_desc -> type code (dev_common.h); < 0 = invalid combination
int col_type_code(const mi355q_col_desc& c) {
  if (!valid_type(c.type)) return -1;
  switch (c.encoding) {
    case MI355Q_ENC_NONE:
      return c.type;
    case MI355Q_ENC_FIXED:
      if (!int_type(c.type) || !int_type(c.logical_type) || c.logical_type <= c.type) return -1;
      return tc_make(c.type, MI355Q_ENC_FIXED, c.logical_type, c.nullable);
    case MI355Q_ENC_DICT:
      if (c.type == MI355Q_INT32) return MI355Q_INT32;  // 4-byte ids: plain signed int32
      if (c.type != MI355Q_INT8 && c.type != MI355Q_INT16) return -1;
      return tc_make(c.type, MI355Q_ENC_DICT, MI355Q_INT32, c.nullable);
    case MI355Q_ENC_DATE_IN_DAYS:
      if (c.type != MI355Q_INT16 && c.type != MI355Q_INT32) return -1;
      return tc_make(c.type, MI355Q_ENC_DATE_IN_DAYS, MI355Q_INT64, 1);
    default:
      return -1;
  }
}

// ---------------------------------------------------------------- projected expressions
// Type rules of the micro-op programs (what Analyzer::BinOper::normalize_simple_predicate / the analyzer's
// common-type casts leave for the code generator): both operands of + - * already have the node's type,
// casts go between any two numeric types, a column node takes the column's logical type.
int32_t lower_exprs(const mi355q_plan& p, mi355q_plan* lowered, DevExprSet* dev, bool widen_filter_bools) {
  if (p.n_exprs < 0 || p.n_exprs > MI355Q_MAX_EXPRS || p.n_cols < 0 || p.n_cols + p.n_exprs > MI355Q_MAX_COLS)
    return MI355Q_ERR_INVALID_PLAN;
  if (lowered != &p) *lowered = p;
  DevExprSet local;
  DevExprSet& ds = dev ? *dev : local;
  std::memset(&ds, 0, sizeof(ds));
  ds.n = p.n_exprs;
  ds.n_cols = p.n_cols;
  for (int k = 0; k < p.n_exprs; ++k) {
    const mi355q_expr& x = p.exprs[k];
    if (x.n_nodes < 1 || x.n_nodes > MI355Q_MAX_EXPR_NODES) return MI355Q_ERR_INVALID_PLAN;
    DevExpr& d = ds.e[k];
    d.n_nodes = x.n_nodes;
    int st_type[MI355Q_MAX_EXPR_STACK];
    bool st_null[MI355Q_MAX_EXPR_STACK];
    int sp = 0;
    for (int i = 0; i < x.n_nodes; ++i) {
      const mi355q_expr_node& n = x.nodes[i];
      DevExprNode& o = d.nodes[i];
      o.op = n.op;
      o.ilit = n.ilit;
      o.flit = n.flit;
      switch (n.op) {
        case MI355Q_EX_COL: {
          // a physical column, or the value of an EARLIER expression of the plan (column n_cols + j, j < k): the projection
          // evaluates the expressions of a row in order, so expression j's dense temporary column already holds it
          if (n.arg < 0 || n.arg >= p.n_cols + k || sp >= MI355Q_MAX_EXPR_STACK) return MI355Q_ERR_INVALID_PLAN;
          const mi355q_col_desc cd = n.arg < p.n_cols ? p.cols[n.arg] : lowered->cols[n.arg];
          const int code = col_type_code(cd);
          if (code < 0) return MI355Q_ERR_INVALID_PLAN;
          o.arg = n.arg;
          o.type = tc_logical(code);
          o.ilit = code;
          o.flags = cd.nullable ? EXF_NULLABLE : 0;
          st_type[sp] = o.type;
          st_null[sp] = cd.nullable != 0;
          ++sp;
          break;
        }
        case MI355Q_EX_LIT: {
          if (!valid_type(n.type) || sp >= MI355Q_MAX_EXPR_STACK || (n.reserved != 0 && n.reserved != 1)) return MI355Q_ERR_INVALID_PLAN;
          if (n.reserved == 1) {  // the NULL literal of the type
            o.type = n.type;
            o.flags = EXF_NULLABLE;
            o.ilit = n.type == MI355Q_DOUBLE ? kNullDoubleBits
                     : n.type == MI355Q_FLOAT ? (int64_t)(uint32_t)kNullFloatBits : plain_int_null(n.type);
            o.flit = 0.0;
            o.arg = 1;  // (the evaluator pushes ilit as it is)
            st_type[sp] = n.type;
            st_null[sp] = true;
            ++sp;
            break;
          }
          o.arg = 0;
          if (int_type(n.type) &&
              (n.ilit > (n.type == MI355Q_INT8 ? INT8_MAX : n.type == MI355Q_INT16 ? INT16_MAX
                         : n.type == MI355Q_INT32 ? (int64_t)INT32_MAX : INT64_MAX) ||
               n.ilit < plain_int_null(n.type)))
            return MI355Q_ERR_INVALID_PLAN;
          o.type = n.type;
          o.flags = 0;
          st_type[sp] = n.type;
          st_null[sp] = false;
          ++sp;
          break;
        }
        case MI355Q_EX_CAST: {
          if (!valid_type(n.type) || sp < 1) return MI355Q_ERR_INVALID_PLAN;
          o.type = n.type;
          o.arg = st_type[sp - 1];
          o.flags = st_null[sp - 1] ? (EXF_NULLABLE | EXF_LHS_NULLABLE) : 0;
          st_type[sp - 1] = n.type;
          break;
        }
        case MI355Q_EX_ADD:
        case MI355Q_EX_SUB:
        case MI355Q_EX_MUL:
        case MI355Q_EX_DIV:
        case MI355Q_EX_MOD: {
          if (!valid_type(n.type) || sp < 2) return MI355Q_ERR_INVALID_PLAN;
          if (n.op == MI355Q_EX_MOD && !int_type(n.type)) return MI355Q_ERR_INVALID_PLAN;
          if (st_type[sp - 1] != n.type || st_type[sp - 2] != n.type) return MI355Q_ERR_INVALID_PLAN;
          o.type = n.type;
          o.flags = (st_null[sp - 2] ? EXF_LHS_NULLABLE : 0) | (st_null[sp - 1] ? EXF_RHS_NULLABLE : 0);
          const bool nul = st_null[sp - 2] || st_null[sp - 1];
          if (nul) o.flags |= EXF_NULLABLE;
          --sp;
          st_null[sp - 1] = nul;
          break;
        }
        case MI355Q_EX_EQ:
        case MI355Q_EX_NE:
        case MI355Q_EX_LT:
        case MI355Q_EX_LE:
        case MI355Q_EX_GT:
        case MI355Q_EX_GE: {
          if (n.type != MI355Q_INT8 || sp < 2 || st_type[sp - 1] != st_type[sp - 2]) return MI355Q_ERR_INVALID_PLAN;
          o.type = MI355Q_INT8;
          o.arg = st_type[sp - 1];  // the operands' type
          o.flags = (st_null[sp - 2] ? EXF_LHS_NULLABLE : 0) | (st_null[sp - 1] ? EXF_RHS_NULLABLE : 0);
          const bool nul = st_null[sp - 2] || st_null[sp - 1];
          if (nul) o.flags |= EXF_NULLABLE;
          --sp;
          st_type[sp - 1] = MI355Q_INT8;
          st_null[sp - 1] = nul;
          break;
        }
        case MI355Q_EX_CASE: {  // stack: ELSE, THEN, cond
          if (!valid_type(n.type) || sp < 3 || st_type[sp - 1] != MI355Q_INT8 || st_type[sp - 2] != n.type ||
              st_type[sp - 3] != n.type)
            return MI355Q_ERR_INVALID_PLAN;
          o.type = n.type;
          const bool nul = st_null[sp - 2] || st_null[sp - 3];
          o.flags = nul ? EXF_NULLABLE : 0;
          sp -= 2;
          st_null[sp - 1] = nul;
          break;
        }
        case MI355Q_EX_NOT: {
          if (n.type != MI355Q_INT8 || sp < 1 || st_type[sp - 1] != MI355Q_INT8) return MI355Q_ERR_INVALID_PLAN;
          o.type = MI355Q_INT8;
          o.flags = st_null[sp - 1] ? (EXF_NULLABLE | EXF_LHS_NULLABLE) : 0;
          break;
        }
        case MI355Q_EX_AND:
        case MI355Q_EX_OR: {
          if (n.type != MI355Q_INT8 || sp < 2 || st_type[sp - 1] != MI355Q_INT8 || st_type[sp - 2] != MI355Q_INT8 ||
              (n.reserved != 0 && n.reserved != 1))
            return MI355Q_ERR_INVALID_PLAN;
          o.type = MI355Q_INT8;
          o.flags = (st_null[sp - 2] ? EXF_LHS_NULLABLE : 0) | (st_null[sp - 1] ? EXF_RHS_NULLABLE : 0) |
                    (n.reserved == 1 ? EXF_SHORT_CIRCUIT : 0);
          const bool nul = st_null[sp - 2] || st_null[sp - 1];
          if (nul) o.flags |= EXF_NULLABLE;
          --sp;
          st_null[sp - 1] = nul;
          break;
        }
        case MI355Q_EX_IS_NULL: {
          if (n.type != MI355Q_INT8 || sp < 1) return MI355Q_ERR_INVALID_PLAN;
          o.type = MI355Q_INT8;
          o.arg = st_type[sp - 1];  // the operand's type
          o.flags = st_null[sp - 1] ? EXF_LHS_NULLABLE : 0;
          st_type[sp - 1] = MI355Q_INT8;
          st_null[sp - 1] = false;
          break;
        }
        case MI355Q_EX_UMINUS: {
          if (!valid_type(n.type) || sp < 1 || st_type[sp - 1] != n.type) return MI355Q_ERR_INVALID_PLAN;
          o.type = n.type;
          o.flags = st_null[sp - 1] ? (EXF_NULLABLE | EXF_LHS_NULLABLE) : 0;
          break;
        }
        default:
          return MI355Q_ERR_UNSUPPORTED;
      }
    }
    if (sp != 1) return MI355Q_ERR_INVALID_PLAN;
    d.type = st_type[0];
    d.store_type = d.type;
    d.nullable = st_null[0] ? 1 : 0;
    const int c = p.n_cols + k;
    lowered->cols[c] = mi355q_col_desc{d.type, d.nullable, MI355Q_ENC_NONE, 0};
    lowered->col_ranges[c] = x.range;
  }
  if (widen_filter_bools) {
    for (int k = 0; k < p.n_exprs; ++k) {
      const int c = p.n_cols + k;
      if (ds.e[k].type != MI355Q_INT8) continue;
      bool filter_only = false, other = false;
      for (int i = 0; i < p.n_quals && i < MI355Q_MAX_QUALS; ++i) filter_only = filter_only || p.quals[i].col == c;
      for (int i = 0; i < p.n_group_cols && i < MI355Q_MAX_GROUP_COLS; ++i) other = other || p.group_cols[i] == c;
      for (int i = 0; i < p.n_targets && i < MI355Q_MAX_TARGETS; ++i)
        other = other || (p.targets[i].table == 0 && p.targets[i].col == c) || p.targets[i].cond.col == c;
      other = other || p.join_outer_col == c;
      for (int i = 1; i < p.n_join_cols && i < MI355Q_MAX_GROUP_COLS; ++i) other = other || p.join_outer_cols[i] == c;
      for (int j = k + 1; j < p.n_exprs; ++j)
        for (int i = 0; i < p.exprs[j].n_nodes; ++i) other = other || (p.exprs[j].nodes[i].op == MI355Q_EX_COL && p.exprs[j].nodes[i].arg == c);
      if (!filter_only || other) continue;
      ds.e[k].store_type = MI355Q_INT32;
      lowered->cols[c].type = MI355Q_INT32;
    }
  }
  lowered->n_cols = p.n_cols + p.n_exprs;
  lowered->n_exprs = 0;
  return MI355Q_OK;
}

// Which expressions are evaluated for EVERY row: the ones a qual reads, and — transitively — the earlier expressions those
// read (the reference evaluates a filter before anything else of the row, Executor::compileBody).  The checks of the other
// expressions count only for rows that pass the quals and, under an INNER join, find a match.
uint32_t expr_qual_mask(const mi355q_plan& p) {
  uint32_t mask = 0;
  for (int i = 0; i < p.n_quals && i < MI355Q_MAX_QUALS; ++i)
    if (p.quals[i].col >= p.n_cols && p.quals[i].col < p.n_cols + MI355Q_MAX_EXPRS) mask |= 1u << (p.quals[i].col - p.n_cols);
  for (int k = (p.n_exprs < MI355Q_MAX_EXPRS ? p.n_exprs : MI355Q_MAX_EXPRS) - 1; k >= 0; --k) {
    if (!((mask >> k) & 1u)) continue;
    const mi355q_expr& x = p.exprs[k];
    for (int i = 0; i < x.n_nodes && i < MI355Q_MAX_EXPR_NODES; ++i)
      if (x.nodes[i].op == MI355Q_EX_COL && x.nodes[i].arg >= p.n_cols && x.nodes[i].arg < p.n_cols + k)
        mask |= 1u << (x.nodes[i].arg - p.n_cols);
  }
  return mask;
}

int32_t resolve_targets(const mi355q_plan& p, bool grouped, ResolvedTarget* out) {
  for (int i = 0; i < p.n_targets; ++i) {
    const mi355q_target& t = p.targets[i];
    ResolvedTarget& r = out[i];
    r = ResolvedTarget{};
    r.agg = t.agg;
    r.col = t.col;
    r.table = t.table;
    switch (t.agg) {
      case MI355Q_AVG:
      case MI355Q_MIN:
      case MI355Q_MAX:
      case MI355Q_SUM:
        if (t.col < 0) return MI355Q_ERR_INVALID_PLAN;
        break;
      case MI355Q_COUNT:
        break;
      case MI355Q_SUM_IF:
        if (t.col < 0) return MI355Q_ERR_INVALID_PLAN;
        [[fallthrough]];
      case MI355Q_COUNT_IF:
        if (t.cond.col < 0 || t.cond.col >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
        if (t.agg == MI355Q_COUNT_IF) r.col = -1;  // the condition is the argument
        // `x IS [NOT] NULL` is itself never NULL (a NOT NULL BOOLEAN, Analyzer UOper kISNULL)
        r.cond_nullable = p.cols[t.cond.col].nullable != 0 && t.cond.op != MI355Q_IS_NULL && t.cond.op != MI355Q_IS_NOT_NULL;
        break;
      case MI355Q_PROJECT:  // a Projection step's target: the value of an outer column / expression, no aggregate
        if (grouped || t.col < 0) return MI355Q_ERR_INVALID_PLAN;
        if (t.table != 0 && p.join_outer_col < 0) return MI355Q_ERR_INVALID_PLAN;  // (an inner column needs the join)
        break;
      case MI355Q_PROJECT_KEY:
        if (!grouped) return MI355Q_ERR_INVALID_PLAN;
        r.key_idx = t.col < 0 ? 0 : t.col;
        if (r.key_idx >= p.n_group_cols) return MI355Q_ERR_INVALID_PLAN;
        r.col = p.group_cols[r.key_idx];
        r.table = 0;
        break;
      default:
        return MI355Q_ERR_UNSUPPORTED;
    }
    if (r.col >= 0) {
      const int ncols = r.table ? p.n_inner_cols : p.n_cols;
      if (r.col >= ncols || (r.table && p.join_outer_col < 0)) return MI355Q_ERR_INVALID_PLAN;
      const mi355q_col_desc& cd = r.table ? p.inner_cols[r.col] : p.cols[r.col];
      r.arg_type = col_type_code(cd);
      if (r.arg_type < 0) return MI355Q_ERR_INVALID_PLAN;
      // inner columns of an outer join are nullable whatever their declaration
      r.arg_nullable = cd.nullable != 0 || (r.table && p.join_kind == MI355Q_JOIN_LEFT);
      r.arg_fp = type_is_fp(cd.type);
      r.arg_f32 = type_is_f32(cd.type);
      r.range = r.table ? &p.inner_col_ranges[r.col] : &p.col_ranges[r.col];
    }
    const bool is_agg = t.agg != MI355Q_PROJECT_KEY && t.agg != MI355Q_PROJECT;
    // constrained_not_null (OutputBufferInitialization.cpp:301-324): `arg IS NOT NULL` among the quals
    r.constrained = false;
    if (is_agg && r.col >= 0 && r.table == 0) {
      for (int k = 0; k < p.n_quals; ++k)
        if (p.quals[k].op == MI355Q_IS_NOT_NULL && p.quals[k].col == r.col) r.constrained = true;  // (a top-level conjunct: group 0)
    }
    r.skip_null = is_agg && r.col >= 0 && ((r.arg_nullable && !r.constrained) || !grouped);
    // COUNT_IF's argument is the condition itself (TargetInfo.cpp:60-82)
    if (t.agg == MI355Q_COUNT_IF) r.skip_null = r.cond_nullable || !grouped;
    r.n_slots = t.agg == MI355Q_AVG ? 2 : 1;
  }
  return MI355Q_OK;
}

namespace {

// Can one aggregate's slot tell a touched entry from an untouched one?  If so the layout
// may drop the key column ("keyless"); returns the SLOT index of that aggregate.
void keyless_decision(const mi355q_plan& p, const ResolvedTarget* ts, bool* keyless_out,
                      int* slot_index_out) {
  bool keyless = true, found = false;
  int index = 0;
  for (int i = 0; i < p.n_targets; ++i) {
    const ResolvedTarget& t = ts[i];
    if (!found && t.agg != MI355Q_PROJECT_KEY) {
      ArgInfo a{t.arg_type, t.arg_nullable, t.arg_fp || t.arg_f32, t.range, t.arg_f32};
      // a column from the inner side of an outer join can be NULL whatever its metadata says
      // (getExpressionRange sets hasNulls for outer-join projections, ExpressionRange.cpp)
      mi355q_range outer_rng;
      if (t.table && p.join_kind == MI355Q_JOIN_LEFT && a.range) {
        outer_rng = *a.range;
        outer_rng.has_nulls = 1;
        a.range = &outer_rng;
      }
      const bool rng_ok = a.range && a.range->valid;
      switch (t.agg) {
        case MI355Q_AVG:
          ++index;
          if (t.arg_nullable && (!rng_ok || a.range->has_nulls)) break;
          found = true;
          break;
        case MI355Q_COUNT:
          if (t.col >= 0 && t.arg_nullable && (!rng_ok || a.range->has_nulls)) break;
          found = true;
          break;
        case MI355Q_SUM:
          if (t.arg_nullable && !t.constrained) {  // GroupByAndAggregate.cpp:531
            found = rng_ok && !a.range->has_nulls;
          } else if (rng_ok) {
            found = a.fp ? (a.range->fp_max < 0 || a.range->fp_min > 0)
                         : (a.range->max < 0 || a.range->min > 0);
          }
          break;
        case MI355Q_MIN:
          if (rng_ok) {
            const int64_t init = initial_val(MI355Q_MIN, a, !t.arg_nullable);
            found = a.fp ? a.range->fp_max < bits_dbl(init) : a.range->max < init;
          }
          break;
        case MI355Q_MAX:
          if (rng_ok && !a.range->has_nulls) {
            const int64_t init = initial_val(MI355Q_MAX, a, !t.arg_nullable);
            found = a.fp ? a.range->fp_min > bits_dbl(init) : a.range->min > init;
          }
          break;
        default:
          keyless = false;
      }
    }
    if (!keyless) break;
    if (!found) ++index;
  }
  *keyless_out = keyless && found;
  *slot_index_out = index;
}

}  // namespace

// QueryDescriptionType::Projection (QueryMemoryDescriptor::init, Descriptors/QueryMemoryDescriptor.cpp:394-410): one
// "group column" — the 8-byte row offset get_scan_output_slot writes (GroupByRuntime.cpp:242-255; groupby_exprs of a
// projection is one nullptr, get_col_byte_widths gives it sizeof(int64_t)) — and one slot per target from
// ColSlotContext(target_exprs, {}) (ColSlotContext.cpp:35-100: the target type's width), whose padded sizes the
// constructor then sets to 8 (setAllUnsetSlotsPaddedSize, :507) — or, for a columnar projection, to the logical widths
// (isLogicalSizedColumnsAllowed :1129-1135).  entry_count = scan_limit, else max_groups_buffer_entry_count.
int32_t qmd_init_projection(const mi355q_plan& p, const ResolvedTarget* ts, mi355q_qmd* q) {
  // (a join: every joined row is one entry — the one-to-one tables are executed, api_projection.cpp)
  if (p.n_group_cols != 0) return MI355Q_ERR_INVALID_PLAN;
  if (p.scan_limit < 0) return MI355Q_ERR_INVALID_PLAN;
  if (p.output_columnar_hint == MI355Q_OUTPUT_ROWWISE_COLUMNAR_DECISIONS) return MI355Q_ERR_INVALID_PLAN;
  q->desc_type = MI355Q_PROJECTION;
  q->n_targets = p.n_targets;
  q->group_col_count = 1;
  q->idx_target_as_key = -1;
  q->key_width = 8;
  q->key_bytes = 8;
  q->entry_count = p.scan_limit > 0 ? p.scan_limit : (p.max_groups_buffer_entry_guess > 0 ? p.max_groups_buffer_entry_guess : 16384);
  // pos / output_buffer_entry_count are uint32 in get_scan_output_slot, total_matched an int32
  if (q->entry_count > (int64_t)INT32_MAX) return MI355Q_ERR_UNSUPPORTED;
  q->output_columnar = p.output_columnar_hint == MI355Q_OUTPUT_COLUMNAR;
  q->slot_count = p.n_targets;
  q->slot_width = 8;
  for (int i = 0; i < p.n_targets; ++i) {
    const ResolvedTarget& t = ts[i];
    const int logical = tc_logical(t.arg_type);
    q->target_agg[i] = MI355Q_PROJECT;
    q->target_slot[i] = i;
    q->target_key_idx[i] = 0;
    q->target_skip_null[i] = 0;
    q->target_is_fp[i] = t.arg_fp || t.arg_f32;
    // row-wise: a FLOAT value is widened to double for its 8-byte slot (castToTypeIn(target_lv, 64) + agg_id_double,
    // TargetExprBuilder.cpp:485-560; run_query_external fills the slot the same way, ExternalExecutor.cpp:470-479);
    // columnar: the 4-byte float itself
    q->target_arg_is_fp[i] = t.arg_fp || (t.arg_f32 && !q->output_columnar);
    q->target_arg_is_f32[i] = t.arg_f32 && q->output_columnar;
    q->slot_bytes[i] = q->output_columnar ? plain_width(logical) : 8;
    q->init_vals[i] = 0;  // init_agg_val_vec: a non-aggregate target starts at 0 (OutputBufferInitialization.cpp:40-47)
    if (!t.arg_nullable) q->target_null[i] = kEmptyKey64;  // (ResultSet::isNull tests the type first: never NULL)
    else if (t.arg_fp) q->target_null[i] = kNullDoubleBits;
    else if (t.arg_f32) q->target_null[i] = q->output_columnar ? (int64_t)kNullFloatBits : dbl_bits((double)bits_flt(kNullFloatBits));
    else q->target_null[i] = int_null_of(t.arg_type);
  }
  q->row_size = 8 + 8 * q->slot_count;
  return MI355Q_OK;
}

int32_t qmd_init(const mi355q_plan& p, mi355q_qmd* q) {
  std::memset(q, 0, sizeof(*q));
  if (p.abi_version != MI355Q_ABI_VERSION) return MI355Q_ERR_INVALID_PLAN;
  if (p.n_exprs != 0) {  // the layout of a plan with expressions is the layout of its lowered form
    mi355q_plan lp;
    if (int32_t e = lower_exprs(p, &lp, nullptr)) return e;
    if (lp.join_outer_col >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;  // expressions are not join keys
    for (int i = 1; i < lp.n_join_cols && i < MI355Q_MAX_GROUP_COLS; ++i)
      if (lp.join_outer_cols[i] >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
    return qmd_init(lp, q);
  }
  if (p.n_cols < 0 || p.n_cols > MI355Q_MAX_COLS || p.n_inner_cols < 0 ||
      p.n_inner_cols > MI355Q_MAX_COLS || p.n_quals < 0 || p.n_quals > MI355Q_MAX_QUALS ||
      p.n_targets < 1 || p.n_targets > MI355Q_MAX_TARGETS || p.n_group_cols < 0 ||
      p.n_group_cols > MI355Q_MAX_GROUP_COLS || p.output_columnar_hint < MI355Q_OUTPUT_ROWWISE ||
      p.output_columnar_hint > MI355Q_OUTPUT_ROWWISE_COLUMNAR_DECISIONS) {
    return MI355Q_ERR_INVALID_PLAN;
  }
  for (int i = 0; i < p.n_cols; ++i) {
    if (col_type_code(p.cols[i]) < 0) return MI355Q_ERR_INVALID_PLAN;
  }
  for (int i = 0; i < p.n_inner_cols; ++i) {
    if (col_type_code(p.inner_cols[i]) < 0) return MI355Q_ERR_INVALID_PLAN;
  }
  for (int i = 0; i < p.n_quals; ++i) {
    if (p.quals[i].col < 0 || p.quals[i].col >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
    // the OR-group bits are stated once, here: every later reader of quals[].op may rely on MI355Q_QUAL_OP() /
    // MI355Q_QUAL_OR_GROUP() being in range (build_dev_plan repeats the check for plans that reach it another way)
    const int32_t op = p.quals[i].op;
    if (op < 0 || (op >> 16) != 0 || MI355Q_QUAL_OR_GROUP(op) > MI355Q_MAX_OR_GROUPS) return MI355Q_ERR_INVALID_PLAN;
  }
  const bool grouped = p.n_group_cols >= 1;
  ResolvedTarget ts[MI355Q_MAX_TARGETS];
  if (int32_t e = resolve_targets(p, grouped, ts)) return e;

  int n_project = 0;
  for (int i = 0; i < p.n_targets; ++i) n_project += p.targets[i].agg == MI355Q_PROJECT;
  if (n_project) {
    if (n_project != p.n_targets) return MI355Q_ERR_INVALID_PLAN;  // (a target list is all aggregates or none)
    return qmd_init_projection(p, ts, q);
  }
  if (p.scan_limit != 0) return MI355Q_ERR_INVALID_PLAN;  // (scan_limit exists for projections only: RelAlgExecutionUnit.h:178)

  q->n_targets = p.n_targets;
  q->group_col_count = p.n_group_cols;
  q->idx_target_as_key = -1;
  q->key_width = 8;
  q->entry_count = 1;
  q->desc_type = MI355Q_NON_GROUPED_AGGREGATE;

  for (int g = 0; g < p.n_group_cols; ++g) {
    const int gc = p.group_cols[g];
    if (gc < 0 || gc >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
  }
  // floating-point group keys: always the baseline layout (getColRangeInfo's Float / Double cases,
  // GroupByAndAggregate.cpp:199-207), 8-byte components (pick_baseline_key_component_width: "no
  // compaction for floating point yet"); the key is the bit pattern of the value cast to double
  // (groupByColumnCodegen: castToTypeIn(group_key, 64) + bitcast, IRCodegen.cpp:1505-1507)
  bool fp_key = false;
  for (int g = 0; g < p.n_group_cols; ++g)
    fp_key = fp_key || type_is_fp(p.cols[p.group_cols[g]].type) || type_is_f32(p.cols[p.group_cols[g]].type);
  const int64_t baseline_entries =
      p.max_groups_buffer_entry_guess > 0 ? p.max_groups_buffer_entry_guess : 16384;

  if (p.n_group_cols == 1) {
    const int gc = p.group_cols[0];
    const mi355q_range& r = p.col_ranges[gc];
    bool use_baseline = fp_key || !r.valid || r.min > r.max;
    if (!use_baseline) {
      const int64_t col_count = p.n_group_cols + p.n_targets;
      const int64_t max_entries = kMaxBufferSize / (col_count * (int64_t)sizeof(int64_t));
      const __int128 span = (__int128)r.max - (__int128)r.min;
      // a bucketed range stays on the perfect hash (":344 is_baseline_candidate && !bucket"); so does a
      // dictionary-encoded string key when the step has no filters (:312-343: "we are better off attempting
      // perfect hash ... and failing later due to excessive memory use"; with filters and no cardinality
      // estimate — none reaches this seam — the range that is too big takes the baseline layout)
      const bool too_big = span >= (__int128)max_entries;
      const bool dict_key = p.cols[gc].encoding == MI355Q_ENC_DICT && !(r.bucket > 0);
      use_baseline = dict_key ? (too_big && p.n_quals > 0) : (too_big && !(r.bucket > 0));
      if (!use_baseline && span / (r.bucket > 0 ? r.bucket : 1) >= (__int128)INT32_MAX)
        return MI355Q_ERR_UNSUPPORTED;
    }
    if (use_baseline) {
      q->desc_type = MI355Q_GROUP_BY_BASELINE_HASH;
    } else {
      q->desc_type = MI355Q_GROUP_BY_PERFECT_HASH;
      q->min_val = r.min;
      q->max_val = r.max;
      q->bucket = r.bucket > 0 ? r.bucket : 0;
      q->has_nulls = r.has_nulls != 0;
      const int64_t card = bucketed_cardinality(r);
      q->entry_count = card > 1 ? card : 1;
      q->group_min[0] = r.min;
      q->group_card[0] = card;
      q->group_bucket[0] = q->bucket;
      q->group_null_key[0] = wrap_add(r.max, q->bucket ? q->bucket : 1);
      q->group_has_nulls[0] = r.has_nulls != 0;
    }
  } else if (p.n_group_cols > 1) {
    // getColRangeInfo, groupby_exprs.size() != 1: perfect hash iff every column has an
    // integer range and the product of the bucketed cardinalities is within
    // g_baseline_groupby_threshold; zero / overflow -> baseline
    bool perfect = !fp_key;
    __int128 card = 1;
    for (int g = 0; g < p.n_group_cols && perfect; ++g) {
      const mi355q_range& r = p.col_ranges[p.group_cols[g]];
      if (!r.valid || r.min > r.max) {
        perfect = false;
        break;
      }
      const __int128 c = ((__int128)r.max - (__int128)r.min) / (r.bucket > 0 ? r.bucket : 1) + 1 +
                         (r.has_nulls ? 1 : 0);
      if (c > (__int128)INT64_MAX) {
        perfect = false;
        break;
      }
      card *= c;
      if (card > (__int128)INT64_MAX) perfect = false;
    }
    const int64_t threshold = t_twin_max_entries > 0 ? t_twin_max_entries : kBaselineGroupbyThreshold;
    if (perfect && (card == 0 || card > (__int128)threshold)) perfect = false;
    if (perfect) {
      q->desc_type = MI355Q_GROUP_BY_PERFECT_HASH;
      q->entry_count = (int64_t)card;
      q->min_val = 0;
      q->max_val = (int64_t)card;  // "col range info max contains the expected cardinality"
      for (int g = 0; g < p.n_group_cols; ++g) {
        const mi355q_range& r = p.col_ranges[p.group_cols[g]];
        q->group_min[g] = r.min;
        q->group_card[g] = bucketed_cardinality(r);
        q->group_bucket[g] = r.bucket > 0 ? r.bucket : 0;
        q->group_null_key[g] = wrap_add(r.max, r.bucket > 0 ? r.bucket : 1);
        q->group_has_nulls[g] = r.has_nulls != 0;
        if (r.has_nulls) q->has_nulls = 1;
      }
    } else {
      q->desc_type = MI355Q_GROUP_BY_BASELINE_HASH;
    }
  }
  if (q->desc_type == MI355Q_GROUP_BY_PERFECT_HASH) {
    bool keyless = false;
    int key_slot = 0;
    keyless_decision(p, ts, &keyless, &key_slot);
    if (p.n_group_cols == 1 && q->bucket) keyless = false;  // "!col_range_info.bucket"
    q->keyless = keyless;
    q->idx_target_as_key = key_slot;
  } else if (q->desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
    q->entry_count = baseline_entries;
    if (q->entry_count > (int64_t)UINT32_MAX) return MI355Q_ERR_UNSUPPORTED;  // h is uint32
    // pick_baseline_key_width: 4 only if every component's range is a valid int32 range;
    // "output_columnar ? 8 : pick_baseline_key_width(...)" (QueryMemoryDescriptor.cpp:386-388)
    int kw = (p.output_columnar_hint || fp_key) ? 8 : 4;
    for (int g = 0; g < p.n_group_cols && kw == 4; ++g) {
      const int gc = p.group_cols[g];
      const mi355q_range& r = p.col_ranges[gc];
      const int logical_w = plain_width(tc_logical(col_type_code(p.cols[gc])));
      int w = 8;
      if (r.valid && !(logical_w == 8 && r.has_nulls) && r.min > (int64_t)INT32_MIN &&
          r.max < (int64_t)kEmptyKey32 - 1) {
        w = 4;
      }
      if (w > kw) kw = w;
    }
    q->key_width = kw;
  }

  int slot = 0;
  for (int i = 0; i < p.n_targets; ++i) {
    const ResolvedTarget& t = ts[i];
    ArgInfo a{t.arg_type, t.arg_nullable, t.arg_fp || t.arg_f32, t.range, t.arg_f32};
    q->target_agg[i] = t.agg;
    q->target_skip_null[i] = t.skip_null;
    q->target_key_idx[i] = t.key_idx;
    q->target_arg_is_fp[i] = t.arg_fp && t.agg != MI355Q_COUNT;
    q->target_arg_is_f32[i] = t.arg_f32 && t.agg != MI355Q_COUNT;
    q->target_is_fp[i] = t.agg == MI355Q_AVG || ((t.arg_fp || t.arg_f32) && t.agg != MI355Q_COUNT);
    const bool key_in_row =
        t.agg == MI355Q_PROJECT_KEY && q->desc_type == MI355Q_GROUP_BY_BASELINE_HASH;
    if (key_in_row) {
      q->target_slot[i] = -1;  // target_groupby_indices: read from the key column
    } else {
      if (slot + t.n_slots > MI355Q_MAX_SLOTS) return MI355Q_ERR_INVALID_PLAN;
      q->target_slot[i] = slot;
      const bool init_notnull = grouped ? (!t.arg_nullable || t.constrained) : false;
      q->init_vals[slot] = initial_val(t.agg, a, init_notnull);
      if (t.agg == MI355Q_AVG) q->init_vals[slot + 1] = 0;
      slot += t.n_slots;
    }
    switch (t.agg) {
      case MI355Q_AVG:
        q->target_null[i] = kNullDoubleBits;
        break;
      case MI355Q_SUM:
      case MI355Q_SUM_IF:
        q->target_null[i] = t.arg_f32 ? (int64_t)kNullFloatBits : t.arg_fp ? kNullDoubleBits : INT64_MIN;
        break;
      case MI355Q_COUNT:
      case MI355Q_COUNT_IF:
        q->target_null[i] = p.bigint_count ? INT64_MIN : (int64_t)INT32_MIN;
        break;
      default:
        q->target_null[i] = t.arg_f32 ? (int64_t)kNullFloatBits
                                      : t.arg_fp ? kNullDoubleBits : int_null_of(t.arg_type);
    }
    // ResultSet::isNull looks at the TYPE first: a projected key of a NOT NULL column is never NULL,
    // even when it holds the inline NULL pattern (ExecuteTest.cpp:4911 expects -2147483648 back from
    // an `int not null` column).  EMPTY_KEY_64 can never be a key, so it stands for "no NULL pattern".
    // a FLOAT key sits in its key column as the double it was widened to: so does its NULL (FLT_MIN)
    if (t.agg == MI355Q_PROJECT_KEY && t.arg_f32) q->target_null[i] = dbl_bits((double)bits_flt((int32_t)kNullFloatBits));
    if (t.agg == MI355Q_PROJECT_KEY && !t.arg_nullable) q->target_null[i] = kEmptyKey64;
  }
  q->slot_count = slot;
  q->key_bytes = (grouped && !q->keyless) ? ((q->group_col_count * q->key_width + 7) & ~7) : 0;
  // pick_target_compact_width (QueryMemoryDescriptor.cpp:748-840): one group column, only
  // COUNT(*) and projections of keys of at most 4 bytes, <= UINT32_MAX input rows -> 4-byte slots.
  // A baseline-hash step rebuilds its slot context from the target list (:382-384) after that width was
  // applied (:263-269), so its slots come out unset and the constructor pads them to 8 (:507): the
  // narrowing survives only for perfect hash (ColSlotContext run on the reference's own code,
  // tests/golden/ref_layout_vectors.json).
  bool compact = !p.bigint_count && p.n_group_cols == 1 && q->desc_type != MI355Q_GROUP_BY_BASELINE_HASH &&
                 (uint64_t)(p.num_tuples < 0 ? 0 : p.num_tuples) <= (uint64_t)UINT32_MAX;
  for (int i = 0; i < p.n_targets && compact; ++i) {
    const ResolvedTarget& t = ts[i];
    if (t.agg == MI355Q_COUNT && t.col < 0) continue;
    if (t.agg == MI355Q_PROJECT_KEY && plain_width(tc_logical(t.arg_type)) <= 4 && !t.arg_f32) continue;
    compact = false;
  }
  q->slot_width = compact ? 4 : 8;
  for (int j = 0; j < q->slot_count; ++j) q->slot_bytes[j] = q->slot_width;
  q->row_size = q->key_bytes + ((q->slot_width * q->slot_count + 7) & ~7);
  if (q->row_size <= 0) return MI355Q_ERR_INVALID_PLAN;
  q->output_columnar = p.output_columnar_hint == MI355Q_OUTPUT_COLUMNAR;
  // With a columnar keyless single-column perfect hash the reference still calls
  // get_columnar_group_bin_offset (GroupByAndAggregate.cpp:1425-1430, GroupByRuntime.cpp:228-239),
  // which reads the FIRST SLOT's column as if it were the key column and overwrites an entry equal
  // to EMPTY_KEY_64 with the key.  A first slot that starts at that value (MIN over int64) is refused.
  if (q->output_columnar && q->keyless && p.n_group_cols == 1 && q->slot_width == 8 && q->slot_count > 0 &&
      q->init_vals[0] == kEmptyKey64) {
    // ... restated for the deterministic case: an unbucketed key (every row of a group writes the same key) and a
    // plain MIN in that slot, where "key first, then the aggregate" is MIN(key, values) however the rows are dealt
    // to kernels (DevPlan::col0_key_quirk).  A bucketed key would leave the first row's key there: refused.
    bool min_first = false;
    for (int i = 0; i < p.n_targets; ++i)
      if (q->target_slot[i] == 0 && q->target_agg[i] == MI355Q_MIN && !q->target_skip_null[i] && !q->target_arg_is_fp[i])
        min_first = true;
    if (q->bucket > 0 || !min_first) return MI355Q_ERR_UNSUPPORTED;
  }
  return MI355Q_OK;
}

// getBufferSizeBytes (QueryMemoryDescriptor.cpp:1084-1111); columnar: 8 bytes per group column and
// entry + getTotalBytesOfColumnarBuffers (every slot column align_to_int64(width * entry_count))
PerfectTwinScope::PerfectTwinScope(int64_t max_entries) : saved(t_twin_max_entries) { t_twin_max_entries = max_entries; }
PerfectTwinScope::~PerfectTwinScope() { t_twin_max_entries = saved; }

int64_t qmd_buffer_bytes(const mi355q_qmd& q) {
  if (!q.output_columnar) return q.entry_count * (int64_t)q.row_size;
  return qmd_slot_col_offset(q, q.slot_count);
}

int64_t qmd_group_col_offset(const mi355q_qmd& q, int g) {
  if (!q.output_columnar || q.keyless || g < 0 || g >= q.group_col_count) return -1;
  return (int64_t)g * 8 * q.entry_count;
}

// s == slot_count gives the end of the last column (= the buffer size)
int64_t qmd_slot_col_offset(const mi355q_qmd& q, int s) {
  if (!q.output_columnar || s < 0 || s > q.slot_count) return -1;
  const int64_t keys = q.keyless ? 0 : (int64_t)q.group_col_count * 8 * q.entry_count;
  if (q.desc_type == MI355Q_PROJECTION) {  // logical-sized slot columns (getColOffInBytes, QueryMemoryDescriptor.cpp:906-929)
    int64_t off = keys;
    for (int j = 0; j < s; ++j) off += ((int64_t)q.slot_bytes[j] * q.entry_count + 7) & ~(int64_t)7;
    return off;
  }
  const int64_t col = ((int64_t)q.slot_width * q.entry_count + 7) & ~(int64_t)7;
  return keys + (int64_t)s * col;
}

// fill_empty_key (ResultSet.cpp) + initColumnsPerRow (QueryMemoryInitializer.cpp:617-698):
// every key component EMPTY — with 4-byte components an odd count leaves a zero padding word —
// then the slots' init values.
void row_init_image(const mi355q_qmd& q, int64_t* quad) {
  const int kq = q.key_bytes / 8;
  for (int i = 0; i < kq; ++i) quad[i] = kEmptyKey64;
  if (q.key_width == 4) {
    int32_t* k32 = (int32_t*)quad;
    for (int i = 0; i < 2 * kq; ++i) k32[i] = i < q.group_col_count ? kEmptyKey32 : 0;
  }
  if (q.slot_width == 4) {
    int32_t* s32 = (int32_t*)(quad + kq);
    const int n32 = (q.row_size - q.key_bytes) / 4;
    for (int s = 0; s < n32; ++s) s32[s] = s < q.slot_count ? (int32_t)q.init_vals[s] : 0;
    return;
  }
  for (int s = 0; s < q.slot_count; ++s) quad[kq + s] = q.init_vals[s];
}

void layout_from_qmd(const mi355q_qmd& q, DevPlan* d) {
  d->n_targets = q.n_targets;
  for (int i = 0; i < q.n_targets; ++i) {
    DevTarget& t = d->targets[i];
    t.agg = q.target_agg[i];
    t.skip_null = q.target_skip_null[i];
    t.slot = q.target_slot[i];
    t.arg_fp = q.target_arg_is_fp[i];
    t.arg_f32 = q.target_arg_is_f32[i];
    t.key_idx = q.target_key_idx[i];
  }
  d->slot_count = q.slot_count;
  d->desc_type = q.desc_type;
  d->keyless = q.keyless;
  d->key_width = q.key_width;
  d->row_quad = q.row_size / 8;
  d->key_quad = q.key_bytes / 8;
  d->entry_count = q.entry_count;
  d->min_val = q.min_val;
  d->max_val = q.max_val;
  d->n_group = q.group_col_count;
  d->slot_width = q.slot_width ? q.slot_width : 8;
  int64_t mul = 1;
  for (int g = 0; g < q.group_col_count && g < MI355Q_MAX_GROUP_COLS; ++g) {
    d->group_min[g] = q.group_min[g];
    d->group_card[g] = q.group_card[g];
    d->group_null_key[g] = q.group_null_key[g];
    d->group_bucket[g] = q.group_bucket[g];
    d->group_mul[g] = mul;
    mul *= q.group_card[g] > 0 ? q.group_card[g] : 1;
  }
  for (int i = 0; i < MI355Q_MAX_SLOTS; ++i) d->init_vals[i] = q.init_vals[i];
  d->join_col = -1;
}

int32_t build_dev_plan(const mi355q_plan& p, const mi355q_qmd& q, DevPlan* d) {
  if (p.n_exprs != 0) {
    mi355q_plan lp;
    if (int32_t e = lower_exprs(p, &lp, nullptr)) return e;
    return build_dev_plan(lp, q, d);
  }
  std::memset(d, 0, sizeof(*d));
  const bool grouped = p.n_group_cols >= 1;
  ResolvedTarget ts[MI355Q_MAX_TARGETS];
  if (int32_t e = resolve_targets(p, grouped, ts)) return e;
  layout_from_qmd(q, d);
  // (the step itself runs on the row-wise form of the columnar decisions: either hint)
  d->col0_key_quirk = p.output_columnar_hint != 0 && q.keyless && p.n_group_cols == 1 && q.slot_width == 8 &&
                      q.slot_count > 0 && q.init_vals[0] == kEmptyKey64;
  d->n_cols = p.n_cols;
  d->n_quals = p.n_quals;
  for (int i = 0; i < p.n_quals; ++i) {
    const mi355q_qual& s = p.quals[i];
    DevQual& o = d->quals[i];
    o.col = s.col;
    o.op = MI355Q_QUAL_OP(s.op);
    o.or_group = MI355Q_QUAL_OR_GROUP(s.op);
    o.pad_ = 0;
    if (s.op < 0 || (s.op >> 16) != 0 || o.or_group > MI355Q_MAX_OR_GROUPS) return MI355Q_ERR_INVALID_PLAN;
    o.type = col_type_code(p.cols[s.col]);
    o.nullable = p.cols[s.col].nullable != 0;
    o.ival = s.ival;
    o.fval = s.fval;
    switch (o.op) {
      case MI355Q_EQ: case MI355Q_NE: case MI355Q_LT: case MI355Q_GT: case MI355Q_LE:
      case MI355Q_GE: case MI355Q_IS_NULL: case MI355Q_IS_NOT_NULL:
        break;
      default:
        return MI355Q_ERR_UNSUPPORTED;
    }
  }
  for (int i = 0; i < p.n_targets; ++i) {
    DevTarget& o = d->targets[i];
    o.col = ts[i].col;
    o.table = ts[i].table;
    o.arg_type = ts[i].arg_type;
    o.arg_nullable = ts[i].arg_nullable;
    o.arg_fp = ts[i].arg_fp;  // COUNT(double col) still decodes a double
    o.arg_f32 = ts[i].arg_f32;
    o.arg_rng = 0;
    o.arg_lo = o.arg_hi = 0;
    o.arg_has_nulls = 1;
    if (const mi355q_range* rg = ts[i].range) {
      if (rg->valid && !o.arg_fp && rg->bucket == 0 && rg->min <= rg->max && rg->min >= INT32_MIN + 1 && rg->max <= INT32_MAX) {
        o.arg_rng = 1;
        o.arg_lo = (int32_t)rg->min;
        o.arg_hi = (int32_t)rg->max;
        o.arg_has_nulls = rg->has_nulls != 0;
      }
    }
    if (ts[i].agg == MI355Q_COUNT_IF || ts[i].agg == MI355Q_SUM_IF) {
      const mi355q_qual& c = p.targets[i].cond;
      if (MI355Q_QUAL_OR_GROUP(c.op) != 0 || c.op < 0 || (c.op >> 16) != 0) return MI355Q_ERR_INVALID_PLAN;
      o.cond.col = c.col;
      o.cond.op = c.op;
      o.cond.or_group = 0;
      o.cond.pad_ = 0;
      o.cond.type = col_type_code(p.cols[c.col]);
      o.cond.nullable = p.cols[c.col].nullable != 0;
      o.cond.ival = c.ival;
      o.cond.fval = c.fval;
      switch (c.op) {
        case MI355Q_EQ: case MI355Q_NE: case MI355Q_LT: case MI355Q_GT: case MI355Q_LE: case MI355Q_GE:
        case MI355Q_IS_NULL: case MI355Q_IS_NOT_NULL:   // Select.CountIf / SumIf: COUNT_IF(x IS NULL), SUM_IF(v, x IS NOT NULL)
          break;
        default:
          return MI355Q_ERR_UNSUPPORTED;
      }
    }
  }
  for (int g = 0; g < p.n_group_cols; ++g) {
    const mi355q_col_desc& cd = p.cols[p.group_cols[g]];
    d->group_cols[g] = p.group_cols[g];
    d->group_types[g] = col_type_code(cd);
    // NULL keys are translated to max + 1 where the range says the column has them
    // (groupByColumnCodegen translate_null_val, IRCodegen.cpp:1413-1512)
    d->group_translate[g] = q.desc_type == MI355Q_GROUP_BY_PERFECT_HASH && cd.nullable &&
                            (p.n_group_cols == 1 || q.group_has_nulls[g]);
  }
  d->group_col = grouped ? d->group_cols[0] : -1;
  d->group_type = grouped ? d->group_types[0] : 0;
  d->group_nullable = grouped ? p.cols[p.group_cols[0]].nullable != 0 : 0;
  return MI355Q_OK;
}

}  // namespace mq
