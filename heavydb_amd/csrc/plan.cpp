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
  bool fp = false;
  const mi355q_range* range = nullptr;
};

bool valid_type(int t) { return t >= MI355Q_INT8 && t <= MI355Q_DOUBLE; }

// Initial slot value for an aggregate whose init type has `notnull`.
int64_t initial_val(int agg, const ArgInfo& a, bool notnull) {
  switch (agg) {
    case MI355Q_SUM:
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
      case MI355Q_PROJECT_KEY:
        if (!grouped) return MI355Q_ERR_INVALID_PLAN;
        r.col = p.group_cols[0];
        r.table = 0;
        break;
      default:
        return MI355Q_ERR_UNSUPPORTED;
    }
    if (r.col >= 0) {
      const int ncols = r.table ? p.n_inner_cols : p.n_cols;
      if (r.col >= ncols || (r.table && p.join_outer_col < 0)) return MI355Q_ERR_INVALID_PLAN;
      const mi355q_col_desc& cd = r.table ? p.inner_cols[r.col] : p.cols[r.col];
      if (!valid_type(cd.type)) return MI355Q_ERR_INVALID_PLAN;
      r.arg_type = cd.type;
      r.arg_nullable = cd.nullable != 0;
      r.arg_fp = type_is_fp(cd.type);
      r.range = r.table ? &p.inner_col_ranges[r.col] : &p.col_ranges[r.col];
    }
    const bool is_agg = t.agg != MI355Q_PROJECT_KEY;
    r.skip_null = is_agg && r.col >= 0 && (r.arg_nullable || !grouped);
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
      ArgInfo a{t.arg_type, t.arg_nullable, t.arg_fp, t.range};
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
          if (t.arg_nullable) {
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

int32_t qmd_init(const mi355q_plan& p, mi355q_qmd* q) {
  std::memset(q, 0, sizeof(*q));
  if (p.abi_version != MI355Q_ABI_VERSION) return MI355Q_ERR_INVALID_PLAN;
  if (p.n_cols < 0 || p.n_cols > MI355Q_MAX_COLS || p.n_inner_cols < 0 ||
      p.n_inner_cols > MI355Q_MAX_COLS || p.n_quals < 0 || p.n_quals > MI355Q_MAX_QUALS ||
      p.n_targets < 1 || p.n_targets > MI355Q_MAX_TARGETS || p.n_group_cols < 0) {
    return MI355Q_ERR_INVALID_PLAN;
  }
  if (p.n_group_cols > 1) return MI355Q_ERR_UNSUPPORTED;  // multi-column keys: SURVEY f1
  for (int i = 0; i < p.n_cols; ++i) {
    if (!valid_type(p.cols[i].type)) return MI355Q_ERR_INVALID_PLAN;
  }
  for (int i = 0; i < p.n_quals; ++i) {
    if (p.quals[i].col < 0 || p.quals[i].col >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
  }
  const bool grouped = p.n_group_cols == 1;
  ResolvedTarget ts[MI355Q_MAX_TARGETS];
  if (int32_t e = resolve_targets(p, grouped, ts)) return e;

  q->n_targets = p.n_targets;
  q->group_col_count = p.n_group_cols;
  q->idx_target_as_key = -1;
  q->key_width = 8;
  q->entry_count = 1;
  q->desc_type = MI355Q_NON_GROUPED_AGGREGATE;

  if (grouped) {
    const int gc = p.group_cols[0];
    if (gc < 0 || gc >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
    if (type_is_fp(p.cols[gc].type)) return MI355Q_ERR_UNSUPPORTED;  // fp keys
    const mi355q_range& r = p.col_ranges[gc];
    bool use_baseline = !r.valid || r.min > r.max;
    if (!use_baseline) {
      const int64_t col_count = p.n_group_cols + p.n_targets;
      const int64_t max_entries = kMaxBufferSize / (col_count * (int64_t)sizeof(int64_t));
      const __int128 span = (__int128)r.max - (__int128)r.min;
      use_baseline = span >= (__int128)max_entries;
    }
    if (use_baseline) {
      q->desc_type = MI355Q_GROUP_BY_BASELINE_HASH;
      q->entry_count =
          p.max_groups_buffer_entry_guess > 0 ? p.max_groups_buffer_entry_guess : 16384;
      if (q->entry_count > (int64_t)UINT32_MAX) return MI355Q_ERR_UNSUPPORTED;  // h is uint32
      if (r.valid && !(type_width(p.cols[gc].type) == 8 && r.has_nulls) &&
          r.min > (int64_t)INT32_MIN && r.max < (int64_t)kEmptyKey32 - 1) {
        q->key_width = 4;
      }
    } else {
      q->desc_type = MI355Q_GROUP_BY_PERFECT_HASH;
      q->min_val = r.min;
      q->max_val = r.max;
      q->has_nulls = r.has_nulls != 0;
      const int64_t card = r.max - r.min + 1 + (r.has_nulls ? 1 : 0);
      q->entry_count = card > 1 ? card : 1;
      bool keyless = false;
      int key_slot = 0;
      keyless_decision(p, ts, &keyless, &key_slot);
      q->keyless = keyless;
      q->idx_target_as_key = key_slot;
    }
  }

  int slot = 0;
  for (int i = 0; i < p.n_targets; ++i) {
    const ResolvedTarget& t = ts[i];
    ArgInfo a{t.arg_type, t.arg_nullable, t.arg_fp, t.range};
    q->target_agg[i] = t.agg;
    q->target_skip_null[i] = t.skip_null;
    q->target_arg_is_fp[i] = t.arg_fp && t.agg != MI355Q_COUNT;
    q->target_is_fp[i] = t.agg == MI355Q_AVG || (t.arg_fp && t.agg != MI355Q_COUNT);
    const bool key_in_row =
        t.agg == MI355Q_PROJECT_KEY && q->desc_type == MI355Q_GROUP_BY_BASELINE_HASH;
    if (key_in_row) {
      q->target_slot[i] = -1;  // target_groupby_indices: read from the key column
    } else {
      if (slot + t.n_slots > MI355Q_MAX_SLOTS) return MI355Q_ERR_INVALID_PLAN;
      q->target_slot[i] = slot;
      const bool init_notnull = grouped ? !t.arg_nullable : false;
      q->init_vals[slot] = initial_val(t.agg, a, init_notnull);
      if (t.agg == MI355Q_AVG) q->init_vals[slot + 1] = 0;
      slot += t.n_slots;
    }
    switch (t.agg) {
      case MI355Q_AVG:
        q->target_null[i] = kNullDoubleBits;
        break;
      case MI355Q_SUM:
        q->target_null[i] = t.arg_fp ? kNullDoubleBits : INT64_MIN;
        break;
      case MI355Q_COUNT:
        q->target_null[i] = p.bigint_count ? INT64_MIN : (int64_t)INT32_MIN;
        break;
      default:
        q->target_null[i] = t.arg_fp ? kNullDoubleBits : int_null_of(t.arg_type);
    }
  }
  q->slot_count = slot;
  q->key_bytes = (grouped && !q->keyless) ? ((q->group_col_count * q->key_width + 7) & ~7) : 0;
  q->row_size = q->key_bytes + 8 * q->slot_count;
  if (q->row_size <= 0) return MI355Q_ERR_INVALID_PLAN;
  return MI355Q_OK;
}

int32_t build_dev_plan(const mi355q_plan& p, const mi355q_qmd& q, DevPlan* d) {
  std::memset(d, 0, sizeof(*d));
  const bool grouped = p.n_group_cols == 1;
  ResolvedTarget ts[MI355Q_MAX_TARGETS];
  if (int32_t e = resolve_targets(p, grouped, ts)) return e;
  d->n_cols = p.n_cols;
  d->n_quals = p.n_quals;
  for (int i = 0; i < p.n_quals; ++i) {
    const mi355q_qual& s = p.quals[i];
    DevQual& o = d->quals[i];
    o.col = s.col;
    o.op = s.op;
    o.type = p.cols[s.col].type;
    o.nullable = p.cols[s.col].nullable != 0;
    o.ival = s.ival;
    o.fval = s.fval;
    switch (s.op) {
      case MI355Q_EQ: case MI355Q_NE: case MI355Q_LT: case MI355Q_GT: case MI355Q_LE:
      case MI355Q_GE:
        break;
      default:
        return MI355Q_ERR_UNSUPPORTED;
    }
  }
  d->n_targets = p.n_targets;
  for (int i = 0; i < p.n_targets; ++i) {
    DevTarget& o = d->targets[i];
    o.agg = ts[i].agg;
    o.col = ts[i].col;
    o.table = ts[i].table;
    o.arg_type = ts[i].arg_type;
    o.arg_nullable = ts[i].arg_nullable;
    o.skip_null = ts[i].skip_null;
    o.slot = q.target_slot[i];
    o.arg_fp = ts[i].arg_fp;
  }
  d->slot_count = q.slot_count;
  d->desc_type = q.desc_type;
  d->keyless = q.keyless;
  d->key_width = q.key_width;
  d->row_quad = q.row_size / 8;
  d->key_quad = q.key_bytes / 8;
  d->group_col = grouped ? p.group_cols[0] : -1;
  d->group_type = grouped ? p.cols[p.group_cols[0]].type : 0;
  d->group_nullable = grouped ? p.cols[p.group_cols[0]].nullable != 0 : 0;
  d->entry_count = q.entry_count;
  d->min_val = q.min_val;
  d->max_val = q.max_val;
  for (int i = 0; i < MI355Q_MAX_SLOTS; ++i) d->init_vals[i] = q.init_vals[i];
  d->join_col = -1;
  return MI355Q_OK;
}

}  // namespace mq
