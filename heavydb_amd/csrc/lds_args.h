// lds_args.h — the plan as the LDS group-by families see it (kernels_lds.hip: the table replicated / windowed in every
// workgroup's LDS; kernels_idx.hip: perfect-hash tables too large for that, partitioned by entry index).
#pragma once

#include <cstring>

#include "boolfilter.h"
#include "fast_common.h"

namespace mq {

using namespace fast;

constexpr int kLdsVals = 3;              // value columns
constexpr int kLdsKeys = 3;              // key columns (perfect hash)
constexpr int kLdsGenericFlt = 4;        // filter columns of the run-time-role member's widest instantiation (NF)
constexpr uint32_t kLdsHashSmall = 256;  // slots of one baseline replica, first attempt (many replicas)
constexpr uint32_t kLdsHashMax = 4096;   // ... at most, second attempt
constexpr uint32_t kLdsMaxWindows = 8;   // windows of a table that does not fit one LDS (the columns are read once per window)

struct LdsVal {
  int32_t col, type, nullable;           // type: MI355Q_INT32 / _INT64 / _DOUBLE (plain)
  int32_t off_cnt, off_sum, off_min, off_max;  // byte offsets of the arrays inside a replica, -1 = not kept
};
struct LdsArgs {
  int32_t n_vals, n_flt, n_keys;
  int32_t baseline;                      // 0: perfect-hash index; 1: open addressing on one 8-byte key
  uint32_t entries;                      // arrays' length: entries of ONE window (perfect), or the hash slots (power of two)
  uint32_t windows;                      // T >= 1: workgroup b keeps the rows of window b % T
  int32_t copies_lg;                     // log2(K)
  uint32_t copy_bytes;                   // bytes of one replica (16-byte multiple)
  int32_t off_rows, off_keys;            // rows[entries] (u32); keys[entries] (int64, baseline)
  LdsVal v[kLdsVals];
  RangeFilter flt[MI355Q_MAX_QUALS];
  int32_t flt_type[MI355Q_MAX_QUALS];
  int32_t key_col[kLdsKeys], key_type[kLdsKeys], key_translate[kLdsKeys];
  int64_t key_min[kLdsKeys], key_card[kLdsKeys], key_mul[kLdsKeys], key_null_key[kLdsKeys];
  int32_t target_v[MI355Q_MAX_TARGETS];  // index into v[] of each target's argument, -1 = none
  // typed members (k_groupby_lds_typed): every value column a plain INT32, no quals; one replica =
  //   keys[E] i64 (baseline) | sum[NV][E] i64 | rows[E] u32 | cnt[NV][E] u32 | min[NV][E] i32 | max[NV][E] i32 (mm only)
  int32_t typed, mm;
  uint32_t xcd_aware;                    // windows: stripe-mates on one XCD (lds_window_map); set by the launcher
  uint32_t t_off_keys, t_off_sum, t_off_rows, t_off_cnt, t_off_min, t_off_max;
  // a filter compiled at plan time (boolfilter.h) instead of range quals: flt[k].col / flt_type[k] name its columns
  int32_t bf_on, pad_bf_;
  const BoolFilter* bf;                  // DEVICE memory
};


// What a few-groups / index-partitioned GROUP BY needs of a plan: quals, key columns (perfect hash: 1 - 3 plain integer columns
// with their ranges; baseline: one 8-byte-wide key), value columns and the accumulators the targets need of each
// (need[c] = {non-NULL count, sum, min, max}).  Perfect-hash tables of more than `max_entries` entries are refused.
// allow_int8_flt: the caller's kernels load 1-byte filter columns (kernels_lds.hip)
inline bool lds_describe(const DevPlan& p, const FragView& fv, int64_t max_entries, uint32_t knob_flags, LdsArgs* out,
                         bool (&need)[kLdsVals][4], bool allow_int8_flt = false) {
  LdsArgs& a = *out;
  std::memset(&a, 0, sizeof(a));
  a.windows = 1;
  if (p.desc_type == MI355Q_NON_GROUPED_AGGREGATE || p.join_col >= 0 || p.col0_key_quirk || p.slot_width != 8) return false;
  if (p.n_quals > MI355Q_MAX_QUALS) return false;
  for (int i = 0; i < p.n_quals; ++i) {
    if (!make_range_filter(p.quals[i], &a.flt[i], allow_int8_flt)) return false;
    a.flt_type[i] = p.quals[i].type;
    if (!all_aligned16(fv, p.quals[i].col)) return false;
  }
  a.n_flt = merge_range_filters(a.flt, a.flt_type, p.n_quals);
  if (p.bf_active) {  // the compiled filter's columns take the filter slots
    const BoolFilter* bf = step_bool_filter();
    if (!bf || p.n_quals != 0 || bf->n_cols > 4) return false;
    // program atoms: only in their lean form (PairAtom: INT32 operands, one operation), at most kLdsFusedProgs of them, and
    // only in the typed member (make_lds_args) — everything else takes the row-mask pre-pass (kernels_filter.hip)
    if (bf->n_progs != 0 && !(bf->all_lean && bf->all_i32 && bf->n_progs <= kLdsFusedProgs)) return false;
    for (int k = 0; k < bf->n_cols; ++k) {
      if (!all_aligned16(fv, bf->col[k])) return false;
      a.flt[k] = no_filter();
      a.flt[k].col = bf->col[k];
      a.flt_type[k] = bf->col_type[k];
    }
    a.n_flt = bf->n_cols;
    a.bf_on = 1;
    a.bf = step_bool_filter_dev();
  }
  // keys
  if (p.desc_type == MI355Q_GROUP_BY_PERFECT_HASH) {
    if (p.n_group < 1 || p.n_group > kLdsKeys || p.entry_count < 1 || p.entry_count > max_entries) return false;
    for (int g = 0; g < p.n_group; ++g) {
      if (p.group_types[g] != MI355Q_INT32 && p.group_types[g] != MI355Q_INT64) return false;
      if (p.group_bucket[g] != 0 || !all_aligned16(fv, p.group_cols[g])) return false;
      a.key_col[g] = p.group_cols[g];
      a.key_type[g] = p.group_types[g];
      a.key_translate[g] = p.group_translate[g];
      a.key_min[g] = p.group_min[g];
      a.key_card[g] = p.group_card[g];
      a.key_mul[g] = p.group_mul[g];
      a.key_null_key[g] = p.group_null_key[g];
      if (g > 0 && p.group_mul[g] < p.group_mul[g - 1]) return false;  // (the flush divides in descending order)
    }
    a.n_keys = p.n_group;
    a.entries = (uint32_t)p.entry_count;
  } else if (p.desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
    // one 8-byte-wide key column (BIGINT, or DOUBLE as its bit pattern); 4-byte integer keys take the value
    // sign-extended, FLOAT keys the bit pattern of the double they widen to
    if (p.n_group != 1) return false;
    const int kt = p.group_types[0];
    if (kt != MI355Q_INT64 && kt != MI355Q_DOUBLE && kt != MI355Q_INT32 && kt != MI355Q_FLOAT) return false;
    if (!all_aligned16(fv, p.group_cols[0])) return false;
    a.key_col[0] = p.group_cols[0];
    a.key_type[0] = kt;
    a.n_keys = 1;
    a.baseline = 1;
    // the group count of a baseline table is only known afterwards: first 256-slot replicas (a table with a handful
    // of groups gets one replica per few lanes), then the largest replica that fits, then kLdsMaxWindows windows (classes
    // of a key hash) of it, then another family.  (Round 4 tried to size the second attempt from the table's entry count
    // — the caller's NDV guess x 2 — and skip the first: the reference's default guess is 16 384 whatever the data holds
    // (g_default_max_groups_buffer_entry_guess, Execute.cpp:111), so BH001's ten groups got three windows and ran 2.4 x
    // slower, 5.5 -> 13.5 ms per 1 B rows, profiles/r04_refbench_sel_call5.jsonl; reverted.)
    // (Also measured: FOUR windows as the third attempt where an NDV estimate says the groups fill them to 0.61 — BH007,
    // 10 K groups: 32.2 ms against 29.6 ms in eight, profiles/r04_refbench_bh007_four_windows_call20.jsonl: the four-window
    // tables overflow their probe limit and the eight-window rung runs after all; reverted.)
    const uint32_t fl = knob_flags;
    a.entries = (fl & (MI355Q_OPT_LDS_BASELINE_LARGE | MI355Q_OPT_LDS_BASELINE_WINDOWS)) ? kLdsHashMax : kLdsHashSmall;
    if (fl & MI355Q_OPT_LDS_BASELINE_WINDOWS) a.windows = kLdsMaxWindows;
  } else {
    return false;
  }
  // targets -> value columns and the accumulators each needs
  for (int c = 0; c < kLdsVals; ++c) a.v[c].off_cnt = a.v[c].off_sum = a.v[c].off_min = a.v[c].off_max = -1;
  for (int c = 0; c < kLdsVals; ++c)
    for (int k = 0; k < 4; ++k) need[c][k] = false;
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    a.target_v[i] = -1;
    if (t.agg == MI355Q_PROJECT_KEY) continue;
    if (t.table != 0 || t.arg_f32) return false;
    if (t.agg == MI355Q_COUNT && t.col < 0) continue;
    if (t.agg != MI355Q_COUNT && t.agg != MI355Q_SUM && t.agg != MI355Q_MIN && t.agg != MI355Q_MAX && t.agg != MI355Q_AVG) return false;
    if (t.col < 0) return false;
    if (t.arg_type != MI355Q_INT32 && t.arg_type != MI355Q_INT64 && t.arg_type != MI355Q_DOUBLE) return false;
    int c = -1;
    for (int k = 0; k < a.n_vals; ++k)
      if (a.v[k].col == t.col) c = k;
    if (c < 0) {
      if (a.n_vals >= kLdsVals || !all_aligned16(fv, t.col)) return false;
      c = a.n_vals++;
      a.v[c].col = t.col;
      a.v[c].type = t.arg_type;
      a.v[c].nullable = t.skip_null;
    } else if (a.v[c].nullable != t.skip_null) {
      return false;
    }
    a.target_v[i] = c;
    if (t.skip_null) need[c][0] = true;  // (a NOT NULL column's count is the entry's `rows`)
    if (t.agg == MI355Q_SUM || t.agg == MI355Q_AVG) need[c][1] = true;
    if (t.agg == MI355Q_MIN) need[c][2] = true;
    if (t.agg == MI355Q_MAX) need[c][3] = true;
  }
  return true;
}

}  // namespace mq
