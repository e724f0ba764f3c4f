// plan.h — internal host-side declarations of libmi355q (not part of the C-ABI).
#pragma once

#include "dev_common.h"

namespace mq {

struct ResolvedTarget {
  int agg = 0, col = -1, table = 0;
  int arg_type = 0;  // type code (dev_common.h)
  int key_idx = 0;   // PROJECT_KEY: index into group_cols
  bool cond_nullable = false;  // COUNT_IF / SUM_IF: the condition column is nullable
  bool arg_nullable = false, arg_fp = false, skip_null = false;
  bool arg_f32 = false;  // FLOAT argument: single-precision slot arithmetic
  bool constrained = false;  // constrained_not_null: a qual `arg IS NOT NULL`
  int n_slots = 1;
  const mi355q_range* range = nullptr;
};

int col_type_code(const mi355q_col_desc& c);  // < 0 = invalid
int32_t resolve_targets(const mi355q_plan& p, bool grouped, ResolvedTarget* out);
int32_t qmd_init(const mi355q_plan& p, mi355q_qmd* q);
int32_t qmd_init_projection(const mi355q_plan& p, const ResolvedTarget* ts, mi355q_qmd* q);  // the Projection descriptor
// While one lives (per thread), qmd_init lays multi-column integer keys out as a perfect hash up to `max_entries`
// entries instead of g_baseline_groupby_threshold: the library's own intermediate table of a baseline step (api.cpp
// execute_perfect_twin); never the layout a caller sees.
struct PerfectTwinScope {
  explicit PerfectTwinScope(int64_t max_entries);
  ~PerfectTwinScope();
  int64_t saved;
};
int64_t qmd_buffer_bytes(const mi355q_qmd& q);
int64_t qmd_group_col_offset(const mi355q_qmd& q, int g);  // columnar descriptors; -1 otherwise
int64_t qmd_slot_col_offset(const mi355q_qmd& q, int s);
int32_t build_dev_plan(const mi355q_plan& p, const mi355q_qmd& q, DevPlan* d);
// the layout part of a DevPlan (what reduce / iteration / sort need) from a descriptor alone
void layout_from_qmd(const mi355q_qmd& q, DevPlan* d);
// Projected expressions (mi355q_expr): validates every program and returns the plan in which expression k
// is an ordinary described column n_cols + k (its result type / nullability, the caller's range) and
// n_exprs == 0 — the plan every kernel family and the layout code see after the projection pass
// (kernels_generic.hip k_project) has written the expression's values into a dense temporary column.
// `dev` (optional) receives the lowered programs.
// widen_filter_bools: an INT8 (BOOLEAN) expression that ONLY quals read is described — and stored by the projection pass —
// as an INT32 column (its NULL becomes the INT32 NULL), so that the typed families, which filter on 4- and 8-byte
// columns, take the lowered step
int32_t lower_exprs(const mi355q_plan& p, mi355q_plan* lowered, DevExprSet* dev, bool widen_filter_bools = false);
uint32_t expr_qual_mask(const mi355q_plan& p);  // expressions evaluated for every row (read by a qual, directly or through another)
// one initialised row (key quads then slot init values); quad holds row_size / 8 entries
void row_init_image(const mi355q_qmd& q, int64_t* quad);

}  // namespace mq
