// emu.cpp — TEST-ONLY host emulation of the generic kernel's row logic.
//
// Compiles heavydb_amd/csrc/rowfunc.h + plan.cpp with plain g++ and -DMQ_EMU (device atomics
// become single-threaded plain ops) so the plan-time layout decisions and the per-row
// semantics of the product can be checked against the oracle on a machine without a GPU.
// It is never part of libmi355q.so and never reachable from the product API.
#include <cstring>
#include <vector>

#include "../../heavydb_amd/csrc/plan.h"
#include "emu_atomics.h"  // before rowfunc.h: the host stand-ins of its atomics
#include "../../heavydb_amd/csrc/rowfunc.h"
#include "../../heavydb_amd/csrc/expr.h"

using namespace mq;

extern "C" int32_t emu_qmd_init(const mi355q_plan* p, mi355q_qmd* q) { return qmd_init(*p, q); }

// api.cpp col_layout_of
static ColLayout emu_col_layout(const mi355q_qmd& q) {
  ColLayout L{};
  L.entry_count = q.entry_count;
  L.slot_col_bytes = ((int64_t)q.slot_width * q.entry_count + 7) & ~(int64_t)7;
  L.key_quads = q.key_bytes / 8;
  L.slot_count = q.slot_count;
  L.slot_width = q.slot_width;
  L.row_quad = q.row_size / 8;
  return L;
}

extern "C" int64_t emu_buffer_bytes(const mi355q_qmd* q) { return qmd_buffer_bytes(*q); }
extern "C" int64_t emu_group_col_offset(const mi355q_qmd* q, int g) { return qmd_group_col_offset(*q, g); }
extern "C" int64_t emu_slot_col_offset(const mi355q_qmd* q, int s) { return qmd_slot_col_offset(*q, s); }

// initColumnarGroups through the product's code (k_init_columns: entry_to_columns of the init row)
extern "C" void emu_init_buffer(const mi355q_qmd* q, int64_t* buf) {
  int64_t img[MI355Q_MAX_GROUP_COLS + MI355Q_MAX_SLOTS];
  row_init_image(*q, img);
  if (q->output_columnar) {
    const ColLayout L = emu_col_layout(*q);
    for (int64_t e = 0; e < q->entry_count; ++e) entry_to_columns(L, img, (int8_t*)buf, e);
  } else {
    for (int64_t e = 0; e < q->entry_count; ++e) std::memcpy(buf + e * (q->row_size / 8), img, q->row_size);
  }
}

// the product's evaluator (heavydb_amd/csrc/expr.h) on one row: value pattern, result type; 0 or ErrorCode 7
extern "C" int32_t emu_eval_expr(const mi355q_plan* plan, int32_t k, const void* const* cols, int64_t pos,
                                 int64_t* out_bits, int32_t* out_type) {
  mi355q_plan lp;
  DevExprSet xs;
  if (int32_t e = lower_exprs(*plan, &lp, &xs)) return e;
  if (k < 0 || k >= xs.n) return MI355Q_ERR_INVALID_PLAN;
  // (one program on the caller's PHYSICAL columns: a program that reads the value of an earlier expression needs the
  // projection's extended column table — emu_execute — as orc_eval_expr says too)
  for (int i = 0; i < xs.e[k].n_nodes; ++i)
    if (xs.e[k].nodes[i].op == MI355Q_EX_COL && xs.e[k].nodes[i].arg >= plan->n_cols) return MI355Q_ERR_INVALID_PLAN;
  int32_t err = 0;
  *out_bits = eval_expr(xs.e[k], (const int8_t* const*)cols, pos, &err);
  *out_type = xs.e[k].type;
  return err;
}

static void emu_attach_join(const mi355q_plan* plan, const mi355q_inputs* in, int join_hash_type, const void* join_buf,
                            int64_t join_min, int64_t join_max, int64_t join_entries, int join_n_keys, int join_width,
                            DevPlan* dp) {
  DevPlan& d = *dp;
  if (plan->join_outer_col < 0) return;
  const int nk = plan->n_join_cols > 1 ? plan->n_join_cols : 1;
  for (int i = 0; i < nk; ++i) {
    const int c = (i == 0 && plan->n_join_cols <= 1) ? plan->join_outer_col : plan->join_outer_cols[i];
    d.join_cols[i] = c;
    d.join_types[i] = col_type_code(plan->cols[c]);
    d.join_nullables[i] = plan->cols[c].nullable != 0;
  }
  d.join_col = d.join_cols[0];
  d.join_type = d.join_types[0];
  d.join_nullable = d.join_nullables[0];
  d.join_n_keys = join_n_keys;
  d.join_width = join_width;
  d.join_kind = plan->join_kind;
  d.join_hash_type = join_hash_type;
  d.join_buf = join_buf;
  d.join_min = join_min;
  d.join_max = join_max;
  d.join_entries = join_entries;
  for (int i = 0; i < plan->n_inner_cols; ++i) d.inner_cols[i] = (const int8_t*)in->inner_col_buffers[i];
}

extern "C" int32_t emu_execute(const mi355q_plan* plan, const mi355q_inputs* in,
                               int join_hash_type, const void* join_buf, int64_t join_min,
                               int64_t join_max, int64_t join_entries, int join_n_keys,
                               int join_width, int64_t* out, mi355q_qmd* out_qmd) {
  if (plan->n_exprs != 0) {
    // like mi355q_execute (api.cpp execute_projected + kernels_generic.hip k_project): the expressions are
    // evaluated into dense temporary columns and the step runs on the lowered plan
    mi355q_plan lp;
    DevExprSet xs;
    if (int32_t e = lower_exprs(*plan, &lp, &xs, true)) return e;
    mi355q_qmd ql;
    if (int32_t e = qmd_init(*plan, &ql)) return e;
    DevPlan dl;
    if (int32_t e = build_dev_plan(lp, ql, &dl)) return e;
    emu_attach_join(&lp, in, join_hash_type, join_buf, join_min, join_max, join_entries, join_n_keys, join_width, &dl);
    const int nc = plan->n_cols, nx = plan->n_exprs, nc2 = nc + nx;
    const uint32_t qual_expr_mask = expr_qual_mask(*plan);
    std::vector<std::vector<int64_t>> store((size_t)in->n_frags * nx);
    std::vector<const void*> cols2((size_t)in->n_frags * nc2);
    for (int f = 0; f < in->n_frags; ++f) {
      const int64_t n = in->num_rows[f];
      for (int c = 0; c < nc; ++c) cols2[(size_t)f * nc2 + c] = in->col_buffers[(size_t)f * nc + c];
      for (int k = 0; k < nx; ++k) {
        store[(size_t)f * nx + k].assign((size_t)n + 2, 0);
        cols2[(size_t)f * nc2 + nc + k] = store[(size_t)f * nx + k].data();
      }
      const int8_t* const* fc = (const int8_t* const*)(cols2.data() + (size_t)f * nc2);
      for (int64_t pos = 0; pos < n; ++pos) {
        uint32_t err_mask = 0;
        int32_t first_err = 0, first_qual_err = 0;  // as k_project: the first failing expression's code (7 / 1)
        for (int k = 0; k < nx; ++k) {
          int32_t err = 0;
          const int64_t v = eval_expr(xs.e[k], fc, pos, &err);
          store_expr_value((int8_t*)fc[nc + k], xs.e[k], pos, v);
          if (err) {
            err_mask |= 1u << k;
            if (!first_err) first_err = err;
            if (!first_qual_err && ((qual_expr_mask >> k) & 1u)) first_qual_err = err;
          }
        }
        if (!err_mask) continue;
        bool counts = (err_mask & qual_expr_mask) != 0;
        if (!counts) {
          counts = quals_pass(dl, fc, pos);
          if (counts && dl.join_col >= 0 && dl.join_kind != MI355Q_JOIN_LEFT) {
            int64_t jk[MI355Q_MAX_GROUP_COLS];
            bool null_key = false;
            for (int i = 0; i < dl.join_n_keys; ++i) {
              jk[i] = decode_int(fc[dl.join_cols[i]], dl.join_types[i], pos);
              null_key = null_key || (dl.join_nullables[i] && jk[i] == int_null_of(dl.join_types[i]));
            }
            counts = !null_key && join_lookup(dl, jk).count > 0;
          }
        }
        if (counts) return first_qual_err ? first_qual_err : first_err;
      }
    }
    mi355q_inputs in2 = *in;
    in2.col_buffers = cols2.data();
    return emu_execute(&lp, &in2, join_hash_type, join_buf, join_min, join_max, join_entries, join_n_keys, join_width,
                       out, out_qmd);
  }
  mi355q_qmd q;
  if (int32_t e = qmd_init(*plan, &q)) return e;
  if (q.output_columnar) {
    // like mi355q_execute: the step runs on the row-wise form of the same decisions, then every
    // entry is moved into the columns (rowfunc.h entry_to_columns)
    mi355q_plan pr = *plan;
    pr.output_columnar_hint = MI355Q_OUTPUT_ROWWISE_COLUMNAR_DECISIONS;
    mi355q_qmd qr;
    if (int32_t e = qmd_init(pr, &qr)) return e;
    if (qr.entry_count != q.entry_count || qr.row_size != q.row_size || qr.key_bytes != q.key_bytes ||
        qr.slot_count != q.slot_count || qr.slot_width != q.slot_width)
      return MI355Q_ERR_UNSUPPORTED;
    std::vector<int64_t> rows((size_t)qr.entry_count * (qr.row_size / 8));
    if (int32_t e = emu_execute(&pr, in, join_hash_type, join_buf, join_min, join_max, join_entries, join_n_keys,
                                join_width, rows.data(), nullptr))
      return e;
    const ColLayout L = emu_col_layout(q);
    for (int64_t e = 0; e < q.entry_count; ++e) entry_to_columns(L, rows.data() + e * L.row_quad, (int8_t*)out, e);
    if (out_qmd) *out_qmd = q;
    return 0;
  }
  if (q.slot_width == 4) {
    // like mi355q_execute: the step runs on the 8-byte layout of the same plan, then every row is
    // narrowed (rowfunc.h narrow_row)
    mi355q_plan p8 = *plan;
    p8.bigint_count = 1;
    mi355q_qmd q8;
    if (int32_t e = qmd_init(p8, &q8)) return e;
    std::vector<int64_t> wide((size_t)q8.entry_count * (q8.row_size / 8));
    if (int32_t e = emu_execute(&p8, in, join_hash_type, join_buf, join_min, join_max, join_entries, join_n_keys,
                                join_width, wide.data(), nullptr))
      return e;
    for (int64_t e = 0; e < q.entry_count; ++e) {
      narrow_row(wide.data() + e * (q8.row_size / 8), q.key_bytes / 8, q.slot_count, q.row_size / 8,
                 out + e * (q.row_size / 8));
    }
    if (out_qmd) *out_qmd = q;
    return 0;
  }
  DevPlan d;
  if (int32_t e = build_dev_plan(*plan, q, &d)) return e;
  emu_attach_join(plan, in, join_hash_type, join_buf, join_min, join_max, join_entries, join_n_keys, join_width, &d);
  if (out_qmd) *out_qmd = q;
  const int rq = q.row_size / 8;
  for (int64_t e = 0; e < q.entry_count; ++e) row_init_image(q, out + e * rq);
  const bool ng = q.desc_type == MI355Q_NON_GROUPED_AGGREGATE;
  // emulate several "threads" for the non-grouped fold: rows are dealt round-robin to 7
  // partial rows which are then folded exactly like the kernel's block fold + global merge
  constexpr int kThreads = 7;
  std::vector<int64_t> loc(kThreads * (MI355Q_MAX_SLOTS + 1));
  for (int t = 0; t < kThreads; ++t)
    for (int s = 0; s < d.slot_count; ++s) loc[t * (MI355Q_MAX_SLOTS + 1) + s] = d.init_vals[s];
  for (int f = 0; f < in->n_frags; ++f) {
    const int8_t* const* cols = (const int8_t* const*)(in->col_buffers + (size_t)f * plan->n_cols);
    for (int64_t pos = 0; pos < in->num_rows[f]; ++pos) {
      int32_t e;
      if (ng) {
        e = process_row<false>(d, cols, pos, out, loc.data() + (pos % kThreads) * (MI355Q_MAX_SLOTS + 1));
      } else {
        e = process_row<true>(d, cols, pos, out, nullptr);
      }
      if (e) return e;
    }
  }
  if (ng) {
    for (int ti = 0; ti < d.n_targets; ++ti) {
      const DevTarget& t = d.targets[ti];
      int64_t acc[2] = {d.init_vals[t.slot], t.agg == MI355Q_AVG ? d.init_vals[t.slot + 1] : 0};
      DevTarget lt = t;
      lt.slot = 0;
      int64_t lin[2] = {d.init_vals[t.slot], 0};
      for (int th = 0; th < kThreads; ++th) {
        reduce_target<false>(lt, lin, acc, loc.data() + th * (MI355Q_MAX_SLOTS + 1) + t.slot);
      }
      int64_t that[2] = {acc[0], acc[1]};
      reduce_target<true>(lt, lin, out + t.slot, that);
    }
  }
  return 0;
}

// this += that through the same code the k_reduce kernel runs per entry
extern "C" int32_t emu_reduce(const mi355q_qmd* q, int64_t* this_buf, const int64_t* that_rows,
                              int64_t that_entries) {
  if (q->output_columnar) {
    // mi355q_result_reduce on columnar handles: row-wise twins, the reduce kernel's code, store back
    mi355q_qmd qr = *q;
    qr.output_columnar = 0;
    const ColLayout L = emu_col_layout(*q);
    std::vector<int64_t> a((size_t)q->entry_count * L.row_quad), b((size_t)that_entries * L.row_quad);
    ColLayout Lb = L;
    Lb.entry_count = that_entries;
    Lb.slot_col_bytes = ((int64_t)q->slot_width * that_entries + 7) & ~(int64_t)7;
    for (int64_t e = 0; e < q->entry_count; ++e) entry_from_columns(L, (const int8_t*)this_buf, e, a.data() + e * L.row_quad);
    for (int64_t e = 0; e < that_entries; ++e) entry_from_columns(Lb, (const int8_t*)that_rows, e, b.data() + e * L.row_quad);
    if (int32_t err = emu_reduce(&qr, a.data(), b.data(), that_entries)) return err;
    for (int64_t e = 0; e < q->entry_count; ++e) entry_to_columns(L, a.data() + e * L.row_quad, (int8_t*)this_buf, e);
    return 0;
  }
  DevPlan d;
  std::memset(&d, 0, sizeof(d));
  layout_from_qmd(*q, &d);
  for (int64_t e = 0; e < that_entries; ++e) {
    if (int32_t err = reduce_entry<true>(d, q->idx_target_as_key, this_buf, that_rows + e * d.row_quad, e)) return err;
  }
  return 0;
}
