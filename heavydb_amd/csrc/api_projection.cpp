// api_projection.cpp — the host side of the PROJECTION family (kernels_proj.hip): plan -> descriptor -> one
// compaction launch -> result handle, and the accessors that iterate a Projection buffer.
//
// Reference shape (heavyai/heavydb): Executor::executePlanWithGroupBy with a Projection descriptor
// (Execute.cpp:4179-4366) — the kernel gets total_matched / max_matched (KernelParam::TOTAL_MATCHED / MAX_MATCHED,
// enums.h:62-77; QueryExecutionContext.cpp:660-735), the buffer is compacted to the matched count afterwards
// (compactProjectionBuffersCpu / Gpu, QueryMemoryInitializer.cpp) — and ResultSet iteration over
// QueryDescriptionType::Projection storage (ResultSetIteration.cpp: getNextRowImpl skips entries whose key is
// EMPTY_KEY_64).  The seam's own example of this layout is run_query_external (ExternalExecutor.cpp:408-515).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "api_internal.h"
#include "expr.h"

using namespace mq;
using namespace mq::api;

namespace mq {
namespace api {

namespace {

// device workspace of the family: [ lowered expressions | counters, tile table, descriptors ]
constexpr int64_t kExprArea = (sizeof(DevExprSet) + 255) & ~(int64_t)255;

int32_t proj_spec_of(const mi355q_plan& lp, int n_phys, const mi355q_qmd& q, ProjSpec* ps) {
  std::memset(ps, 0, sizeof(*ps));
  ps->n_targets = q.n_targets;
  ps->columnar = q.output_columnar;
  ps->row_quad = 1 + q.n_targets;
  ps->n_phys_cols = n_phys;
  ps->n_cols_table = n_phys;
  ps->entry_count = q.entry_count;
  for (int i = 0; i < q.n_targets; ++i) {
    const mi355q_target& t = lp.targets[i];
    const bool inner = t.table != 0;  // a column of the join's inner table, read through the matched row
    if (t.col < 0 || t.col >= (inner ? lp.n_inner_cols : lp.n_cols)) return MI355Q_ERR_INVALID_PLAN;
    ProjTarget& pt = ps->t[i];
    pt.col = inner ? kProjInnerCol + t.col : t.col;
    pt.code = col_type_code(inner ? lp.inner_cols[t.col] : lp.cols[t.col]);
    if (pt.code < 0) return MI355Q_ERR_INVALID_PLAN;
    const int st = tc_storage(pt.code);
    pt.kind = st == MI355Q_DOUBLE ? PROJ_F64 : st == MI355Q_FLOAT ? (q.output_columnar ? PROJ_F32 : PROJ_F32_TO_F64) : PROJ_INT;
    pt.width = q.slot_bytes[i];
    pt.col_off = q.output_columnar ? qmd_slot_col_offset(q, i) : 0;
  }
  return MI355Q_OK;
}

}  // namespace

int32_t execute_projection(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o, mi355q_result** out,
                           mi355q_exec_report* report, int64_t* reserved) {
  mi355q_plan lp;
  DevExprSet xs;
  std::memset(&xs, 0, sizeof(xs));
  const int n_phys = plan->n_cols;
  if (plan->n_exprs != 0) {
    if (int32_t e = lower_exprs(*plan, &lp, &xs)) return e;
  } else {
    lp = *plan;
  }
  mi355q_qmd q;
  if (int32_t e = qmd_init(lp, &q)) return e;
  if (q.desc_type != MI355Q_PROJECTION) return MI355Q_ERR_INVALID_PLAN;
  DevPlan d;
  if (int32_t e = build_dev_plan(lp, q, &d)) return e;
  if (int32_t e = attach_join(lp, in, &d)) return e;
  // one entry per joined row: a one-to-one table gives at most one per outer row, which the compaction's match bit states;
  // the matching SETS of a one-to-many table (HashJoin::codegenMatchingSet) would need a count per row
  // (round 6: the matching sets of a one-to-many table are entries too — k_proj_join_1n; with expressions the family still
  // answers "unsupported")
  if (d.join_col >= 0 && d.join_hash_type >= 2 && plan->n_exprs != 0) return MI355Q_ERR_UNSUPPORTED;
  for (int k = 0; k < d.n_quals; ++k)  // (a member of a disjunction is a plain column comparison: the binding never states one over an expression)
    if (d.quals[k].or_group != 0 && d.quals[k].col >= n_phys) return MI355Q_ERR_UNSUPPORTED;
  ProjSpec ps;
  if (int32_t e = proj_spec_of(lp, n_phys, q, &ps)) return e;
  ProjForms forms;
  forms.ok = 0;
  if (plan->n_exprs != 0) {  // the device copy: every node with its typed handler (physical columns are loaded where they are read)
    // (before the labelling rewrites the nodes: targets that are `[CAST](column) <op> literal` — the fast member's forms)
    projection_forms(xs, ps, expr_qual_mask(*plan), &forms);
    const int deepest = xh_label_programs(&xs, false);
    ps.x_info = deepest | (xs.n << 8);
  }
  const uint32_t qmask = plan->n_exprs ? expr_qual_mask(*plan) : 0u;

  const int nf = in->n_frags, nc = n_phys;
  int64_t total_rows = 0, max_frag_rows = 0;
  for (int f = 0; f < nf; ++f) {
    if (in->num_rows[f] < 0) return MI355Q_ERR_INVALID_PLAN;
    total_rows += in->num_rows[f];
    max_frag_rows = std::max(max_frag_rows, in->num_rows[f]);
  }
  const int64_t ws_need = kExprArea + projection_scratch_bytes(nf, in->num_rows);
  if (reserved) {  // mi355q_reserve_workspace / mi355q_explain: nothing is launched
    route_note(d.join_col >= 0 ? "k_proj_compact (join probe per row)" : plan->n_exprs ? "k_proj_compact (expressions in registers)" : "k_proj_compact");
    *reserved = ws_need;
    if (t_plan_only) return MI355Q_OK;
  }

  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  const int n_cus = o.tune_cus > 0 ? std::min(o.tune_cus, cu_count_of(in->device_id)) : cu_count_of(in->device_id);
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  if (ctx.projws_bytes < ws_need) {
    if (ctx.projws) (void)hipFree(ctx.projws);
    ctx.projws = nullptr;
    ctx.projws_bytes = 0;
    HIP_TRY(hipMalloc(&ctx.projws, (size_t)ws_need));
    ctx.projws_bytes = ws_need;
  }
  if (reserved) return MI355Q_OK;

  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  // column table | row counts | error word, one upload out of pinned memory
  const size_t ptr_bytes = sizeof(void*) * (size_t)std::max(1, nf * nc);
  const size_t rows_bytes = sizeof(int64_t) * (size_t)std::max(1, nf);
  const size_t meta_bytes = ptr_bytes + rows_bytes + 64;
  if (ctx.meta_bytes < meta_bytes) {
    if (ctx.meta) (void)hipFree(ctx.meta);
    ctx.meta = nullptr;
    ctx.meta_bytes = 0;
    if (ctx.h_meta) (void)hipHostFree(ctx.h_meta);
    ctx.h_meta = nullptr;
    ctx.h_ret_dev = nullptr;
    HIP_TRY(hipMalloc(&ctx.meta, meta_bytes * 2));
    HIP_TRY(hipHostMalloc((void**)&ctx.h_meta, meta_bytes * 2 + 64, hipHostMallocDefault));
    ctx.meta_bytes = meta_bytes * 2;
  }
  char* mp = (char*)ctx.meta;
  const int8_t* const* d_cols = (const int8_t* const*)mp;
  const int64_t* d_rows = (const int64_t*)(mp + ptr_bytes);
  int32_t* d_err = (int32_t*)(mp + ptr_bytes + rows_bytes);
  {
    char* hm = ctx.h_meta;
    if (nf > 0) {
      std::memcpy(hm, in->col_buffers, sizeof(void*) * (size_t)(nf * nc));
      std::memcpy(hm + ptr_bytes, in->num_rows, sizeof(int64_t) * (size_t)nf);
    }
    std::memset(hm + ptr_bytes + rows_bytes, 0, 64);
    const size_t lo = nf > 0 ? 0 : ptr_bytes + rows_bytes;
    ctx.meta_shadow.clear();  // (execute_impl's record of what `meta` holds)
    ctx.meta_err_clean = false;
    HIP_TRY(hipMemcpyAsync(mp + lo, hm + lo, ptr_bytes + rows_bytes + 64 - lo, hipMemcpyHostToDevice, s));
  }
  const DevExprSet* d_xs = nullptr;
  if (plan->n_exprs != 0) {
    HIP_TRY(hipMemcpyAsync(ctx.projws, &xs, sizeof(xs), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // (xs lives on this frame)
    d_xs = (const DevExprSet*)ctx.projws;
  }

  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { mi355q_result_free(r); }
  } rg{res};

  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  LaunchStats st;
  if (report) {
    while ((int)ctx.events.size() < 4) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      ctx.events.push_back(e);
    }
    ev_start = ctx.events[0];
    ev_stop = ctx.events[1];
    st.k_start = ctx.events[2];
    st.k_stop = ctx.events[3];
  }
  FragView fv{d_cols, d_rows, in->col_buffers, in->num_rows, nf, nc, total_rows, max_frag_rows};
  if (ev_start) HIP_TRY(hipEventRecord(ev_start, s));
  unsigned long long* d_total = nullptr;
  HIP_TRY(launch_projection(d, ps, d_xs, qmask, fv, (char*)ctx.projws + kExprArea, res->buf, d_err, &d_total, n_cus, s, &st, &forms));
  if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, s));
  // the error word and the match count come back together
  int64_t* h_ret = (int64_t*)(ctx.h_meta + ctx.meta_bytes);
  HIP_TRY(hipMemcpyAsync(h_ret, d_err, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_ret + 1, d_total, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const int32_t code = (int32_t)h_ret[0];
  const int64_t total = h_ret[1];
  if (report) {
    std::memset(report, 0, sizeof(*report));
    std::snprintf(report->kernel_name, sizeof(report->kernel_name), "%s", st.kernel_name);
    float ms = 0.f;
    if (nf > 0 && total_rows > 0 && hipEventElapsedTime(&ms, st.k_start, st.k_stop) == hipSuccess) report->kernel_ms = ms;
    if (hipEventElapsedTime(&ms, ev_start, ev_stop) == hipSuccess) report->total_ms = ms;
    report->n_launches = st.n_launches;
    report->variant = st.variant;
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
    report->spilled_rows = 0;
  }
  if (code) return code;
  if (total > q.entry_count && plan->scan_limit == 0) {
    // the row that found the buffer full: the reference's row function answers -pos; here the count the caller needs
    // (counts within 64 of INT32_MAX are reported as that bound: the codes next to INT32_MIN are the step executor's
    // internal ones — kNotTaken, kRetryNo* — and must not leave a route by accident)
    constexpr int64_t kMaxReported = (int64_t)INT32_MAX - 64;
    return -(int32_t)(total > kMaxReported ? kMaxReported : total);
  }
  res->total_matched = total;
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}

// ResultSet::append (ResultSet.cpp:307-335; Executor::resultsUnion -> get_merged_result, Execute.cpp:1642-1694): the
// results of a projection over several devices / kernels are not reduced but laid one behind the other — the entry
// count becomes the sum, iteration walks the first storage and then the appended ones.  A result here is ONE buffer, so
// `this` gets a new buffer of both entry counts: its own rows, then that's, then the EMPTY_KEY_64 tail.
int32_t projection_append(mi355q_result* a, const mi355q_result* b, hipStream_t s) {
  const mi355q_qmd& qa = a->qmd;
  const mi355q_qmd& qb = b->qmd;
  if (qa.desc_type != MI355Q_PROJECTION || qb.desc_type != MI355Q_PROJECTION || qa.output_columnar != qb.output_columnar ||
      qa.row_size != qb.row_size || qa.slot_count != qb.slot_count || qa.n_targets != qb.n_targets || a->device_id != b->device_id)
    return MI355Q_ERR_INVALID_PLAN;
  for (int j = 0; j < qa.slot_count && j < MI355Q_MAX_SLOTS; ++j)
    if (qa.slot_bytes[j] != qb.slot_bytes[j]) return MI355Q_ERR_INVALID_PLAN;
  for (int t = 0; t < qa.n_targets && t < MI355Q_MAX_TARGETS; ++t)
    if (qa.target_slot[t] != qb.target_slot[t] || qa.target_is_fp[t] != qb.target_is_fp[t] ||
        qa.target_arg_is_f32[t] != qb.target_arg_is_f32[t] || qa.target_null[t] != qb.target_null[t])
      return MI355Q_ERR_INVALID_PLAN;
  const int64_t na = projection_row_count(a), nb = projection_row_count(b);
  if (na < 0 || nb < 0) return MI355Q_ERR_HIP;
  // (a wrapped buffer may hold its rows anywhere: only results whose rows are known to sit at the front are appended)
  if ((a->total_matched < 0 && a->live_rows < 0) || (b->total_matched < 0 && b->live_rows < 0)) return MI355Q_ERR_UNSUPPORTED;
  mi355q_qmd nq = qa;
  nq.entry_count = qa.entry_count + qb.entry_count;
  if (nq.entry_count > (int64_t)INT32_MAX) return MI355Q_ERR_UNSUPPORTED;
  mi355q_result* tmp = nullptr;
  if (int32_t e = result_create_impl(&nq, a->device_id, nullptr, &tmp)) return e;
  struct TmpGuard {
    mi355q_result*& r;
    ~TmpGuard() { if (r) mi355q_result_free(r); }
  } tg{tmp};
  DeviceGuard g(a->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  char* dst = (char*)tmp->buf;
  const int64_t tail = nq.entry_count - (na + nb);
  if (!nq.output_columnar) {
    const int64_t rb = nq.row_size;
    if (na) HIP_TRY(hipMemcpyAsync(dst, a->buf, (size_t)(na * rb), hipMemcpyDeviceToDevice, s));
    if (nb) HIP_TRY(hipMemcpyAsync(dst + na * rb, b->buf, (size_t)(nb * rb), hipMemcpyDeviceToDevice, s));
    if (tail) {
      RowInit ri{};
      ri.row_quad = nq.row_size / 8;
      row_init_image(nq, ri.quad);
      HIP_TRY(launch_init_buffer((int64_t*)(dst + (na + nb) * rb), tail, ri, s));
    }
  } else {
    // the key column (8 bytes per entry), then every slot column at its logical width
    if (na) HIP_TRY(hipMemcpyAsync(dst, a->buf, (size_t)(na * 8), hipMemcpyDeviceToDevice, s));
    if (nb) HIP_TRY(hipMemcpyAsync(dst + na * 8, b->buf, (size_t)(nb * 8), hipMemcpyDeviceToDevice, s));
    if (tail) {
      RowInit ri{};
      ri.row_quad = 1;
      ri.quad[0] = kEmptyKey64;
      HIP_TRY(launch_init_buffer((int64_t*)(dst + (na + nb) * 8), tail, ri, s));
    }
    for (int j = 0; j < nq.slot_count; ++j) {
      const int64_t w = nq.slot_bytes[j];
      char* col = dst + qmd_slot_col_offset(nq, j);
      if (na) HIP_TRY(hipMemcpyAsync(col, (const char*)a->buf + qmd_slot_col_offset(qa, j), (size_t)(na * w), hipMemcpyDeviceToDevice, s));
      if (nb) HIP_TRY(hipMemcpyAsync(col + na * w, (const char*)b->buf + qmd_slot_col_offset(qb, j), (size_t)(nb * w), hipMemcpyDeviceToDevice, s));
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (a->owns_buf && a->buf) (void)hipFree(a->buf);
  a->buf = tmp->buf;
  a->bytes = tmp->bytes;
  a->owns_buf = true;
  a->qmd = nq;
  a->dplan = tmp->dplan;
  tmp->buf = nullptr;  // (moved)
  tmp->owns_buf = false;
  a->live_rows = na + nb;
  // the rows that passed the quals on both sides, where both sides know (a scan_limit may have cut either output)
  const int64_t ta = a->total_matched >= 0 ? a->total_matched : na, tb = b->total_matched >= 0 ? b->total_matched : nb;
  a->total_matched = ta + tb;
  return MI355Q_OK;
}

int64_t projection_row_count(const mi355q_result* r) {
  const mi355q_qmd& q = r->qmd;
  if (r->live_rows >= 0) return r->live_rows;
  if (r->total_matched >= 0) return std::min(r->total_matched, q.entry_count);
  // a wrapped buffer: the entries whose key is not EMPTY_KEY_64 (ResultSet::isEmptyEntry)
  DeviceGuard g(r->device_id);
  DevWord cnt;
  if (hipMalloc(&cnt.p, 8) != hipSuccess) return -1;
  if (launch_projection_count_live(r->buf, q.output_columnar ? 1 : q.row_size / 8, q.entry_count, (unsigned long long*)cnt.p,
                                   nullptr) != hipSuccess)
    return -1;
  unsigned long long h = 0;
  if (hipMemcpy(&h, cnt.p, 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)h;
}

// ResultSet::getNextRow over Projection storage: the non-empty entries in entry order, one value per target —
// integers as stored (sign-extended; NULL = the type's inline sentinel), DOUBLE / FLOAT targets as doubles
int32_t projection_fetch_rows(const mi355q_result* r, int64_t max_rows, int64_t* ival, double* dval, int8_t* is_null,
                              int64_t* n_rows) {
  const mi355q_qmd& q = r->qmd;
  const int64_t live = projection_row_count(r);
  if (live < 0) return MI355Q_ERR_HIP;
  const int64_t n = std::min(live, max_rows);
  *n_rows = n;
  if (n == 0) return MI355Q_OK;
  DeviceGuard g(r->device_id);
  const int nt = q.n_targets;
  std::vector<char> host;
  auto value_out = [&](int64_t row, int t, int64_t v) {
    const size_t o = (size_t)row * nt + t;
    ival[o] = 0;
    dval[o] = 0.0;
    const bool nullable = q.target_null[t] != kEmptyKey64;  // (a NOT NULL target is never NULL, whatever its bits)
    if (q.target_arg_is_f32[t]) {
      dval[o] = (double)bits_flt((int32_t)v);
      is_null[o] = nullable && (int32_t)v == (int32_t)q.target_null[t];
    } else if (q.target_is_fp[t]) {
      dval[o] = bits_dbl(v);
      is_null[o] = nullable && v == q.target_null[t];
    } else {
      ival[o] = v;
      is_null[o] = nullable && v == q.target_null[t];
    }
  };
  // a WRAPPED buffer (no match count, no live count: the caller's bytes) need not keep its rows at the front: the entries
  // whose key is not EMPTY_KEY_64, in entry order (ResultSet::isEmptyEntry).  Otherwise the first n entries.
  const bool wrapped = r->live_rows < 0 && r->total_matched < 0;
  const int64_t n_scan = wrapped ? q.entry_count : n;
  try {
    std::vector<int64_t> pick;  // wrapped: the entry of output row i
    if (!q.output_columnar) {
      const int rq = q.row_size / 8;
      host.resize((size_t)n_scan * rq * 8);
      HIP_TRY(hipMemcpy(host.data(), r->buf, host.size(), hipMemcpyDeviceToHost));
      const int64_t* rows = (const int64_t*)host.data();
      int64_t o = 0;
      for (int64_t e = 0; e < n_scan && o < n; ++e) {
        if (wrapped && rows[e * rq] == kEmptyKey64) continue;
        for (int t = 0; t < nt; ++t) value_out(o, t, rows[e * rq + 1 + q.target_slot[t]]);
        ++o;
      }
    } else {
      if (wrapped) {
        host.resize((size_t)n_scan * 8);
        HIP_TRY(hipMemcpy(host.data(), r->buf, host.size(), hipMemcpyDeviceToHost));
        const int64_t* keys = (const int64_t*)host.data();
        for (int64_t e = 0; e < n_scan && (int64_t)pick.size() < n; ++e)
          if (keys[e] != kEmptyKey64) pick.push_back(e);
      }
      for (int t = 0; t < nt; ++t) {
        const int w = q.slot_bytes[q.target_slot[t]];
        host.resize((size_t)n_scan * w);
        HIP_TRY(hipMemcpy(host.data(), (const char*)r->buf + qmd_slot_col_offset(q, q.target_slot[t]), host.size(),
                          hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) {
          const int64_t e = wrapped ? pick[(size_t)i] : i;
          int64_t v;
          switch (w) {
            case 1: v = ((const int8_t*)host.data())[e]; break;
            case 2: v = ((const int16_t*)host.data())[e]; break;
            case 4: v = ((const int32_t*)host.data())[e]; break;
            default: v = ((const int64_t*)host.data())[e];
          }
          value_out(i, t, v);
        }
      }
    }
  } catch (const std::bad_alloc&) {
    return MI355Q_ERR_OUT_OF_CPU_MEM;
  }
  return MI355Q_OK;
}

}  // namespace api
}  // namespace mq
