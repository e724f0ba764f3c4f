// Mi355qExecutor.cpp — binds libmi355q.so into HeavyDB as the GPU executor of one query step.
//
// This is the translation unit INTEGRATION.md section 2 describes, compiled here against
// integration/mock/heavydb_mock.h (-DMI355Q_GLUE_MOCK_HEADERS) because HeavyDB itself cannot be built in this
// environment; nothing below depends on the mock beyond the reference's own class and accessor names.  It is
// exercised by integration/glue_check.cpp (a native program: execution units built from Analyzer expressions, a
// ResultSetStorage-layout buffer filled through mi355q_result_copy_to_host, checked against the oracle).
//
// What it replaces inside ExecutionKernel::runImpl (QueryEngine/ExecutionKernel.cpp:186-481): everything between
// fetchChunks and the ResultSet — executePlanWithGroupBy / executePlanWithoutGroupBy (Execute.cpp:4179-4366),
// QueryExecutionContext::launchGpuCode (QueryExecutionContext.cpp:211-582) and the JIT behind them.
#include "Mi355qExecutor.h"

#include <stdexcept>

namespace mi355q_glue {

namespace {

void check(int32_t code) {
  if (code) throw QueryExecutionError(code);  // heavyai::ErrorCode values; < 0 = out of group slots
}

// (storage_type / logical_type / agg_kind / emit_expr / translate_qual / translate_agg: Mi355qTranslate.h — the half of
// this binding that compiles against the reference's own headers in this image)

mi355q_range to_range(const ExpressionRange& r) {
  mi355q_range out{};
  out.valid = r.getType() != ExpressionRangeType::Invalid;
  out.has_nulls = r.hasNulls();
  if (r.getType() == ExpressionRangeType::Integer) {
    out.min = r.getIntMin();
    out.max = r.getIntMax();
    out.bucket = r.getBucket();
  } else if (out.valid) {
    out.fp_min = r.getFpMin();
    out.fp_max = r.getFpMax();
  }
  return out;
}

struct Translator {
  const RelAlgExecutionUnit& ra;
  const std::vector<InputTableInfo>& query_infos;
  const Executor* executor;
  mi355q_plan& p;
  // (table_id, column_id) of every outer / inner input column, in input_col_descs order = FetchResult column order
  std::vector<shared::ColumnKey> outer_cols, inner_cols;
  std::vector<const Analyzer::Expr*> expr_of;  // expression k of the plan (structural identity by pointer)

  int find(const std::vector<shared::ColumnKey>& v, const shared::ColumnKey& k) const {
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i].db_id == k.db_id && v[i].table_id == k.table_id && v[i].column_id == k.column_id) return (int)i;
    unsupported("column is not among input_col_descs");
  }

  void emit(const Analyzer::Expr* e, mi355q_expr& x) {
    emit_expr(e, x, [this](const Analyzer::ColumnVar* cv) { return find(outer_cols, cv->getColumnKey()); });
  }

  // outer column index of a value expression: a plain column, or the virtual column of a projected expression
  int value_col(const Analyzer::Expr* e) {
    if (auto cv = dynamic_cast<const Analyzer::ColumnVar*>(e)) {
      if (cv->get_rte_idx() == 0) return find(outer_cols, cv->getColumnKey());
      unsupported("inner column where an outer value is expected");
    }
    for (size_t k = 0; k < expr_of.size(); ++k)
      if (expr_of[k] == e) return p.n_cols + (int)k;
    mi355q_expr x{};
    try {
      emit(e, x);
    } catch (const ExprTooLong&) {
      // (the operands first: an expression reads EARLIER ones)
      const OuterCol oc = [this](const Analyzer::ColumnVar* cv) { return find(outer_cols, cv->getColumnKey()); };
      if (!split_plain_logic(e, x, oc, [this](const Analyzer::Expr* v) { return value_col(v); }) &&
          !split_in_list(e, x, oc, [this](const ExprFiller& fill) { return new_bool_col(fill); })) {
        // ... or a CASE whose conditions cannot raise: those are evaluated ahead, in expressions of their own
        std::vector<const Analyzer::Expr*> conds;
        hoistable_conditions(e, conds);
        if (conds.empty()) throw;
        std::vector<std::pair<const Analyzer::Expr*, int>> at;
        for (const Analyzer::Expr* c : conds) at.emplace_back(c, value_col(c));
        const std::function<int(const Analyzer::Expr*)> hoisted = [&at](const Analyzer::Expr* n) {
          for (const auto& pr : at)
            if (pr.first == n) return pr.second;
          return -1;
        };
        x = mi355q_expr{};
        emit_expr(e, x, oc, &hoisted);
      }
    }
    if (p.n_exprs >= MI355Q_MAX_EXPRS) unsupported("too many projected expressions");
    check_expr_stack(x);
    x.range = to_range(getExpressionRange(e, query_infos, executor));
    p.exprs[p.n_exprs] = x;
    expr_of.push_back(e);
    return p.n_cols + p.n_exprs++;
  }

  // a BOOLEAN expression column that no single Analyzer node stands for (translate_where's guarded conjunction)
  int new_bool_col(const ExprFiller& fill) {
    if (p.n_exprs >= MI355Q_MAX_EXPRS) unsupported("too many projected expressions");
    mi355q_expr& x = p.exprs[p.n_exprs];
    x = mi355q_expr{};
    fill(x, [this](const Analyzer::ColumnVar* cv) { return find(outer_cols, cv->getColumnKey()); });
    check_expr_stack(x);
    x.range = mi355q_range{1, 1, 0, 1, 0.0, 0.0, 0};
    expr_of.push_back(nullptr);
    return p.n_cols + p.n_exprs++;
  }

  // simple_quals / quals entry: <value> <cmp> <literal>, <value> IS NULL, NOT(<value> IS NULL)
  mi355q_qual qual(const Analyzer::Expr* e) {
    return translate_qual(e, [this](const Analyzer::Expr* v) { return value_col(v); });
  }
};

}  // namespace

mi355q_plan to_plan(const RelAlgExecutionUnit& ra, const std::vector<InputTableInfo>& query_infos,
                    const Executor* executor, const mi355q_join_table* join_table,
                    size_t max_groups_buffer_entry_guess, bool output_columnar_hint) {
  mi355q_plan p{};
  p.abi_version = MI355Q_ABI_VERSION;
  Translator t{ra, query_infos, executor, p, {}, {}, {}};
  // input_col_descs -> cols[] / inner_cols[] (+ the ExpressionRange chunk metadata gives each column)
  for (const auto& icd : ra.input_col_descs) {
    const bool inner = icd->getScanDesc().getNestLevel() != 0;
    const auto& tk = icd->getScanDesc().getTableKey();
    const SQLTypeInfo ti = get_column_type(icd->getColId(), tk, executor);
    mi355q_col_desc cd{};
    cd.type = storage_type(ti);
    cd.nullable = !ti.get_notnull();
    cd.encoding = encoding_of(ti);
    if (cd.encoding == MI355Q_ENC_FIXED) cd.logical_type = logical_type(ti);
    Analyzer::ColumnVar cv(ti, shared::ColumnKey{tk.db_id, tk.table_id, icd->getColId()}, inner ? 1 : 0);
    const mi355q_range r = to_range(getExpressionRange(&cv, query_infos, executor));
    int32_t& n = inner ? p.n_inner_cols : p.n_cols;
    if (n >= MI355Q_MAX_COLS) unsupported("too many input columns");
    (inner ? p.inner_cols : p.cols)[n] = cd;
    (inner ? p.inner_col_ranges : p.col_ranges)[n] = r;
    (inner ? t.inner_cols : t.outer_cols).push_back(shared::ColumnKey{tk.db_id, tk.table_id, icd->getColId()});
    ++n;
  }
  // groupby_exprs: {nullptr} = non-grouped; a ColumnVar or a projected expression each
  std::vector<const Analyzer::Expr*> group_exprs;
  for (const auto& g : ra.groupby_exprs) {
    if (!g) continue;
    if (p.n_group_cols >= MI355Q_MAX_GROUP_COLS) unsupported("too many group-by expressions");
    group_exprs.push_back(g.get());
    p.group_cols[p.n_group_cols++] = t.value_col(g.get());
  }
  // simple_quals and quals: a conjunction
  int32_t n_or_groups = 0;
  std::vector<const Analyzer::Expr*> where;
  for (const auto* lst : {&ra.simple_quals, &ra.quals})
    for (const auto& q : *lst) where.push_back(q.get());
  translate_where(where, [&t](const Analyzer::Expr* v) { return t.value_col(v); },
                  [&t](const ExprFiller& fill) { return t.new_bool_col(fill); }, p.quals, &p.n_quals, &n_or_groups);
  // A Projection (QueryDescriptionType::Projection): groupby_exprs is {nullptr} and no target is an aggregate — every target
  // is then a row-wise value (a column or an expression, TargetInfo.is_agg == false) and the step emits one entry per
  // row that passes the quals, scan_limit of them at most (GroupByAndAggregate::initQueryMemoryDescriptor :870-905 picks the
  // description type the same way: `group_col_widths.empty() ... is_agg` -> NonGroupedAggregate, else Projection).
  bool projection = group_exprs.empty() && !ra.target_exprs.empty();
  for (const auto* te : ra.target_exprs) projection = projection && !dynamic_cast<const Analyzer::AggExpr*>(te);
  // (a projection through a join: one entry per joined row; an inner table's column is a target of table 1, read through
  // the matched row — one-to-one hash tables; the library answers "unsupported" for a one-to-many table)
  p.scan_limit = projection ? (int64_t)ra.scan_limit : 0;
  // target_exprs (get_target_info, Shared/TargetInfo.h:48-56): aggregates, or projections of a group key
  for (const auto* te : ra.target_exprs) {
    if (p.n_targets >= MI355Q_MAX_TARGETS) unsupported("too many targets");
    mi355q_target tg{};
    tg.col = -1;
    if (auto agg = dynamic_cast<const Analyzer::AggExpr*>(te)) {
      tg = translate_agg(agg, [&t](const Analyzer::Expr* v) { return t.value_col(v); },
                         [&t](const Analyzer::ColumnVar* cv) { return t.find(t.inner_cols, cv->getColumnKey()); });
    } else if (projection) {
      tg.agg = MI355Q_PROJECT;
      auto cv = dynamic_cast<const Analyzer::ColumnVar*>(te);
      if (cv && cv->get_rte_idx() != 0) {  // a column of the join's inner table
        tg.table = 1;
        tg.col = t.find(t.inner_cols, cv->getColumnKey());
        if (tg.col < 0) unsupported("a projected inner column that is not among the inputs");
      } else {
        tg.col = t.value_col(te);
      }
    } else {
      tg.agg = MI355Q_PROJECT_KEY;
      int idx = -1;
      for (size_t g = 0; g < group_exprs.size(); ++g) {
        auto a = dynamic_cast<const Analyzer::ColumnVar*>(te);
        auto b = dynamic_cast<const Analyzer::ColumnVar*>(group_exprs[g]);
        if (group_exprs[g] == te || (a && b && a->getColumnKey().table_id == b->getColumnKey().table_id &&
                                     a->getColumnKey().column_id == b->getColumnKey().column_id))
          idx = (int)g;
      }
      if (idx < 0) unsupported("projection of something that is not a group key");
      tg.col = idx;
    }
    p.targets[p.n_targets++] = tg;
  }
  // join_quals: one equi-join level; every `outer.col = inner.key` conjunct is one key component
  // (HashJoin::normalizeColumnPairs)
  p.join_outer_col = -1;
  if (!ra.join_quals.empty()) {
    if (ra.join_quals.size() > 1) unsupported("more than one join level");
    const JoinCondition& jc = ra.join_quals.front();
    int n = 0;
    for (const auto& q : jc.quals) {
      auto b = dynamic_cast<const Analyzer::BinOper*>(q.get());
      auto l = b ? dynamic_cast<const Analyzer::ColumnVar*>(b->get_left_operand()) : nullptr;
      auto r = b ? dynamic_cast<const Analyzer::ColumnVar*>(b->get_right_operand()) : nullptr;
      if (!b || b->get_optype() != kEQ || !l || !r) unsupported("join qual shape");
      const Analyzer::ColumnVar* outer = l->get_rte_idx() == 0 ? l : r;
      if (n >= MI355Q_MAX_GROUP_COLS) unsupported("too many join key components");
      p.join_outer_cols[n++] = t.find(t.outer_cols, outer->getColumnKey());
    }
    p.join_outer_col = p.join_outer_cols[0];
    p.n_join_cols = n > 1 ? n : 0;
    p.join_kind = jc.type == JoinType::LEFT ? MI355Q_JOIN_LEFT : MI355Q_JOIN_INNER;
  }
  p.join_table = join_table;
  p.max_groups_buffer_entry_guess = (int64_t)max_groups_buffer_entry_guess;
  p.bigint_count = g_bigint_count;
  for (const auto& qi : query_infos) p.num_tuples += qi.num_tuples;  // 4-byte slots decision
  p.output_columnar_hint = output_columnar_hint ? MI355Q_OUTPUT_COLUMNAR : MI355Q_OUTPUT_ROWWISE;
  return p;
}

std::string explain_query_mi355q(const RelAlgExecutionUnit& ra, const std::vector<InputTableInfo>& query_infos,
                                 const Executor* executor, int device_id, size_t max_groups_buffer_entry_guess,
                                 const mi355q_join_table* join_table, int64_t inner_num_rows, bool output_columnar_hint) {
  const mi355q_plan plan = to_plan(ra, query_infos, executor, join_table, max_groups_buffer_entry_guess, output_columnar_hint);
  // only the SHAPE of the input decides the route: one entry per fragment of the outer table, no chunk pointers
  std::vector<int64_t> rows;
  if (!query_infos.empty()) {
#ifdef MI355Q_GLUE_MOCK_HEADERS
    rows = query_infos.front().fragment_rows;
    if (rows.empty()) rows.push_back(query_infos.front().num_tuples);
#else
    for (const auto& frag : query_infos.front().info.fragments) rows.push_back((int64_t)frag.getNumTuples());
#endif
  }
  mi355q_inputs in{};
  in.device_id = device_id;
  in.n_frags = (int32_t)rows.size();
  in.num_rows = rows.data();
  in.inner_num_rows = inner_num_rows;
  char route[512] = {0};
  int64_t scratch = 0;
  check(mi355q_explain(&plan, &in, nullptr, route, (int64_t)sizeof(route), &scratch));
  return std::string("mi355q route: ") + route + " (partition scratch " + std::to_string(scratch >> 20) + " MiB)";
}

ResultSetPtr run_query_mi355q(const RelAlgExecutionUnit& ra, const FetchResult& fetch_result,
                              const std::vector<InputTableInfo>& query_infos,
                              const QueryMemoryDescriptor& query_mem_desc, Executor* executor, int device_id,
                              size_t max_groups_buffer_entry_guess, const mi355q_join_table* join_table,
                              const std::vector<const int8_t*>& inner_col_buffers, int64_t inner_num_rows) {
  const mi355q_plan plan = to_plan(ra, query_infos, executor, join_table, max_groups_buffer_entry_guess,
                                   query_mem_desc.didOutputColumnar());
  // FetchResult -> mi355q_inputs: col_buffers[frag][col] are GPU_LEVEL chunk pointers, outer columns first
  std::vector<const void*> bufs;
  std::vector<int64_t> rows;
  for (size_t f = 0; f < fetch_result.col_buffers.size(); ++f) {
    CHECK(fetch_result.col_buffers[f].size() >= (size_t)plan.n_cols);
    for (int c = 0; c < plan.n_cols; ++c) bufs.push_back(fetch_result.col_buffers[f][c]);
    rows.push_back(fetch_result.fragment_info.num_rows[f][0]);
  }
  std::vector<const void*> inner(inner_col_buffers.begin(), inner_col_buffers.end());
  mi355q_inputs in{};
  in.device_id = device_id;
  in.n_frags = (int32_t)rows.size();
  in.col_buffers = bufs.data();
  in.num_rows = rows.data();
  in.inner_col_buffers = inner.data();
  in.inner_num_rows = inner_num_rows;

  // Layout self-check: the library derives the same QueryMemoryDescriptor from the plan as HeavyDB did.
  mi355q_qmd qmd;
  check(mi355q_qmd_init(&plan, &qmd));
  CHECK_EQ(qmd.entry_count, (int64_t)query_mem_desc.getEntryCount());
  CHECK_EQ(qmd.row_size, (int32_t)query_mem_desc.getRowSize());
  if (qmd.desc_type != MI355Q_PROJECTION) CHECK_EQ(qmd.slot_width, (int32_t)query_mem_desc.getCompactByteWidth());
  CHECK_EQ(qmd.keyless != 0, query_mem_desc.hasKeylessHash());
  CHECK_EQ(qmd.output_columnar != 0, query_mem_desc.didOutputColumnar());
  CHECK_EQ(mi355q_qmd_buffer_bytes(&qmd), (int64_t)query_mem_desc.getBufferSizeBytes(ExecutorDeviceType::GPU));

  mi355q_exec_options opts{};
  opts.stream = executor->getCudaStream(device_id);  // the per-device stream (Execute.cpp:5648); NULL = the library's
  mi355q_result* res = nullptr;
  mi355q_exec_report rep;
  check(mi355q_execute(&plan, &in, &opts, &res, &rep));  // same numeric codes as heavyai::ErrorCode; < 0 = out of
                                                         // slots -> RelAlgExecutor doubles the guess and retries
                                                         // (RelAlgExecutor.cpp:4143-4231)
  // ResultSet over the returned storage: allocateStorage() + device-to-host copy of the buffer, exactly what
  // launchGpuCode does with the group-by buffer (QueryExecutionContext.cpp:519-560)
  std::vector<TargetInfo> targets;
  for (const auto* te : ra.target_exprs) {
    auto agg = dynamic_cast<const Analyzer::AggExpr*>(te);
    targets.push_back(TargetInfo{agg != nullptr, agg ? agg->get_aggtype() : kMIN, te->get_type_info(),
                                 agg && agg->get_arg() ? agg->get_arg()->get_type_info() : SQLTypeInfo(),
                                 qmd.target_skip_null[targets.size()] != 0, false});
  }
  auto rs = std::make_shared<ResultSet>(targets, ExecutorDeviceType::GPU, query_mem_desc, executor->getRowSetMemoryOwner(), 0, 0);
  auto* storage = rs->allocateStorage();
  const int32_t rc = mi355q_result_copy_to_host(res, storage->getUnderlyingBuffer(),
                                                (int64_t)query_mem_desc.getBufferSizeBytes(ExecutorDeviceType::GPU));
  mi355q_result_free(res);
  check(rc);
  return rs;
}

}  // namespace mi355q_glue
