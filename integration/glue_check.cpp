// glue_check.cpp — native driver of the HeavyDB binding (integration/Mi355qExecutor.cpp): no Python.
//
// For each query an execution unit is built the way RelAlgExecutor would hand it over — Analyzer::ColumnVar /
// Constant / UOper(kCAST) / BinOper / AggExpr objects in a RelAlgExecutionUnit, GPU_LEVEL chunk pointers in a
// FetchResult, a QueryMemoryDescriptor for the layout self-check — run_query_mi355q runs the step through the
// C-ABI and copies the table into ResultSet::allocateStorage()'s buffer with mi355q_result_copy_to_host, and that
// ResultSetStorage-layout buffer is compared with what the ORACLE (oracle/liboracle.so, the CPU restatement of the
// reference's executor) produces from a HAND-WRITTEN mi355q_plan of the same query over the same rows.  The plan
// the binding derives is also compared with the hand-written one field by field, so a mistranslation cannot hide
// behind both sides sharing it.  Exit code 0 = every query agreed.
//
//   integration/build.sh && integration/glue_check
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "Mi355qExecutor.h"

extern "C" {
int32_t orc_qmd_init(const mi355q_plan* plan, mi355q_qmd* out);
int64_t orc_buffer_bytes(const mi355q_qmd* q);
int32_t orc_execute(const mi355q_plan* plan, const mi355q_inputs* in, const void* join, int32_t n_threads,
                    int64_t* out_buf, mi355q_qmd* out_qmd);
void* orc_join_build(const void* key_col, int key_type, int key_nullable, int64_t num_rows, int64_t min_key,
                     int64_t max_key, int prefer_baseline, int64_t max_perfect_entries, int32_t* err);
void* orc_join_build_n(const void* const* key_cols, const int32_t* key_types, const int32_t* key_nullables, int32_t n_keys, int64_t num_rows,
                       int64_t min_key, int64_t max_key, int prefer_baseline, int64_t max_perfect_entries, int32_t one_to_many,
                       int64_t keyed_entries, int32_t* err);
void orc_join_free(void* j);
}

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)
#define MQCK(x) do { int32_t c_ = (x); if (c_) { std::printf("mi355q error %d at line %d\n", c_, __LINE__); std::exit(2); } } while (0)

namespace {

constexpr int kTable = 1, kDim = 2, kDb = 1;
int g_failures = 0;
void expect(bool ok, const char* what) {
  if (!ok) {
    std::printf("  FAILED: %s\n", what);
    ++g_failures;
  }
}

struct Column {
  SQLTypeInfo ti;
  int gen_kind;
  uint64_t seed;
  int64_t a, b, c;
  double a_f;
  void* dev = nullptr;
  std::vector<int8_t> host;
};

// a fact table generated on the device (mi355q_generate_column), mirrored to the host for the oracle
struct Table {
  int64_t n_rows = 0;
  std::vector<Column> cols;
  void generate() {
    for (Column& c : cols) {
      const size_t bytes = (size_t)n_rows * c.ti.get_size();
      HIPCK(hipMalloc(&c.dev, bytes));
      MQCK(mi355q_generate_column(0, c.dev, n_rows, 0, c.gen_kind, c.seed, c.a, c.b, c.c, c.a_f, 0, nullptr));
      HIPCK(hipDeviceSynchronize());
      c.host.resize(bytes);
      HIPCK(hipMemcpy(c.host.data(), c.dev, bytes, hipMemcpyDeviceToHost));
    }
  }
};

std::shared_ptr<Analyzer::ColumnVar> colvar(const Table& t, int table_id, int col, int rte = 0) {
  return std::make_shared<Analyzer::ColumnVar>(t.cols[col].ti, shared::ColumnKey{kDb, table_id, col}, rte);
}
std::shared_ptr<Analyzer::Constant> int_lit(SQLTypes t, int64_t v) {
  Datum d{};
  if (t == kINT) d.intval = (int32_t)v; else d.bigintval = v;
  return std::make_shared<Analyzer::Constant>(SQLTypeInfo(t, true), false, d);
}

std::shared_ptr<Analyzer::Constant> dbl_lit(double v) {
  Datum d{};
  d.doubleval = v;
  return std::make_shared<Analyzer::Constant>(SQLTypeInfo(kDOUBLE, true), false, d);
}

// two fragments per column (16-byte aligned split), as fetchChunks would hand them over
FetchResult fetch(const Table& t, const std::vector<int>& cols, std::vector<int64_t>* frag_rows) {
  const int64_t cut = (t.n_rows / 3) / 4 * 4;
  FetchResult fr;
  for (int f = 0; f < 2; ++f) {
    std::vector<const int8_t*> ptrs;
    for (int c : cols) ptrs.push_back((const int8_t*)t.cols[c].dev + (f ? cut * t.cols[c].ti.get_size() : 0));
    fr.col_buffers.push_back(ptrs);
    fr.fragment_info.num_rows.push_back({f ? t.n_rows - cut : cut});
    frag_rows->push_back(f ? t.n_rows - cut : cut);
  }
  return fr;
}

// the oracle's table of a plan over the host mirror of the same columns
std::vector<int64_t> oracle_table(const mi355q_plan& plan, const Table& t, const std::vector<int>& cols,
                                  const std::vector<int64_t>& frag_rows, const void* join, const std::vector<const void*>& inner,
                                  int64_t inner_rows, mi355q_qmd* q) {
  MQCK(orc_qmd_init(&plan, q));
  std::vector<int64_t> buf((size_t)orc_buffer_bytes(q) / 8);
  std::vector<const void*> ptrs;
  int64_t off = 0;
  for (size_t f = 0; f < frag_rows.size(); ++f) {
    for (int c : cols) ptrs.push_back(t.cols[c].host.data() + off * t.cols[c].ti.get_size());
    off += frag_rows[f];
  }
  mi355q_inputs in{};
  in.device_id = -1;
  in.n_frags = (int32_t)frag_rows.size();
  in.col_buffers = ptrs.data();
  in.num_rows = frag_rows.data();
  in.inner_col_buffers = inner.data();
  in.inner_num_rows = inner_rows;
  mi355q_qmd oq;
  MQCK(orc_execute(&plan, &in, join, 8, buf.data(), &oq));
  return buf;
}

QueryMemoryDescriptor qmd_of(const mi355q_qmd& q) {  // what HeavyDB's own initQueryMemoryDescriptor would say
  QueryMemoryDescriptor d;
  d.entry_count = (size_t)q.entry_count;
  d.row_size = (size_t)q.row_size;
  d.compact_byte_width = (int8_t)q.slot_width;
  d.keyless = q.keyless != 0;
  d.columnar = q.output_columnar != 0;
  d.buffer_bytes = (size_t)orc_buffer_bytes(&q);
  return d;
}

// rows of a (row-wise, 8-byte slot) table as key -> slots, baseline tables place keys in insertion order
void compare_tables(const mi355q_qmd& q, const int64_t* want, const int64_t* got, const std::vector<bool>& slot_is_fp) {
  const int rq = q.row_size / 8, kq = q.key_bytes / 8;
  if (q.desc_type != MI355Q_GROUP_BY_BASELINE_HASH) {  // index-aligned layouts: quad by quad
    for (int64_t e = 0; e < q.entry_count; ++e)
      for (int j = 0; j < rq; ++j) {
        const int64_t a = want[e * rq + j], b = got[e * rq + j];
        if (a == b) continue;
        double x, y;
        std::memcpy(&x, &a, 8);
        std::memcpy(&y, &b, 8);
        const bool fp = j >= kq && slot_is_fp[j - kq];
        if (!fp || !(std::fabs(x - y) <= 1e-9 * std::max(std::fabs(x), std::fabs(y)))) {
          expect(false, "dense table differs");
          return;
        }
      }
    return;
  }
  std::map<std::vector<int64_t>, const int64_t*> w, g;
  for (int64_t e = 0; e < q.entry_count; ++e) {
    if (want[e * rq] != INT64_MAX) w[std::vector<int64_t>(want + e * rq, want + e * rq + kq)] = want + e * rq + kq;
    if (got[e * rq] != INT64_MAX) g[std::vector<int64_t>(got + e * rq, got + e * rq + kq)] = got + e * rq + kq;
  }
  expect(w.size() == g.size(), "group count differs");
  for (const auto& kv : w) {
    auto it = g.find(kv.first);
    if (it == g.end()) {
      expect(false, "a group is missing");
      return;
    }
    for (int j = 0; j < rq - kq; ++j) {
      const int64_t a = kv.second[j], b = it->second[j];
      if (a == b) continue;
      double x, y;
      std::memcpy(&x, &a, 8);
      std::memcpy(&y, &b, 8);
      if (!slot_is_fp[j] || !(std::fabs(x - y) <= 1e-9 * std::max(std::fabs(x), std::fabs(y)))) {
        expect(false, "a group's slots differ");
        return;
      }
    }
  }
}

// the parts of a plan a translation has to get right
void compare_plans(const mi355q_plan& a, const mi355q_plan& b) {
  expect(a.n_cols == b.n_cols && a.n_inner_cols == b.n_inner_cols && a.n_quals == b.n_quals &&
             a.n_group_cols == b.n_group_cols && a.n_targets == b.n_targets && a.n_exprs == b.n_exprs, "plan: counts");
  expect(!std::memcmp(a.cols, b.cols, sizeof(mi355q_col_desc) * a.n_cols), "plan: column descriptors");
  for (int i = 0; i < a.n_cols; ++i)
    expect(a.col_ranges[i].valid == b.col_ranges[i].valid && a.col_ranges[i].min == b.col_ranges[i].min &&
               a.col_ranges[i].max == b.col_ranges[i].max && a.col_ranges[i].has_nulls == b.col_ranges[i].has_nulls &&
               a.col_ranges[i].fp_min == b.col_ranges[i].fp_min && a.col_ranges[i].fp_max == b.col_ranges[i].fp_max, "plan: column range");
  for (int i = 0; i < a.n_quals; ++i)
    expect(a.quals[i].col == b.quals[i].col && a.quals[i].op == b.quals[i].op && a.quals[i].ival == b.quals[i].ival, "plan: qual");
  for (int i = 0; i < a.n_group_cols; ++i) expect(a.group_cols[i] == b.group_cols[i], "plan: group column");
  for (int i = 0; i < a.n_targets; ++i)
    expect(a.targets[i].agg == b.targets[i].agg && a.targets[i].col == b.targets[i].col && a.targets[i].table == b.targets[i].table, "plan: target");
  for (int k = 0; k < a.n_exprs; ++k) {
    expect(a.exprs[k].n_nodes == b.exprs[k].n_nodes, "plan: expression length");
    for (int i = 0; i < a.exprs[k].n_nodes; ++i) {
      const auto &x = a.exprs[k].nodes[i], &y = b.exprs[k].nodes[i];
      expect(x.op == y.op && x.arg == y.arg && x.ilit == y.ilit && x.flit == y.flit && (x.op == MI355Q_EX_COL || x.type == y.type), "plan: expression node");
    }
    const bool range_ok = a.exprs[k].range.min == b.exprs[k].range.min && a.exprs[k].range.max == b.exprs[k].range.max &&
                          a.exprs[k].range.fp_min == b.exprs[k].range.fp_min && a.exprs[k].range.fp_max == b.exprs[k].range.fp_max;
    if (!range_ok)
      std::printf("  expression %d range: by hand [%lld, %lld] [%g, %g], translated [%lld, %lld] [%g, %g]\n", k, (long long)a.exprs[k].range.min,
                  (long long)a.exprs[k].range.max, a.exprs[k].range.fp_min, a.exprs[k].range.fp_max, (long long)b.exprs[k].range.min,
                  (long long)b.exprs[k].range.max, b.exprs[k].range.fp_min, b.exprs[k].range.fp_max);
    expect(range_ok, "plan: expression range");
  }
  expect(a.join_outer_col == b.join_outer_col && a.join_kind == b.join_kind && a.n_join_cols == b.n_join_cols, "plan: join");
  expect(a.max_groups_buffer_entry_guess == b.max_groups_buffer_entry_guess && a.num_tuples == b.num_tuples && a.scan_limit == b.scan_limit,
         "plan: options");
}

}  // namespace

int main() {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
    std::printf("glue_check: no GPU\n");
    return 77;
  }
  Executor executor;
  const int64_t N = 6000000;
  const int64_t n_keys = 90000;
  // fact table: 0 key BIGINT NOT NULL (sparse), 1 f64 DOUBLE NOT NULL, 2 i32 INT NOT NULL, 3 x INT nullable-typed in [1, 40],
  //             4 fk BIGINT NOT NULL in [-50, M + 50)
  const int64_t M = 200000;
  Table t;
  t.n_rows = N;
  t.cols = {{SQLTypeInfo(kBIGINT, true), MI355Q_GEN_I64_MOD_MUL, 0xC0FFEE00, n_keys, 1000003, 7, 0.0},
            {SQLTypeInfo(kDOUBLE, true), MI355Q_GEN_F64_UNIT, 0xC0FFEE01, 0, 0, 0, 1000.0},
            {SQLTypeInfo(kINT, true), MI355Q_GEN_I32_UNIFORM31, 0xC0FFEE02, 0, 0, 0, 0.0},
            {SQLTypeInfo(kINT, false), MI355Q_GEN_I32_MOD, 0xC0FFEE03, 40, 1, 0, 0.0},
            {SQLTypeInfo(kBIGINT, true), MI355Q_GEN_I64_MOD, 0xC0FFEE04, M + 100, -50, 0, 0.0}};
  t.generate();
  const int64_t key_max = (n_keys - 1) * 1000003 + 7;
  executor.column_ranges[{kTable, 0}] = ExpressionRange::makeIntRange(7, key_max, 0, false);
  executor.column_ranges[{kTable, 1}] = ExpressionRange::makeDoubleRange(0.0, 1000.0, false);
  executor.column_ranges[{kTable, 2}] = ExpressionRange::makeIntRange(0, INT32_MAX, 0, false);
  executor.column_ranges[{kTable, 3}] = ExpressionRange::makeIntRange(1, 40, 0, false);
  executor.column_ranges[{kTable, 4}] = ExpressionRange::makeIntRange(-50, M + 49, 0, false);
  for (int c = 0; c < 5; ++c) executor.column_types[{kTable, c}] = t.cols[c].ti;
  const std::vector<InputTableInfo> query_infos = {{shared::TableKey{kDb, kTable}, N}};

  // ---------------------------------------------------------------- query 1: the headline shape
  //   SELECT key, COUNT(*), AVG(f64) FROM t WHERE i32 < 2^30 GROUP BY key
  {
    std::printf("query 1: SELECT key, COUNT(*), AVG(f64) FROM t WHERE i32 < 1073741824 GROUP BY key\n");
    RelAlgExecutionUnit ra;
    for (int c : {0, 1, 2}) ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(c, kTable, kDb, 0));
    auto key = colvar(t, kTable, 0), f64 = colvar(t, kTable, 1), i32 = colvar(t, kTable, 2);
    ra.simple_quals.push_back(std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, true), kLT, i32, int_lit(kINT, 1 << 30)));
    ra.groupby_exprs.push_back(key);
    Analyzer::AggExpr cnt(SQLTypeInfo(kBIGINT, true), kCOUNT, nullptr), avg(SQLTypeInfo(kDOUBLE, true), kAVG, f64);
    ra.target_exprs = {key.get(), &cnt, &avg};
    // the plan written by hand
    mi355q_plan hp{};
    hp.abi_version = MI355Q_ABI_VERSION;
    hp.n_cols = 3;
    hp.cols[0] = {MI355Q_INT64, 0, 0, 0};
    hp.cols[1] = {MI355Q_DOUBLE, 0, 0, 0};
    hp.cols[2] = {MI355Q_INT32, 0, 0, 0};
    hp.col_ranges[0] = {1, 0, 7, key_max, 0, 0, 0};
    hp.col_ranges[1] = {1, 0, 0, 0, 0.0, 1000.0, 0};
    hp.col_ranges[2] = {1, 0, 0, INT32_MAX, 0, 0, 0};
    hp.n_quals = 1;
    hp.quals[0] = {2, MI355Q_LT, 1 << 30, 0.0};
    hp.n_group_cols = 1;
    hp.group_cols[0] = 0;
    hp.n_targets = 3;
    hp.targets[0] = {MI355Q_PROJECT_KEY, 0, 0, 0, {}};
    hp.targets[1] = {MI355Q_COUNT, -1, 0, 0, {}};
    hp.targets[2] = {MI355Q_AVG, 1, 0, 0, {}};
    hp.join_outer_col = -1;
    hp.max_groups_buffer_entry_guess = 2 * n_keys;
    hp.num_tuples = N;
    compare_plans(hp, mi355q_glue::to_plan(ra, query_infos, &executor, nullptr, 2 * n_keys, false));
    std::vector<int64_t> frag_rows;
    const FetchResult fr = fetch(t, {0, 1, 2}, &frag_rows);
    mi355q_qmd q;
    const std::vector<int64_t> want = oracle_table(hp, t, {0, 1, 2}, frag_rows, nullptr, {}, 0, &q);
    // what just_explain would show for this step
    const std::string route = mi355q_glue::explain_query_mi355q(ra, query_infos, &executor, 0, 2 * n_keys, nullptr, 0, false);
    std::printf("  %s\n", route.c_str());
    expect(route.find("k_generic") == std::string::npos || N < (8 << 20), "the headline shape does not plan the row kernel");
    const ResultSetPtr rs = mi355q_glue::run_query_mi355q(ra, fr, query_infos, qmd_of(q), &executor, 0, 2 * n_keys, nullptr, {}, 0);
    expect(q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH && q.entry_count == 2 * n_keys && q.row_size == 32, "layout");
    compare_tables(q, want.data(), (const int64_t*)const_cast<ResultSetStorage*>(rs->getStorage())->getUnderlyingBuffer(), {false, true, false});
  }

  // ---------------------------------------------------------------- query 2: projected expressions (BH001 / MSBS001 shapes)
  //   SELECT cast(x AS DOUBLE), COUNT(*), MAX(x + 1), SUM(x + 1) FROM t GROUP BY cast(x AS DOUBLE)
  {
    std::printf("query 2: SELECT cast(x AS DOUBLE) k, COUNT(*), MAX(x + 1), SUM(x + 1) FROM t GROUP BY k\n");
    RelAlgExecutionUnit ra;
    ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(3, kTable, kDb, 0));
    auto x = colvar(t, kTable, 3);
    auto k = std::make_shared<Analyzer::UOper>(SQLTypeInfo(kDOUBLE, false), kCAST, x);
    auto x1 = std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kINT, false), kPLUS, x, int_lit(kINT, 1));
    ra.groupby_exprs.push_back(k);
    Analyzer::AggExpr cnt(SQLTypeInfo(kBIGINT, true), kCOUNT, nullptr), mx(SQLTypeInfo(kINT, false), kMAX, x1), sm(SQLTypeInfo(kBIGINT, false), kSUM, x1);
    ra.target_exprs = {k.get(), &cnt, &mx, &sm};
    mi355q_plan hp{};
    hp.abi_version = MI355Q_ABI_VERSION;
    hp.n_cols = 1;
    hp.cols[0] = {MI355Q_INT32, 1, 0, 0};
    hp.col_ranges[0] = {1, 0, 1, 40, 0, 0, 0};
    hp.n_exprs = 2;
    hp.exprs[0].n_nodes = 2;
    hp.exprs[0].nodes[0] = {MI355Q_EX_COL, 0, 0, 0, 0, 0.0};
    hp.exprs[0].nodes[1] = {MI355Q_EX_CAST, MI355Q_DOUBLE, 0, 0, 0, 0.0};
    hp.exprs[0].range = {1, 0, 0, 0, 1.0, 40.0, 0};
    hp.exprs[1].n_nodes = 3;
    hp.exprs[1].nodes[0] = {MI355Q_EX_COL, 0, 0, 0, 0, 0.0};
    hp.exprs[1].nodes[1] = {MI355Q_EX_LIT, MI355Q_INT32, 0, 0, 1, 0.0};
    hp.exprs[1].nodes[2] = {MI355Q_EX_ADD, MI355Q_INT32, 0, 0, 0, 0.0};
    hp.exprs[1].range = {1, 0, 2, 41, 0, 0, 0};
    hp.n_group_cols = 1;
    hp.group_cols[0] = 1;   // expression 0 = virtual column n_cols + 0
    hp.n_targets = 4;
    hp.targets[0] = {MI355Q_PROJECT_KEY, 0, 0, 0, {}};
    hp.targets[1] = {MI355Q_COUNT, -1, 0, 0, {}};
    hp.targets[2] = {MI355Q_MAX, 2, 0, 0, {}};
    hp.targets[3] = {MI355Q_SUM, 2, 0, 0, {}};
    hp.join_outer_col = -1;
    hp.max_groups_buffer_entry_guess = 16384;
    hp.num_tuples = N;
    compare_plans(hp, mi355q_glue::to_plan(ra, query_infos, &executor, nullptr, 16384, false));
    std::vector<int64_t> frag_rows;
    const FetchResult fr = fetch(t, {3}, &frag_rows);
    mi355q_qmd q;
    const std::vector<int64_t> want = oracle_table(hp, t, {3}, frag_rows, nullptr, {}, 0, &q);
    const ResultSetPtr rs = mi355q_glue::run_query_mi355q(ra, fr, query_infos, qmd_of(q), &executor, 0, 16384, nullptr, {}, 0);
    expect(q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH, "a floating-point key takes the baseline layout");
    compare_tables(q, want.data(), (const int64_t*)const_cast<ResultSetStorage*>(rs->getStorage())->getUnderlyingBuffer(), {false, false, false, false});
  }

  // ---------------------------------------------------------------- query 2b: a column-vs-column filter and a CASE argument
  //   SELECT x, COUNT(*), MAX(CASE WHEN i32 > x THEN x ELSE 0 END) FROM t WHERE i32 <> x GROUP BY x
  {
    std::printf("query 2b: SELECT x, COUNT(*), MAX(CASE WHEN i32 > x THEN x ELSE 0 END) FROM t WHERE i32 <> x GROUP BY x\n");
    RelAlgExecutionUnit ra;
    ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(2, kTable, kDb, 0));
    ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(3, kTable, kDb, 0));
    auto i32 = colvar(t, kTable, 2);
    auto x = colvar(t, kTable, 3);
    auto ne = std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, false), kNE, i32, x);
    auto gt = std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, false), kGT, i32, x);
    std::list<std::pair<std::shared_ptr<Analyzer::Expr>, std::shared_ptr<Analyzer::Expr>>> whens{{gt, x}};
    auto cs = std::make_shared<Analyzer::CaseExpr>(SQLTypeInfo(kINT, false), false, whens, int_lit(kINT, 0));
    ra.groupby_exprs.push_back(x);
    ra.quals.push_back(ne);
    Analyzer::AggExpr cnt(SQLTypeInfo(kBIGINT, true), kCOUNT, nullptr), mx(SQLTypeInfo(kINT, false), kMAX, cs);
    ra.target_exprs = {x.get(), &cnt, &mx};
    mi355q_plan hp{};
    hp.abi_version = MI355Q_ABI_VERSION;
    hp.n_cols = 2;
    hp.cols[0] = {MI355Q_INT32, 0, 0, 0};
    hp.col_ranges[0] = {1, 0, 0, INT32_MAX, 0, 0, 0};
    hp.cols[1] = {MI355Q_INT32, 1, 0, 0};
    hp.col_ranges[1] = {1, 0, 1, 40, 0, 0, 0};
    hp.n_exprs = 2;
    hp.exprs[0].n_nodes = 3;   // i32 <> x: a BOOLEAN (INT8 1 / 0 / NULL)
    hp.exprs[0].nodes[0] = {MI355Q_EX_COL, 0, 0, 0, 0, 0.0};
    hp.exprs[0].nodes[1] = {MI355Q_EX_COL, 0, 1, 0, 0, 0.0};
    hp.exprs[0].nodes[2] = {MI355Q_EX_NE, MI355Q_INT8, 0, 0, 0, 0.0};
    hp.exprs[1].n_nodes = 6;   // ELSE 0 | THEN x | i32 > x | CASE
    hp.exprs[1].nodes[0] = {MI355Q_EX_LIT, MI355Q_INT32, 0, 0, 0, 0.0};
    hp.exprs[1].nodes[1] = {MI355Q_EX_COL, 0, 1, 0, 0, 0.0};
    hp.exprs[1].nodes[2] = {MI355Q_EX_COL, 0, 0, 0, 0, 0.0};
    hp.exprs[1].nodes[3] = {MI355Q_EX_COL, 0, 1, 0, 0, 0.0};
    hp.exprs[1].nodes[4] = {MI355Q_EX_GT, MI355Q_INT8, 0, 0, 0, 0.0};
    hp.exprs[1].nodes[5] = {MI355Q_EX_CASE, MI355Q_INT32, 0, 0, 0, 0.0};
    hp.n_group_cols = 1;
    hp.group_cols[0] = 1;
    hp.n_quals = 1;
    hp.quals[0] = {2, MI355Q_EQ, 1, 0.0};   // the comparison column is TRUE
    hp.n_targets = 3;
    hp.targets[0] = {MI355Q_PROJECT_KEY, 0, 0, 0, {}};
    hp.targets[1] = {MI355Q_COUNT, -1, 0, 0, {}};
    hp.targets[2] = {MI355Q_MAX, 3, 0, 0, {}};
    hp.join_outer_col = -1;
    hp.max_groups_buffer_entry_guess = 16384;
    hp.num_tuples = N;
    compare_plans(hp, mi355q_glue::to_plan(ra, query_infos, &executor, nullptr, 16384, false));
    std::vector<int64_t> frag_rows;
    const FetchResult fr = fetch(t, {2, 3}, &frag_rows);
    mi355q_qmd q;
    const std::vector<int64_t> want = oracle_table(hp, t, {2, 3}, frag_rows, nullptr, {}, 0, &q);
    const ResultSetPtr rs = mi355q_glue::run_query_mi355q(ra, fr, query_infos, qmd_of(q), &executor, 0, 16384, nullptr, {}, 0);
    compare_tables(q, want.data(), (const int64_t*)const_cast<ResultSetStorage*>(rs->getStorage())->getUnderlyingBuffer(), {false, false, false});
  }

  // ---------------------------------------------------------------- query 3: hash-join probe + SUM
  //   SELECT SUM(t.i32w), SUM(d.w), COUNT(*) FROM t JOIN d ON t.fk = d.k
  {
    std::printf("query 3: SELECT SUM(t.key), SUM(d.w), COUNT(*) FROM t JOIN d ON t.fk = d.k\n");
    Table d;
    d.n_rows = M;
    d.cols = {{SQLTypeInfo(kBIGINT, true), MI355Q_GEN_I64_MOD, 0, 0, 0, 0, 0.0}, {SQLTypeInfo(kBIGINT, true), MI355Q_GEN_I64_MOD, 0xD1, 2001, -1000, 0, 0.0}};
    // keys 0 .. M-1 (dense, unique): written on the host, then uploaded
    d.cols[0].host.resize((size_t)M * 8);
    for (int64_t i = 0; i < M; ++i) ((int64_t*)d.cols[0].host.data())[i] = (i * 7919) % M;
    HIPCK(hipMalloc(&d.cols[0].dev, (size_t)M * 8));
    HIPCK(hipMemcpy(d.cols[0].dev, d.cols[0].host.data(), (size_t)M * 8, hipMemcpyHostToDevice));
    {
      Column& w = d.cols[1];
      HIPCK(hipMalloc(&w.dev, (size_t)M * 8));
      MQCK(mi355q_generate_column(0, w.dev, M, 0, w.gen_kind, w.seed, w.a, w.b, w.c, w.a_f, 0, nullptr));
      HIPCK(hipDeviceSynchronize());
      w.host.resize((size_t)M * 8);
      HIPCK(hipMemcpy(w.host.data(), w.dev, (size_t)M * 8, hipMemcpyDeviceToHost));
    }
    executor.column_types[{kDim, 0}] = d.cols[0].ti;
    executor.column_types[{kDim, 1}] = d.cols[1].ti;
    executor.column_ranges[{kDim, 0}] = ExpressionRange::makeIntRange(0, M - 1, 0, false);
    executor.column_ranges[{kDim, 1}] = ExpressionRange::makeIntRange(-1000, 1000, 0, false);
    mi355q_join_spec js{};
    js.device_id = 0;
    js.key_type = MI355Q_INT64;
    js.key_buffer = d.cols[0].dev;
    js.num_rows = M;
    js.key_range = {1, 0, 0, M - 1, 0, 0, 0};
    mi355q_join_table* jt = nullptr;
    MQCK(mi355q_join_build(&js, nullptr, &jt));
    int32_t oerr = 0;
    void* oj = orc_join_build(d.cols[0].host.data(), MI355Q_INT64, 0, M, 0, M - 1, 0, 0, &oerr);
    expect(oj != nullptr && oerr == 0, "oracle join build");
    RelAlgExecutionUnit ra;
    ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(4, kTable, kDb, 0));
    ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(0, kTable, kDb, 0));
    ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(0, kDim, kDb, 1));
    ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(1, kDim, kDb, 1));
    auto fk = colvar(t, kTable, 4), v = colvar(t, kTable, 0), dk = colvar(d, kDim, 0, 1), dw = colvar(d, kDim, 1, 1);
    JoinCondition jc;
    jc.type = JoinType::INNER;
    jc.quals.push_back(std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, true), kEQ, fk, dk));
    ra.join_quals.push_back(jc);
    ra.groupby_exprs.push_back(nullptr);
    Analyzer::AggExpr s1(SQLTypeInfo(kBIGINT, false), kSUM, v), s2(SQLTypeInfo(kBIGINT, false), kSUM, dw), cnt(SQLTypeInfo(kBIGINT, true), kCOUNT, nullptr);
    ra.target_exprs = {&s1, &s2, &cnt};
    mi355q_plan hp{};
    hp.abi_version = MI355Q_ABI_VERSION;
    hp.n_cols = 2;
    hp.cols[0] = {MI355Q_INT64, 0, 0, 0};
    hp.cols[1] = {MI355Q_INT64, 0, 0, 0};
    hp.col_ranges[0] = {1, 0, -50, M + 49, 0, 0, 0};
    hp.col_ranges[1] = {1, 0, 7, key_max, 0, 0, 0};
    hp.n_inner_cols = 2;
    hp.inner_cols[0] = {MI355Q_INT64, 0, 0, 0};
    hp.inner_cols[1] = {MI355Q_INT64, 0, 0, 0};
    hp.inner_col_ranges[0] = {1, 0, 0, M - 1, 0, 0, 0};
    hp.inner_col_ranges[1] = {1, 0, -1000, 1000, 0, 0, 0};
    hp.n_targets = 3;
    hp.targets[0] = {MI355Q_SUM, 1, 0, 0, {}};
    hp.targets[1] = {MI355Q_SUM, 1, 1, 0, {}};
    hp.targets[2] = {MI355Q_COUNT, -1, 0, 0, {}};
    hp.join_outer_col = 0;
    hp.join_outer_cols[0] = 0;
    hp.join_kind = MI355Q_JOIN_INNER;
    hp.max_groups_buffer_entry_guess = 16384;
    hp.num_tuples = N + M;
    const std::vector<InputTableInfo> qi2 = {{shared::TableKey{kDb, kTable}, N}, {shared::TableKey{kDb, kDim}, M}};
    compare_plans(hp, mi355q_glue::to_plan(ra, qi2, &executor, jt, 16384, false));
    std::vector<int64_t> frag_rows;
    const FetchResult fr = fetch(t, {4, 0}, &frag_rows);
    mi355q_qmd q;
    const std::vector<int64_t> want = oracle_table(hp, t, {4, 0}, frag_rows, oj, {d.cols[0].host.data(), d.cols[1].host.data()}, M, &q);
    const ResultSetPtr rs = mi355q_glue::run_query_mi355q(ra, fr, qi2, qmd_of(q), &executor, 0, 16384, jt,
                                                          {(const int8_t*)d.cols[0].dev, (const int8_t*)d.cols[1].dev}, M);
    expect(q.desc_type == MI355Q_NON_GROUPED_AGGREGATE, "layout");
    compare_tables(q, want.data(), (const int64_t*)const_cast<ResultSetStorage*>(rs->getStorage())->getUnderlyingBuffer(), {false, false, false});
    std::printf("  SUM(t.key) = %lld, SUM(d.w) = %lld, COUNT(*) = %lld\n", (long long)want[0], (long long)want[1], (long long)want[2]);
    // ------------------------------------------------------------ query 6: a Projection through the same join
    //   SELECT t.key, d.w FROM t LEFT JOIN d ON t.fk = d.k WHERE t.i32 < 2^27   — one entry per joined row, d.w NULL where unmatched
    {
      std::printf("query 6: SELECT t.key, d.w FROM t LEFT JOIN d ON t.fk = d.k WHERE t.i32 < 134217728\n");
      RelAlgExecutionUnit ra6;
      ra6.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(4, kTable, kDb, 0));
      ra6.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(0, kTable, kDb, 0));
      ra6.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(2, kTable, kDb, 0));
      ra6.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(0, kDim, kDb, 1));
      ra6.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(1, kDim, kDb, 1));
      auto i32 = colvar(t, kTable, 2);
      JoinCondition jc6;
      jc6.type = JoinType::LEFT;
      jc6.quals.push_back(std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, true), kEQ, fk, dk));
      ra6.join_quals.push_back(jc6);
      ra6.simple_quals.push_back(std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, true), kLT, i32, int_lit(kINT, 1 << 27)));
      ra6.groupby_exprs.push_back(nullptr);
      ra6.target_exprs = {v.get(), dw.get()};
      const size_t guess = (size_t)(N / 8);
      mi355q_plan h6{};
      h6.abi_version = MI355Q_ABI_VERSION;
      h6.n_cols = 3;
      h6.cols[0] = {MI355Q_INT64, 0, 0, 0};
      h6.cols[1] = {MI355Q_INT64, 0, 0, 0};
      h6.cols[2] = {MI355Q_INT32, 0, 0, 0};
      h6.col_ranges[0] = {1, 0, -50, M + 49, 0, 0, 0};
      h6.col_ranges[1] = {1, 0, 7, key_max, 0, 0, 0};
      h6.col_ranges[2] = {1, 0, 0, INT32_MAX, 0, 0, 0};
      h6.n_inner_cols = 2;
      h6.inner_cols[0] = {MI355Q_INT64, 0, 0, 0};
      h6.inner_cols[1] = {MI355Q_INT64, 0, 0, 0};
      h6.inner_col_ranges[0] = {1, 0, 0, M - 1, 0, 0, 0};
      h6.inner_col_ranges[1] = {1, 0, -1000, 1000, 0, 0, 0};
      h6.n_quals = 1;
      h6.quals[0] = {2, MI355Q_LT, 1 << 27, 0.0};
      h6.n_targets = 2;
      h6.targets[0] = {MI355Q_PROJECT, 1, 0, 0, {}};
      h6.targets[1] = {MI355Q_PROJECT, 1, 1, 0, {}};
      h6.join_outer_col = 0;
      h6.join_outer_cols[0] = 0;
      h6.join_kind = MI355Q_JOIN_LEFT;
      h6.max_groups_buffer_entry_guess = (int64_t)guess;
      h6.num_tuples = N + M;
      compare_plans(h6, mi355q_glue::to_plan(ra6, qi2, &executor, jt, guess, false));
      std::vector<int64_t> fr6_rows;
      const FetchResult fr6 = fetch(t, {4, 0, 2}, &fr6_rows);
      mi355q_qmd q6;
      const std::vector<int64_t> want6 = oracle_table(h6, t, {4, 0, 2}, fr6_rows, oj, {d.cols[0].host.data(), d.cols[1].host.data()}, M, &q6);
      const ResultSetPtr rs6 = mi355q_glue::run_query_mi355q(ra6, fr6, qi2, qmd_of(q6), &executor, 0, guess, jt,
                                                             {(const int8_t*)d.cols[0].dev, (const int8_t*)d.cols[1].dev}, M);
      expect(q6.desc_type == MI355Q_PROJECTION && q6.row_size == 24, "layout");
      const int64_t* got6 = (const int64_t*)const_cast<ResultSetStorage*>(rs6->getStorage())->getUnderlyingBuffer();
      bool same6 = true;
      int64_t nulls = 0, live = 0;
      for (size_t i = 0; i < want6.size() && same6; ++i) same6 = want6[i] == got6[i];
      for (int64_t e = 0; e < q6.entry_count; ++e) {
        if (want6[(size_t)e * 3] == INT64_MAX) break;
        ++live;
        nulls += want6[(size_t)e * 3 + 2] == INT64_MIN;
      }
      expect(same6, "projection-through-a-join buffer differs from the oracle's");
      expect(live > 0 && nulls > 0 && nulls < live, "the LEFT join shows both matched rows and NULL inner values");
      std::printf("  %lld rows, %lld of them without a match (d.w NULL)\n", (long long)live, (long long)nulls);
      // ---------------------------------------------------------- query 7: the same Projection through a ONE-TO-MANY table
      //   the dimension's keys folded onto half their range: every key has two rows -> two entries per matching outer row,
      //   in the payload's order (which depends on the build order: compared per outer row as a set)
      {
        std::printf("query 7: SELECT t.key, d2.w FROM t JOIN d2 ON t.fk = d2.k WHERE t.i32 < 134217728   (d2: every key twice)\n");
        std::vector<int64_t> k2((size_t)M);
        for (int64_t i = 0; i < M; ++i) k2[(size_t)i] = ((const int64_t*)d.cols[0].host.data())[i] % (M / 2);
        void* d_k2 = nullptr;
        HIPCK(hipMalloc(&d_k2, (size_t)M * 8));
        HIPCK(hipMemcpy(d_k2, k2.data(), (size_t)M * 8, hipMemcpyHostToDevice));
        mi355q_join_spec js2 = js;
        js2.key_buffer = d_k2;
        js2.key_range = {1, 0, 0, M / 2 - 1, 0, 0, 0};
        js2.one_to_many = 2;
        mi355q_join_table* jt2 = nullptr;
        MQCK(mi355q_join_build(&js2, nullptr, &jt2));
        const void* kc[1] = {k2.data()};
        const int32_t kt[1] = {MI355Q_INT64}, kn[1] = {0};
        int32_t oerr2 = 0;
        void* oj2 = orc_join_build_n(kc, kt, kn, 1, M, 0, M / 2 - 1, 0, 0, 2, 0, &oerr2);
        expect(oj2 != nullptr && oerr2 == 0, "oracle one-to-many join build");
        executor.column_ranges[{kDim, 0}] = ExpressionRange::makeIntRange(0, M / 2 - 1, 0, false);
        RelAlgExecutionUnit ra7 = ra6;
        ra7.join_quals.clear();
        JoinCondition jc7;
        jc7.type = JoinType::INNER;
        jc7.quals.push_back(std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, true), kEQ, fk, dk));
        ra7.join_quals.push_back(jc7);
        mi355q_plan h7 = h6;
        h7.join_kind = MI355Q_JOIN_INNER;
        h7.inner_col_ranges[0] = {1, 0, 0, M / 2 - 1, 0, 0, 0};
        const size_t guess7 = (size_t)(N / 2);
        h7.max_groups_buffer_entry_guess = (int64_t)guess7;
        compare_plans(h7, mi355q_glue::to_plan(ra7, qi2, &executor, jt2, guess7, false));
        mi355q_qmd q7;
        const std::vector<int64_t> want7 = oracle_table(h7, t, {4, 0, 2}, fr6_rows, oj2, {k2.data(), d.cols[1].host.data()}, M, &q7);
        const ResultSetPtr rs7 = mi355q_glue::run_query_mi355q(ra7, fr6, qi2, qmd_of(q7), &executor, 0, guess7, jt2,
                                                               {(const int8_t*)d_k2, (const int8_t*)d.cols[1].dev}, M);
        const int64_t* got7 = (const int64_t*)const_cast<ResultSetStorage*>(rs7->getStorage())->getUnderlyingBuffer();
        int64_t live7 = 0, pairs = 0;
        bool same7 = true;
        for (int64_t e = 0; e < q7.entry_count && same7;) {
          if (want7[(size_t)e * 3] == INT64_MAX) {
            same7 = got7[(size_t)e * 3] == INT64_MAX;
            break;
          }
          int64_t e1 = e;
          while (e1 < q7.entry_count && want7[(size_t)e1 * 3] == want7[(size_t)e * 3] && want7[(size_t)e1 * 3 + 1] == want7[(size_t)e * 3 + 1]) ++e1;
          std::vector<int64_t> a, b;
          for (int64_t x = e; x < e1; ++x) {
            same7 = same7 && got7[(size_t)x * 3] == want7[(size_t)x * 3] && got7[(size_t)x * 3 + 1] == want7[(size_t)x * 3 + 1];
            a.push_back(want7[(size_t)x * 3 + 2]);
            b.push_back(got7[(size_t)x * 3 + 2]);
          }
          std::sort(a.begin(), a.end());
          std::sort(b.begin(), b.end());
          same7 = same7 && a == b;
          pairs += (e1 - e) >= 2;
          live7 += e1 - e;
          e = e1;
        }
        expect(same7, "one-to-many projection differs from the oracle's");
        expect(live7 > 0 && pairs > 0, "every matching outer row shows its two inner rows");
        std::printf("  %lld entries, %lld outer rows with both of their matches\n", (long long)live7, (long long)pairs);
        executor.column_ranges[{kDim, 0}] = ExpressionRange::makeIntRange(0, M - 1, 0, false);
        mi355q_join_free(jt2);
        orc_join_free(oj2);
        HIPCK(hipFree(d_k2));
      }
    }
    mi355q_join_free(jt);
    orc_join_free(oj);
  }

  // ---------------------------------------------------------------- an error crosses the seam as QueryExecutionError
  {
    std::printf("query 4: the table of query 1 with room for a tenth of its groups -> QueryExecutionError\n");
    RelAlgExecutionUnit ra;
    for (int c : {0, 1}) ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(c, kTable, kDb, 0));
    auto key = colvar(t, kTable, 0), f64 = colvar(t, kTable, 1);
    ra.groupby_exprs.push_back(key);
    Analyzer::AggExpr cnt(SQLTypeInfo(kBIGINT, true), kCOUNT, nullptr), avg(SQLTypeInfo(kDOUBLE, true), kAVG, f64);
    ra.target_exprs = {key.get(), &cnt, &avg};
    std::vector<int64_t> frag_rows;
    const FetchResult fr = fetch(t, {0, 1}, &frag_rows);
    const mi355q_plan p = mi355q_glue::to_plan(ra, query_infos, &executor, nullptr, n_keys / 10, false);
    mi355q_qmd q;
    MQCK(orc_qmd_init(&p, &q));
    bool thrown = false;
    try {
      mi355q_glue::run_query_mi355q(ra, fr, query_infos, qmd_of(q), &executor, 0, n_keys / 10, nullptr, {}, 0);
    } catch (const QueryExecutionError& e) {
      thrown = e.getErrorCode() < 0 || e.getErrorCode() == MI355Q_ERR_OUT_OF_SLOTS;
    }
    expect(thrown, "out of slots must surface as QueryExecutionError (negative code or 3)");
  }
  // ---------------------------------------------------------------- query 5: a Projection (row-emitting filter / project)
  //   SELECT key, f64 * 2.0, x FROM t WHERE i32 < 2^27 [LIMIT 1000]  — groupby_exprs {nullptr}, no aggregate among the targets
  for (const size_t limit : {(size_t)0, (size_t)1000}) {
    std::printf("query 5: SELECT key, f64 * 2.0, x FROM t WHERE i32 < 134217728%s\n", limit ? " LIMIT 1000" : "");
    RelAlgExecutionUnit ra;
    for (int c : {0, 1, 2, 3}) ra.input_col_descs.push_back(std::make_shared<const InputColDescriptor>(c, kTable, kDb, 0));
    auto key = colvar(t, kTable, 0), f64 = colvar(t, kTable, 1), i32 = colvar(t, kTable, 2), x = colvar(t, kTable, 3);
    ra.simple_quals.push_back(std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kBOOLEAN, true), kLT, i32, int_lit(kINT, 1 << 27)));
    ra.groupby_exprs.push_back(nullptr);
    auto twice = std::make_shared<Analyzer::BinOper>(SQLTypeInfo(kDOUBLE, true), kMULTIPLY, f64, dbl_lit(2.0));
    ra.target_exprs = {key.get(), twice.get(), x.get()};
    ra.scan_limit = limit;
    const size_t guess = (size_t)(N / 8);  // (what the COUNT(*) pre-flight would say, with room: 1/16 of the rows match)
    mi355q_plan hp{};
    hp.abi_version = MI355Q_ABI_VERSION;
    hp.n_cols = 4;
    hp.cols[0] = {MI355Q_INT64, 0, 0, 0};
    hp.cols[1] = {MI355Q_DOUBLE, 0, 0, 0};
    hp.cols[2] = {MI355Q_INT32, 0, 0, 0};
    hp.cols[3] = {MI355Q_INT32, 1, 0, 0};
    hp.col_ranges[0] = {1, 0, 7, key_max, 0, 0, 0};
    hp.col_ranges[1] = {1, 0, 0, 0, 0.0, 1000.0, 0};
    hp.col_ranges[2] = {1, 0, 0, INT32_MAX, 0, 0, 0};
    hp.col_ranges[3] = {1, 0, 1, 40, 0, 0, 0};
    hp.n_quals = 1;
    hp.quals[0] = {2, MI355Q_LT, 1 << 27, 0.0};
    hp.n_exprs = 1;
    hp.exprs[0].n_nodes = 3;
    hp.exprs[0].nodes[0] = {MI355Q_EX_COL, 0, 1, 0, 0, 0.0};
    hp.exprs[0].nodes[1] = {MI355Q_EX_LIT, MI355Q_DOUBLE, 0, 0, 0, 2.0};
    hp.exprs[0].nodes[2] = {MI355Q_EX_MUL, MI355Q_DOUBLE, 0, 0, 0, 0.0};
    hp.exprs[0].range = {0, 0, 0, 0, 0.0, 0.0, 0};  // (the mock catalog has no range for a floating-point product; a projection's layout does not look at it)
    hp.n_targets = 3;
    hp.targets[0] = {MI355Q_PROJECT, 0, 0, 0, {}};
    hp.targets[1] = {MI355Q_PROJECT, 4, 0, 0, {}};
    hp.targets[2] = {MI355Q_PROJECT, 3, 0, 0, {}};
    hp.join_outer_col = -1;
    hp.max_groups_buffer_entry_guess = (int64_t)guess;
    hp.num_tuples = N;
    hp.scan_limit = (int64_t)limit;
    compare_plans(hp, mi355q_glue::to_plan(ra, query_infos, &executor, nullptr, guess, false));
    std::vector<int64_t> frag_rows;
    const FetchResult fr = fetch(t, {0, 1, 2, 3}, &frag_rows);
    mi355q_qmd q;
    const std::vector<int64_t> want = oracle_table(hp, t, {0, 1, 2, 3}, frag_rows, nullptr, {}, 0, &q);
    const std::string route = mi355q_glue::explain_query_mi355q(ra, query_infos, &executor, 0, guess, nullptr, 0, false);
    std::printf("  %s\n", route.c_str());
    expect(route.find("k_proj_compact") != std::string::npos, "a projection plans the compaction family");
    const ResultSetPtr rs = mi355q_glue::run_query_mi355q(ra, fr, query_infos, qmd_of(q), &executor, 0, guess, nullptr, {}, 0);
    expect(q.desc_type == MI355Q_PROJECTION && q.entry_count == (int64_t)(limit ? limit : guess) && q.row_size == 32, "layout");
    // the whole ResultSetStorage image, entry by entry: same rows in the same (fragment, row) order, EMPTY_KEY_64 tail
    const int64_t* got = (const int64_t*)const_cast<ResultSetStorage*>(rs->getStorage())->getUnderlyingBuffer();
    bool same = true;
    for (size_t i = 0; i < want.size() && same; ++i) same = want[i] == got[i];
    expect(same, "projection buffer differs from the oracle's");
  }

  std::printf(g_failures ? "glue_check: %d FAILURE(S)\n" : "glue_check: all queries agree with the oracle\n", g_failures);
  return g_failures ? 1 : 0;
}
