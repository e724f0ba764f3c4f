// col_check.cpp — native (no Python) device check of the columnar result path through the C-ABI:
// for a perfect-hash, a compact (4-byte slot) and a baseline-hash step it runs the step with
// MI355Q_OUTPUT_COLUMNAR and with MI355Q_OUTPUT_ROWWISE_COLUMNAR_DECISIONS and requires
//   * the columnar buffer to be the entry-by-entry transposition of the row-wise one
//     (perfect hash: bit-exact; baseline: the same key -> slots map),
//   * row count, iteration (fetch_rows) and reduce (columnar += columnar == 2 x) to agree,
//   * mi355q_result_create on a columnar descriptor to write the init image.
// Build: hipcc -O1 -std=c++17 tools/native/col_check.cpp -Iinclude -Lheavydb_amd/lib -lmi355q
//        -Wl,-rpath,$PWD/heavydb_amd/lib -o tools/native/col_check ;  exit code 0 = all checks passed.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#include "mi355q.h"

#define CK(x)                                                                  \
  do {                                                                         \
    int32_t _e = (x);                                                          \
    if (_e != 0) {                                                             \
      std::printf("FAIL %s -> %d (%s)\n", #x, _e, mi355q_error_string(_e));    \
      return 1;                                                                \
    }                                                                          \
  } while (0)
#define REQ(c)                                            \
  do {                                                    \
    if (!(c)) {                                           \
      std::printf("FAIL line %d: %s\n", __LINE__, #c);    \
      return 1;                                           \
    }                                                     \
  } while (0)

static std::vector<int64_t> to_rows(const mi355q_qmd& q, const std::vector<int64_t>& col) {
  const int rq = q.row_size / 8, kq = q.key_bytes / 8;
  const int64_t E = q.entry_count;
  std::vector<int64_t> rows((size_t)E * rq, 0);
  const char* b = (const char*)col.data();
  for (int64_t e = 0; e < E; ++e) {
    for (int k = 0; k < kq; ++k) rows[e * rq + k] = ((const int64_t*)(b + mi355q_qmd_group_col_offset(&q, k)))[e];
    for (int s = 0; s < q.slot_count; ++s) {
      const char* c = b + mi355q_qmd_slot_col_offset(&q, s);
      if (q.slot_width == 8) rows[e * rq + kq + s] = ((const int64_t*)c)[e];
      else ((int32_t*)&rows[e * rq + kq])[s] = ((const int32_t*)c)[e];
    }
  }
  return rows;
}

static int run_shape(const char* name, int shape, int64_t n) {
  void *d_key = nullptr, *d_val = nullptr;
  REQ(hipMalloc(&d_key, n * 8) == hipSuccess && hipMalloc(&d_val, n * 8) == hipSuccess);
  const bool baseline = shape == 2, compact = shape == 1;
  if (compact) CK(mi355q_generate_column(0, d_key, n, 0, MI355Q_GEN_I32_MOD, 11, 777, 0, 0, 0.0, 0, nullptr));
  else if (baseline) CK(mi355q_generate_column(0, d_key, n, 0, MI355Q_GEN_I64_MOD_MUL, 11, 40000, 1000003, 7, 0.0, 0, nullptr));
  else CK(mi355q_generate_column(0, d_key, n, 0, MI355Q_GEN_I64_MOD, 11, 1000, 5, 0, 0.0, 0, nullptr));
  CK(mi355q_generate_column(0, d_val, n, 0, MI355Q_GEN_I64_MOD, 12, 2001, -1000, 0, 0.0, 0, nullptr));
  mi355q_plan p;
  std::memset(&p, 0, sizeof(p));
  p.abi_version = MI355Q_ABI_VERSION;
  p.n_cols = 2;
  p.cols[0].type = compact ? MI355Q_INT32 : MI355Q_INT64;
  p.cols[1].type = MI355Q_INT64;
  if (!baseline) {
    p.col_ranges[0].valid = 1;
    p.col_ranges[0].min = compact ? 0 : 5;
    p.col_ranges[0].max = compact ? 776 : 1004;
  }
  p.col_ranges[1].valid = 1;
  p.col_ranges[1].min = -1000;
  p.col_ranges[1].max = 1000;
  p.n_group_cols = 1;
  p.group_cols[0] = 0;
  p.join_outer_col = -1;
  p.max_groups_buffer_entry_guess = 131072;
  p.num_tuples = n;
  int nt = 0;
  p.targets[nt].agg = MI355Q_PROJECT_KEY; p.targets[nt].col = 0; ++nt;
  p.targets[nt].agg = MI355Q_COUNT; p.targets[nt].col = -1; ++nt;
  if (!compact) {
    p.targets[nt].agg = MI355Q_AVG; p.targets[nt].col = 1; ++nt;
    p.targets[nt].agg = MI355Q_MIN; p.targets[nt].col = 1; ++nt;
  }
  p.n_targets = nt;
  const void* cols[2] = {d_key, d_val};
  int64_t rows_n[1] = {n};
  mi355q_inputs in;
  std::memset(&in, 0, sizeof(in));
  in.n_frags = 1;
  in.col_buffers = cols;
  in.num_rows = rows_n;
  mi355q_exec_options o;
  std::memset(&o, 0, sizeof(o));
  if (baseline) o.kernel_variant = 2;  // the partitioned family
  mi355q_exec_report rep;
  mi355q_result *rc = nullptr, *rr = nullptr, *rc2 = nullptr;
  p.output_columnar_hint = MI355Q_OUTPUT_COLUMNAR;
  CK(mi355q_execute(&p, &in, &o, &rc, &rep));
  CK(mi355q_execute(&p, &in, &o, &rc2, &rep));
  p.output_columnar_hint = MI355Q_OUTPUT_ROWWISE_COLUMNAR_DECISIONS;
  CK(mi355q_execute(&p, &in, &o, &rr, &rep));
  mi355q_qmd qc, qr;
  CK(mi355q_result_qmd(rc, &qc));
  CK(mi355q_result_qmd(rr, &qr));
  REQ(qc.output_columnar == 1 && qr.output_columnar == 0 && qc.entry_count == qr.entry_count && qc.row_size == qr.row_size);
  REQ(qc.slot_width == (compact ? 4 : 8));
  REQ(mi355q_result_bytes(rc) == mi355q_qmd_buffer_bytes(&qc));
  std::vector<int64_t> hc(mi355q_result_bytes(rc) / 8), hr(mi355q_result_bytes(rr) / 8);
  CK(mi355q_result_copy_to_host(rc, hc.data(), hc.size() * 8));
  CK(mi355q_result_copy_to_host(rr, hr.data(), hr.size() * 8));
  const std::vector<int64_t> tr = to_rows(qc, hc);
  REQ(tr.size() == hr.size());
  const int rq = qc.row_size / 8;
  const int64_t n_c = mi355q_result_row_count(rc), n_r = mi355q_result_row_count(rr);
  REQ(n_c == n_r && n_c > 0);
  if (!baseline) {
    REQ(std::memcmp(tr.data(), hr.data(), hr.size() * 8) == 0);
  } else {  // slot positions depend on insertion order: compare key -> slots maps
    std::map<int64_t, const int64_t*> m;
    for (int64_t e = 0; e < qr.entry_count; ++e)
      if (hr[e * rq] != INT64_MAX) m[hr[e * rq]] = &hr[e * rq];
    int64_t live = 0;
    for (int64_t e = 0; e < qc.entry_count; ++e) {
      if (tr[e * rq] == INT64_MAX) continue;
      ++live;
      auto it = m.find(tr[e * rq]);
      REQ(it != m.end());
      REQ(std::memcmp(it->second, &tr[e * rq], qc.row_size) == 0);
    }
    REQ(live == (int64_t)m.size() && live == n_c);
  }
  // iteration
  const int T = qc.n_targets;
  std::vector<int64_t> iv_c(n_c * T), iv_r(n_c * T);
  std::vector<double> dv_c(n_c * T), dv_r(n_c * T);
  std::vector<int8_t> nu_c(n_c * T), nu_r(n_c * T);
  int64_t g_c = 0, g_r = 0;
  CK(mi355q_result_fetch_rows(rc, n_c, iv_c.data(), dv_c.data(), nu_c.data(), &g_c));
  CK(mi355q_result_fetch_rows(rr, n_c, iv_r.data(), dv_r.data(), nu_r.data(), &g_r));
  REQ(g_c == n_c && g_r == n_c);
  if (!baseline) REQ(iv_c == iv_r && dv_c == dv_r && nu_c == nu_r);
  int64_t tot_c = 0, tot_r = 0;
  for (int64_t i = 0; i < n_c; ++i) {
    tot_c += iv_c[i * T + 1];
    tot_r += iv_r[i * T + 1];
  }
  REQ(tot_c == n && tot_r == n);  // COUNT(*) sums to the row count
  // reduce: columnar += columnar doubles every COUNT
  CK(mi355q_result_reduce(rc, rc2, nullptr));
  CK(mi355q_result_fetch_rows(rc, n_c, iv_c.data(), dv_c.data(), nu_c.data(), &g_c));
  REQ(g_c == n_c);
  tot_c = 0;
  for (int64_t i = 0; i < n_c; ++i) tot_c += iv_c[i * T + 1];
  REQ(tot_c == 2 * n);
  // init image of a fresh columnar result
  mi355q_result* fresh = nullptr;
  CK(mi355q_result_create(&qc, 0, nullptr, &fresh));
  REQ(mi355q_result_row_count(fresh) == 0);
  std::vector<int64_t> hf(mi355q_result_bytes(fresh) / 8);
  CK(mi355q_result_copy_to_host(fresh, hf.data(), hf.size() * 8));
  if (qc.key_bytes) REQ(hf[0] == INT64_MAX && hf[qc.entry_count - 1] == INT64_MAX);
  mi355q_result_free(fresh);
  mi355q_result_free(rc);
  mi355q_result_free(rc2);
  mi355q_result_free(rr);
  (void)hipFree(d_key);
  (void)hipFree(d_val);
  std::printf("ok  %-28s entries %lld groups %lld kernel %s\n", name, (long long)qc.entry_count, (long long)n_c,
              rep.kernel_name);
  return 0;
}

int main() {
  if (mi355q_device_count() < 1) {
    std::printf("FAIL no device\n");
    return 2;
  }
  int rc = 0;
  rc |= run_shape("perfect hash (8-byte slots)", 0, 1 << 20);
  rc |= run_shape("compact (4-byte slots)", 1, 1 << 20);
  rc |= run_shape("baseline hash (partitioned)", 2, 1 << 22);
  std::printf(rc ? "COLUMNAR CHECK FAILED\n" : "COLUMNAR CHECK PASSED\n");
  return rc;
}
