// api.cpp — the C-ABI of libmi355q (include/mi355q.h): host C++ that turns a plan into a
// launch sequence over the HIP kernel family.  No torch, no JIT, no CUDA-compat layer.
//
// Shape of one call, against the reference's step executor (heavyai/heavydb):
//   mi355q_execute ~ Executor::executeWorkUnit (Execute.cpp:2144) after fetchChunks:
//     plan -> QueryMemoryDescriptor mirror (plan.cpp) -> output buffer init (K6/K7 analogue,
//     QueryMemoryInitializer.cpp:1144) -> kernel family member chosen at plan time ->
//     error code folded from one device word (QueryExecutionContext.cpp:366 copies one int32
//     per thread; we keep a single word).
// The result stays in HBM; nothing is copied to the host unless the caller asks.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <new>
#include <vector>

#include "api_internal.h"
#include "boolfilter.h"

using namespace mq;

namespace mq {
namespace api {
thread_local std::string* t_route = nullptr;
thread_local bool t_plan_only = false;
void route_note(const char* what) {
  if (!t_route) return;
  if (!t_route->empty()) *t_route += " > ";
  *t_route += what;
}

thread_local hipError_t last_hip_error = hipSuccess;

int cu_count_of(int dev) {
  static std::mutex mu;
  static int cache[64];
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 0 || dev >= 64) return 256;
  if (!cache[dev]) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    cache[dev] = v;
  }
  return cache[dev];
}

DeviceCtx& ctx_of(int dev) {
  static DeviceCtx ctxs[64];
  return ctxs[dev < 0 ? 0 : dev % 64];
}

}  // namespace api
}  // namespace mq
using namespace mq::api;

namespace {

// MI355Q_OPT_TRACE: host-side wall-clock marks of one execute call on stderr
struct Trace {
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit Trace(bool enabled) : on(enabled), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "[mi355q] %8.3f ms  %s\n", ms, what);
  }
};


void drain_inflight(DeviceCtx& ctx);  // finishes the step mi355q_execute_async left in flight (defined with finish_step)

}  // namespace

int32_t mq::api::attach_join(const mi355q_plan& p, const mi355q_inputs* in, DevPlan* d) {
  if (p.join_outer_col < 0) return MI355Q_OK;
  if (!p.join_table) return MI355Q_ERR_INVALID_PLAN;
  const mi355q_join_table* jt = p.join_table;
  const int n_keys = p.n_join_cols > 1 ? p.n_join_cols : 1;
  if (n_keys > MI355Q_MAX_GROUP_COLS || n_keys != jt->n_keys) return MI355Q_ERR_INVALID_PLAN;
  if (p.join_kind != MI355Q_JOIN_INNER && p.join_kind != MI355Q_JOIN_LEFT) return MI355Q_ERR_UNSUPPORTED;
  for (int i = 0; i < n_keys; ++i) {
    const int c = (i == 0 && p.n_join_cols <= 1) ? p.join_outer_col : p.join_outer_cols[i];
    if (c < 0 || c >= p.n_cols) return MI355Q_ERR_INVALID_PLAN;
    const mi355q_col_desc& jc = p.cols[c];
    if (type_is_fp(jc.type) || type_is_f32(jc.type)) return MI355Q_ERR_UNSUPPORTED;
    d->join_cols[i] = c;
    d->join_types[i] = col_type_code(jc);
    d->join_nullables[i] = jc.nullable != 0;
  }
  d->join_col = d->join_cols[0];
  d->join_type = d->join_types[0];
  d->join_nullable = d->join_nullables[0];
  d->join_n_keys = n_keys;
  d->join_width = jt->width;
  d->join_kind = p.join_kind;
  d->join_hash_type = jt->hash_type;
  d->join_buf = jt->buf;
  d->join_bitmap = (const uint32_t*)jt->bitmap;
  d->join_min = jt->min_key;
  d->join_max = jt->max_key;
  d->join_entries = jt->entry_count;
  for (int i = 0; i < p.n_inner_cols; ++i) {
    d->inner_cols[i] = in && in->inner_col_buffers ? (const int8_t*)in->inner_col_buffers[i] : nullptr;
  }
  for (int i = 0; i < p.n_targets; ++i) {
    // (an empty inner table has no chunks: nothing can match, the pointers are never followed)
    if (d->targets[i].table == 1 && d->targets[i].col >= 0 && !d->inner_cols[d->targets[i].col] &&
        in && in->inner_num_rows > 0)
      return MI355Q_ERR_INVALID_PLAN;
  }
  return MI355Q_OK;
}

int64_t mq::api::algorithmic_bytes(const mi355q_plan& p, const mi355q_inputs& in) {
  // every distinct outer column the plan touches is read once per row
  bool used[MI355Q_MAX_COLS] = {false};
  for (int k = 0; k < p.n_exprs && k < MI355Q_MAX_EXPRS; ++k)  // an expression reads its operand columns
    for (int i = 0; i < p.exprs[k].n_nodes && i < MI355Q_MAX_EXPR_NODES; ++i)
      if (p.exprs[k].nodes[i].op == MI355Q_EX_COL && p.exprs[k].nodes[i].arg >= 0 && p.exprs[k].nodes[i].arg < p.n_cols)
        used[p.exprs[k].nodes[i].arg] = true;
  for (int i = 0; i < p.n_quals; ++i) used[p.quals[i].col] = true;
  for (int g = 0; g < p.n_group_cols; ++g) used[p.group_cols[g]] = true;
  if (p.join_outer_col >= 0) {
    used[p.join_outer_col] = true;
    for (int i = 1; i < p.n_join_cols && i < MI355Q_MAX_GROUP_COLS; ++i) used[p.join_outer_cols[i]] = true;
  }
  for (int i = 0; i < p.n_targets; ++i) {
    if (p.targets[i].table == 0 && p.targets[i].col >= 0 && p.targets[i].agg != MI355Q_PROJECT_KEY)
      used[p.targets[i].col] = true;
    if (p.targets[i].agg == MI355Q_COUNT_IF || p.targets[i].agg == MI355Q_SUM_IF) used[p.targets[i].cond.col] = true;
  }
  int64_t per_row = 0;
  for (int c = 0; c < p.n_cols; ++c) {  // physical columns only (indices >= n_cols are expressions)
    if (used[c]) per_row += type_width(p.cols[c].type);
  }
  int64_t rows = 0;
  for (int f = 0; f < in.n_frags; ++f) rows += in.num_rows[f];
  return per_row * rows;
}


extern "C" {

int32_t mi355q_abi_version(void) { return MI355Q_ABI_VERSION; }

int64_t mi355q_abi_sizeof(int32_t which) {
  switch (which) {
    case 1: return sizeof(mi355q_plan);
    case 2: return sizeof(mi355q_qmd);
    case 3: return sizeof(mi355q_inputs);
    case 4: return sizeof(mi355q_exec_options);
    case 5: return sizeof(mi355q_exec_report);
    case 6: return sizeof(mi355q_join_spec);
    default: return -1;
  }
}

const char* mi355q_error_string(int32_t code) {
  if (code < 0) return "ran out of group slots (negative row position): resize and retry";
  switch (code) {
    case MI355Q_OK: return "No Error";
    case MI355Q_ERR_DIV_BY_ZERO: return "Division by zero";
    case MI355Q_ERR_OUT_OF_GPU_MEM: return "Out of GPU memory";
    case MI355Q_ERR_OUT_OF_SLOTS: return "Out of Slots";
    case MI355Q_ERR_OUT_OF_CPU_MEM: return "Not enough host memory to execute the query";
    case MI355Q_ERR_OVERFLOW_OR_UNDERFLOW: return "Overflow or underflow";
    case MI355Q_ERR_OUT_OF_TIME: return "Query execution has exceeded the time limit";
    case MI355Q_ERR_INTERRUPTED: return "Query execution has been interrupted";
    case MI355Q_ERR_INVALID_PLAN: return "Invalid plan";
    case MI355Q_ERR_UNSUPPORTED: return "Plan shape not supported by this kernel family";
    case MI355Q_ERR_HIP: return hipGetErrorString(last_hip_error);
    case MI355Q_ERR_JOIN_NOT_ONE_TO_ONE: return "Join key column is not unique (one-to-many)";
    case MI355Q_ERR_JOIN_TABLE_FULL: return "Keyed join hash table is full";
    case MI355Q_STEP_RECOMPUTED: return "Step re-run inside mi355q_wait: redo what was enqueued behind it";
    default: return "Unknown error";
  }
}

int32_t mi355q_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t mi355q_device_info(int32_t device_id, char* name, int32_t* cu_count, int64_t* total_mem,
                           int64_t* free_mem, int32_t* mem_clock_khz, int32_t* mem_bus_width) {
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  if (name) {
    std::snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (mem_clock_khz) *mem_clock_khz = prop.memoryClockRate;
  if (mem_bus_width) *mem_bus_width = prop.memoryBusWidth;
  DeviceGuard g(device_id);
  size_t fr = 0, tot = 0;
  HIP_TRY(hipMemGetInfo(&fr, &tot));
  if (total_mem) *total_mem = (int64_t)tot;
  if (free_mem) *free_mem = (int64_t)fr;
  return MI355Q_OK;
}

int32_t mi355q_release_workspace(int32_t device_id) {
  DeviceCtx& ctx = ctx_of(device_id);
  std::lock_guard<std::recursive_mutex> lk(ctx.mu);
  DeviceGuard g(device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  // a step left in flight by mi355q_execute_async still points into the workspace (error words, column table,
  // events, the launch stream): finish it before anything is freed
  drain_inflight(ctx);
  if (ctx.scratch) (void)hipFree(ctx.scratch);
  if (ctx.meta) (void)hipFree(ctx.meta);
  if (ctx.h_meta) (void)hipHostFree(ctx.h_meta);
  ctx.h_meta = nullptr;
  if (ctx.aux) (void)hipFree(ctx.aux);
  ctx.aux = nullptr;
  ctx.aux_bytes = 0;
  if (ctx.wide) (void)hipFree(ctx.wide);
  ctx.wide = nullptr;
  ctx.wide_bytes = 0;
  if (ctx.proj) (void)hipFree(ctx.proj);
  ctx.proj = nullptr;
  ctx.proj_bytes = 0;
  if (ctx.gather) (void)hipFree(ctx.gather);
  ctx.gather = nullptr;
  ctx.gather_bytes = 0;
  if (ctx.lattice) (void)hipFree(ctx.lattice);
  ctx.lattice = nullptr;
  ctx.lattice_bytes = 0;
  if (ctx.bf_table) (void)hipFree(ctx.bf_table);
  ctx.bf_table = nullptr;
  if (ctx.maskws) (void)hipFree(ctx.maskws);
  ctx.maskws = nullptr;
  ctx.maskws_bytes = 0;
  if (ctx.projws) (void)hipFree(ctx.projws);
  ctx.projws = nullptr;
  ctx.projws_bytes = 0;
  if (ctx.stream) (void)hipStreamDestroy(ctx.stream);
  ctx.stream = nullptr;
  for (hipEvent_t e : ctx.events) (void)hipEventDestroy(e);
  ctx.scratch = ctx.meta = nullptr;
  ctx.scratch_bytes = 0;
  ctx.meta_bytes = 0;
  ctx.events.clear();
  return MI355Q_OK;
}

int32_t mi355q_qmd_init(const mi355q_plan* plan, mi355q_qmd* out) {
  if (!plan || !out) return MI355Q_ERR_INVALID_PLAN;
  return qmd_init(*plan, out);
}

int64_t mi355q_qmd_buffer_bytes(const mi355q_qmd* qmd) { return qmd ? qmd_buffer_bytes(*qmd) : 0; }
int64_t mi355q_qmd_group_col_offset(const mi355q_qmd* qmd, int32_t g) { return qmd ? qmd_group_col_offset(*qmd, g) : -1; }
int64_t mi355q_qmd_slot_col_offset(const mi355q_qmd* qmd, int32_t s) {
  return qmd && s < qmd->slot_count ? qmd_slot_col_offset(*qmd, s) : -1;
}

}  // extern "C"

extern "C" {

// ------------------------------------------------------------------------------- packed keys
// Baseline-hash GROUP BY over several columns whose ranges together fit 62 bits: pack the key
// columns of every row into one int64 column (one streaming pass), run the step on that column
// with the single-key fast family (partition-then-aggregate), and re-emit the finished table
// with the real key components — MurmurHash3 over the key bytes, the reference's layout.  The
// slots of a row are the same in both tables (key projections take no slot in the baseline
// layout), so the re-emission copies them.  Anything this cannot take (no ranges, too many
// bits, joins, small inputs) stays with the row kernel.
namespace {

constexpr int32_t kNotTaken = INT32_MIN + 7;  // internal: "use the ordinary path"
constexpr int64_t kIdxPartMinRows = (int64_t)8 << 20;  // below this the row kernel / LDS members are as good
constexpr int32_t kRetryNoIdx = INT32_MIN + 9;  // internal: the index-partitioned family gave up (spill list), plan again without it
constexpr int32_t kRetryNoLds = INT32_MIN + 8;  // internal: the LDS group-by ran out of replica room, plan again without it

bool pack_spec_of(const mi355q_plan& p, const mi355q_qmd& q, const DevPlan& d, PackSpec* ps) {
  if (p.join_outer_col >= 0 || p.n_group_cols < 1 || d.col0_key_quirk) return false;
  std::memset(ps, 0, sizeof(*ps));
  ps->n = p.n_group_cols;
  // ONE plain FLOAT key on a baseline table: the table stores the double it widens to, so the "packed" column is that
  // double's bit pattern and the single-key families (which read 8-byte keys) take the step — the reference
  // benchmark's GROUP BY cast(x AS FLOAT) (MultiStep/MSBS001-005)
  if (p.n_group_cols == 1 && q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH && p.cols[p.group_cols[0]].type == MI355Q_FLOAT &&
      p.cols[p.group_cols[0]].encoding == MI355Q_ENC_NONE && q.key_width == 8) {
    ps->raw_f32 = 1;
    ps->cols[0] = p.group_cols[0];
    ps->types[0] = d.group_types[0];
    return true;
  }
  for (int g = 0; g < p.n_group_cols; ++g)  // other floating-point keys have no integer range to pack
    if (type_is_fp(p.cols[p.group_cols[g]].type) || type_is_f32(p.cols[p.group_cols[g]].type)) return false;
  if (q.desc_type == MI355Q_GROUP_BY_PERFECT_HASH) {
    // single-column tables that fit LDS already have their kernel; multi-column ones reach it
    // through the packed index (mode 2); bucketed keys cannot be restored from their index
    const bool small = q.entry_count * (int64_t)q.row_size <= 64 * 1024;
    // one plain NOT NULL INT / BIGINT (or FIXED(32)) key: k_perfect_lds reads it natively.  Every other
    // single key — dictionary ids, FIXED(8/16), DATE in days, nullable (translated) keys — reaches the
    // same kernel through the index column the pack kernel decodes it into.
    if (small && p.n_group_cols < 2) {
      const mi355q_col_desc& kc = p.cols[p.group_cols[0]];
      const int tc = d.group_types[0];
      const bool native = !kc.nullable && !q.group_bucket[0] &&
                          (tc == MI355Q_INT32 || tc == MI355Q_INT64 ||
                           (tc_enc(tc) == MI355Q_ENC_FIXED && tc_storage(tc) == MI355Q_INT32));
      if (native) return false;
    }
    if (q.entry_count >= ((int64_t)1 << 31)) return false;
    ps->mode = small ? 2 : 1;
    for (int g = 0; g < p.n_group_cols; ++g) {
      // a bucketed range maps (key - min) / bucket to the index: the key can only be restored from the
      // index when every value is a multiple of the bucket above min — DATE in days decoded to seconds
      if (q.group_bucket[g] &&
          !(tc_enc(d.group_types[g]) == MI355Q_ENC_DATE_IN_DAYS && q.group_bucket[g] == 86400 && q.group_min[g] % 86400 == 0))
        return false;
      ps->bucket[g] = q.group_bucket[g];
      ps->cols[g] = p.group_cols[g];
      ps->types[g] = d.group_types[g];
      ps->translate[g] = d.group_translate[g];
      ps->min[g] = q.group_min[g];
      ps->card[g] = (uint64_t)q.group_card[g];
      ps->mul[g] = d.group_mul[g];
      ps->null_key[g] = q.group_null_key[g];
    }
    return true;
  }
  // baseline layouts: several key columns, or ONE key column whose table has 4-byte key components
  // (pick_baseline_key_width: every key fits int32 — the reference benchmark's x10k_s10k / x100k_s10k BIGINT keys):
  // the partitioned family emits 8-byte keys, so the step runs on the packed (key - min) column into a temporary
  // 8-byte-key table and k_unpack_emit lays the finished groups out with the real 4-byte components
  if (q.desc_type != MI355Q_GROUP_BY_BASELINE_HASH) return false;
  if (p.n_group_cols < 2 && q.key_width != 4) return false;
  int shift = 0;
  for (int g = 0; g < p.n_group_cols; ++g) {
    const int c = p.group_cols[g];
    const mi355q_range& r = p.col_ranges[c];
    if (!r.valid || r.min > r.max) return false;
    const __int128 span = (__int128)r.max - (__int128)r.min + 1 + (p.cols[c].nullable ? 1 : 0);
    if (span > ((__int128)1 << 61)) return false;
    int bits = 1;
    while (((__int128)1 << bits) < span) ++bits;
    if (shift + bits > 62) return false;  // the packed key must stay below EMPTY_KEY_64
    ps->cols[g] = c;
    ps->types[g] = d.group_types[g];
    ps->nullable[g] = p.cols[c].nullable != 0;
    ps->shift[g] = shift;
    ps->min[g] = r.min;
    ps->card[g] = (uint64_t)span;
    ps->mask[g] = (((uint64_t)1) << bits) - 1;
    shift += bits;
  }
  return true;
}

int32_t execute_impl(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options* opts, mi355q_result** out,
                     mi355q_exec_report* report, mi355q_pending** pend, int64_t* reserved);

int32_t execute_packed_multi(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o,
                             const mi355q_qmd& q, const DevPlan& d, int n_cus, mi355q_result** out,
                             mi355q_exec_report* report, int64_t* reserved) {
  PackSpec ps;
  // kernel_variant 1 = "the row kernel / direct members", as everywhere else
  if (o.kernel_variant == 1) return kNotTaken;
  if (!pack_spec_of(*plan, q, d, &ps)) return kNotTaken;
  const int nf = in->n_frags, nc = plan->n_cols;
  int64_t total_rows = 0, max_frag_rows = 0;
  for (int f = 0; f < nf; ++f) {
    if (in->num_rows[f] < 0) return MI355Q_ERR_INVALID_PLAN;
    total_rows += in->num_rows[f];
    max_frag_rows = std::max(max_frag_rows, in->num_rows[f]);
  }
  // small inputs: the row kernel is as fast (kernel_variant 2 asks for this path regardless, 1 for
  // the row kernel)
  if (o.kernel_variant == 1 || (o.kernel_variant != 2 && total_rows < ((int64_t)8 << 20))) return kNotTaken;

  // derived plan: the packed column (appended, range unknown -> baseline, 8-byte key) is the
  // only group column; key projections are dropped (they own no slot)
  mi355q_plan p2 = *plan;
  // where the packed column sits in the derived step's column table: appended — or, when the table is full
  // (MI355Q_MAX_COLS), in the place of a key column nothing else reads (its only reader, the grouping, is what the packed
  // column replaces; key projections own no slot in the derived plan)
  int pk = nc;
  if (nc >= MI355Q_MAX_COLS) {
    pk = -1;
    for (int g = 0; g < plan->n_group_cols && pk < 0; ++g) {
      const int c = plan->group_cols[g];
      bool used = false;
      for (int t = 0; t < plan->n_targets; ++t) {
        const mi355q_target& tg = plan->targets[t];
        if (tg.agg == MI355Q_PROJECT_KEY) continue;
        used = used || (tg.table == 0 && tg.col == c) ||
               ((tg.agg == MI355Q_COUNT_IF || tg.agg == MI355Q_SUM_IF) && tg.cond.col == c);
      }
      for (int k = 0; k < plan->n_quals; ++k) used = used || plan->quals[k].col == c;
      if (!used) pk = c;
    }
    if (pk < 0) return kNotTaken;
  }
  const int nc2 = pk == nc ? nc + 1 : nc;
  p2.n_cols = nc2;
  p2.cols[pk] = mi355q_col_desc{MI355Q_INT64, 0, MI355Q_ENC_NONE, 0};
  p2.col_ranges[pk] = mi355q_range{};
  p2.n_group_cols = 1;
  p2.group_cols[0] = pk;
  p2.n_targets = 0;
  for (int t = 0; t < plan->n_targets; ++t) {
    if (plan->targets[t].agg != MI355Q_PROJECT_KEY) p2.targets[p2.n_targets++] = plan->targets[t];
  }
  if (p2.n_targets == 0) return kNotTaken;
  if (ps.mode == 1) {  // a baseline table at 50 % fill for the groups of the perfect layout
    const int64_t guess = 2 * q.entry_count;
    if (guess > (int64_t)UINT32_MAX) return kNotTaken;
    p2.max_groups_buffer_entry_guess = guess;
  }
  if (ps.mode == 2) {  // the index is a perfect-hash key itself: range [0, entries)
    p2.col_ranges[pk].valid = 1;
    p2.col_ranges[pk].min = 0;
    p2.col_ranges[pk].max = q.entry_count - 1;
  }
  // the temporary table always has 8-byte slots: k_unpack_emit / k_unpack_perfect read it quad by
  // quad, and a multi-pass run reduces it with 64-bit adds (pick_target_compact_width would narrow a
  // COUNT(*)-only derived plan to 4-byte slots)
  p2.bigint_count = 1;
  mi355q_qmd q2;
  if (qmd_init(p2, &q2) != MI355Q_OK) return kNotTaken;
  if (q2.slot_width != 8 || q2.row_size % 8 != 0) return kNotTaken;
  if (ps.mode == 2) {
    if (q2.desc_type != MI355Q_GROUP_BY_PERFECT_HASH || q2.entry_count != q.entry_count) return kNotTaken;
    ps.tmp_idx_target = q2.idx_target_as_key;
    ps.tmp_init = q2.keyless ? q2.init_vals[q2.idx_target_as_key] : 0;
  } else if (q2.desc_type != MI355Q_GROUP_BY_BASELINE_HASH) {
    return kNotTaken;
  }
  if (ps.mode == 0 && (q2.slot_count != q.slot_count || q2.entry_count != q.entry_count)) return kNotTaken;
  // where every slot of the final row comes from: a slot of the packed step's row, or (perfect
  // layouts: projected keys own a slot) the original value of a key component
  {
    int t2 = 0;
    for (int i = 0; i < MI355Q_MAX_SLOTS; ++i) ps.slot_src[i] = 0;
    for (int t = 0; t < plan->n_targets; ++t) {
      const int sf = q.target_slot[t];
      if (plan->targets[t].agg == MI355Q_PROJECT_KEY) {
        if (sf >= 0) ps.slot_src[sf] = -(1 + q.target_key_idx[t]);
        continue;
      }
      const int st = q2.target_slot[t2++];
      const int ns = plan->targets[t].agg == MI355Q_AVG ? 2 : 1;
      for (int j = 0; j < ns; ++j) {
        if (sf < 0 || st < 0 || q2.init_vals[st + j] != q.init_vals[sf + j]) return kNotTaken;
        ps.slot_src[sf + j] = st + j;
      }
    }
  }

  if (reserved) {  // RESERVE / explain: the derived single-key step over all fragments (one pass), nothing launched
    route_note(ps.raw_f32 ? "k_pack_keys (FLOAT key widened)" : ps.mode == 0 ? "k_pack_keys (bit-packed key)" : ps.mode == 1 ? "k_pack_keys (entry index, baseline temp)"
                                                                             : "k_pack_keys (entry index, perfect temp)");
    std::vector<const void*> cols2((size_t)nf * nc2, nullptr);
    for (int i = 0; i < nf; ++i)
      for (int c = 0; c < nc; ++c)
        if (c != pk) cols2[(size_t)i * nc2 + c] = in->col_buffers[(size_t)i * nc + c];
    mi355q_inputs in2 = *in;
    in2.col_buffers = cols2.data();
    mi355q_exec_options o2 = o;
    o2.out_buffer = nullptr;
    const int32_t e2 = execute_impl(&p2, &in2, &o2, out, report, nullptr, reserved);
    if (e2 == MI355Q_OK) route_note(ps.mode == 2 ? "k_unpack_perfect" : "k_unpack_emit");
    return e2;
  }
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  // memory: two temporary tables + the packed column of one pass (a group of fragments)
  const int64_t tmp_bytes = (q2.entry_count * (int64_t)q2.row_size + 255) & ~255ll;
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  int64_t pack_budget = std::min<int64_t>((int64_t)16 << 30, ((int64_t)free_b + ctx.aux_bytes) / 3);
  int64_t pass_rows = pack_budget / 8;
  if (o.pass_rows > 0 && o.pass_rows < pass_rows)  // tests: force several passes
    pass_rows = std::max<int64_t>(o.pass_rows, max_frag_rows);
  if (pass_rows < max_frag_rows + 2 * (int64_t)nf) return kNotTaken;
  if (pass_rows > total_rows) pass_rows = total_rows;
  const int64_t pack_bytes = ((pass_rows + 2 * (int64_t)nf) * 8 + 255) & ~255ll;
  const int64_t tab_bytes = sizeof(void*) * (size_t)nf;
  const int64_t need = pack_bytes + 2 * tmp_bytes + ((tab_bytes + 255) & ~255ll) + 256;
  if (ctx.aux_bytes < need) {
    if (ctx.aux) (void)hipFree(ctx.aux);
    ctx.aux = nullptr;
    ctx.aux_bytes = 0;
    if (hipMalloc(&ctx.aux, (size_t)need) != hipSuccess) {
      (void)hipGetLastError();
      return kNotTaken;
    }
    ctx.aux_bytes = need;
  }
  char* aux = (char*)ctx.aux;
  int64_t* packed = (int64_t*)aux;
  int64_t* tmp_a = (int64_t*)(aux + pack_bytes);
  int64_t* tmp_b = (int64_t*)(aux + pack_bytes + tmp_bytes);
  int64_t** d_packed_tab = (int64_t**)(aux + pack_bytes + 2 * tmp_bytes);
  int32_t* d_err = (int32_t*)(aux + pack_bytes + 2 * tmp_bytes + ((tab_bytes + 255) & ~255ll));

  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { mi355q_result_free(r); }
  } rg{res};
  // device copy of the caller's column table (the pack kernel reads the key columns from it)
  DevWord d_tab, d_rows;
  HIP_TRY(hipMalloc(&d_tab.p, sizeof(void*) * (size_t)std::max(1, nf * nc)));
  HIP_TRY(hipMalloc(&d_rows.p, sizeof(int64_t) * (size_t)std::max(1, nf)));
  HIP_TRY(hipMemcpyAsync(d_tab.p, in->col_buffers, sizeof(void*) * (size_t)(nf * nc), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d_rows.p, in->num_rows, sizeof(int64_t) * (size_t)nf, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(d_err, 0, 64, s));

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};

  std::vector<const void*> cols2((size_t)nf * nc2);
  std::vector<int64_t*> h_packed((size_t)nf);
  mi355q_exec_report acc{};
  int pass = 0;
  int f = 0;
  while (f < nf) {
    int f1 = f;
    int64_t rows = 0, off = 0;
    while (f1 < nf && (f1 == f || rows + in->num_rows[f1] <= pass_rows)) {
      h_packed[f1] = packed + off;
      off += (in->num_rows[f1] + 1) & ~(int64_t)1;  // 16-byte aligned fragment chunks
      rows += in->num_rows[f1];
      ++f1;
    }
    const int pnf = f1 - f;
    HIP_TRY(hipMemcpyAsync(d_packed_tab, h_packed.data() + f, sizeof(void*) * (size_t)pnf, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_pack_keys(ps, (const int8_t* const*)d_tab.p + (size_t)f * nc, (const int64_t*)d_rows.p + f, pnf,
                             nc, max_frag_rows, d_packed_tab, d_err, n_cus, s));
    for (int i = 0; i < pnf; ++i) {
      for (int c = 0; c < nc; ++c) cols2[(size_t)i * nc2 + c] = in->col_buffers[(size_t)(f + i) * nc + c];
      cols2[(size_t)i * nc2 + pk] = h_packed[f + i];
    }
    mi355q_inputs in2 = *in;
    in2.n_frags = pnf;
    in2.col_buffers = cols2.data();
    in2.num_rows = in->num_rows + f;
    mi355q_exec_options o2 = o;
    o2.stream = s;
    o2.out_buffer = pass == 0 ? tmp_a : tmp_b;
    mi355q_result* r2 = nullptr;
    mi355q_exec_report rep2{};
    const int32_t e2 = mi355q_execute(&p2, &in2, &o2, &r2, &rep2);
    if (e2 == MI355Q_ERR_OUT_OF_GPU_MEM || e2 == MI355Q_ERR_UNSUPPORTED) return kNotTaken;
    if (e2) return e2;  // incl. < 0: out of slots -> the caller grows the table
    struct R2Guard {
      mi355q_result* r;
      ~R2Guard() { if (r) mi355q_result_free(r); }
    } r2g{r2};
    if (pass > 0) {
      HIP_TRY(launch_reduce(r2->dplan, q2.idx_target_as_key, tmp_a, tmp_b, q2.entry_count, d_err, s));
    }
    if (pass == 0) {  // the report names the member the first (a full-sized) pass ran
      std::snprintf(acc.kernel_name, sizeof(acc.kernel_name), "%s", rep2.kernel_name);
      acc.variant = rep2.variant;
    }
    acc.kernel_ms += rep2.kernel_ms;
    acc.n_launches += rep2.n_launches;
    acc.spilled_rows += rep2.spilled_rows;
    f = f1;
    ++pass;
  }
  HIP_TRY(launch_init_buffer(res->buf, q.entry_count, make_row_init(q), s));
  if (nf > 0)
    HIP_TRY(launch_unpack_emit(ps, d, tmp_a, q2.entry_count, q2.row_size / 8, q2.key_bytes / 8, res->buf, d_err, s));
  if (ev1) HIP_TRY(hipEventRecord(ev1, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_err == MI355Q_ERR_UNSUPPORTED) return kNotTaken;  // a key outside its declared range
  if (h_err) return h_err;
  if (report) {
    *report = acc;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}


// ---------------------------------------------------------------------------------------------------
// The part of a step that needs the host to look at the device: error words, the spill counter, the
// re-run with the direct member when the partitioned family gave up, the report.  mi355q_execute runs it
// right after the launches; mi355q_execute_async parks it in a mi355q_pending until mi355q_wait (or the
// next call on the device) runs it.  Everything it touches is owned here: the caller's host arrays may be
// gone by then.
enum StepKind { K_GENERIC, K_SCAN_COUNT, K_PERFECT_LDS, K_BASELINE_FAST, K_JOIN_SUM, K_JOIN_PART, K_JOIN_PROBE, K_SCAN_AGG, K_LDS_GROUPBY, K_IDX_PART };
struct TailState {
  mi355q_qmd q;
  DevPlan d;
  StepKind kind;
  LaunchStats st;
  mi355q_result* res;
  int32_t* d_err;
  const int8_t* const* d_cols;
  const int64_t* d_rows;
  std::vector<const void*> h_cols;
  std::vector<int64_t> h_rows;
  int nf, nc, n_cus;
  int64_t total_rows, max_frag_rows, alg_bytes;
  hipStream_t s;
  hipEvent_t ev_start, ev_stop;
  hipEvent_t* ev_pool;
  bool trace;
  int32_t* h_ret = nullptr; // 64 pinned bytes the error words (+ the spill counter's copy, word 4) are read back into
  TuneKnobs knobs;          // the knobs the step was planned with: a re-run in mi355q_wait (another thread, another
                            // call's knobs in between) must plan with the same ones
  int32_t* h_ret_dev = nullptr;  // the device's address of h_ret
  bool err_words_zero = false;   // the step left the error words as the upload laid them down
  bool recomputed = false;  // finish_step re-ran the step into res->buf after the first launches had completed
};

struct KnobScope {  // the thread's knobs for the duration of finish_step
  TuneKnobs saved;
  explicit KnobScope(const TuneKnobs& k) : saved(tune_knobs()) { set_tune_knobs(k); }
  ~KnobScope() { set_tune_knobs(saved); }
};

int32_t finish_step(TailState& t, mi355q_exec_report* report) {
  KnobScope knob_scope(t.knobs);
  hipStream_t s = t.s;
  LaunchStats& st = t.st;
  const mi355q_qmd& q = t.q;
  const DevPlan& d = t.d;
  FragView fv{t.d_cols, t.d_rows, t.h_cols.data(), t.h_rows.data(), t.nf, t.nc, t.total_rows, t.max_frag_rows};
  int32_t h_err[2] = {0, 0};
  uint32_t h_spills = 0;
  if (t.h_ret && (!st.spill_counter32 || st.spill_counter32 == (uint32_t*)(t.d_err + 4))) {
    // error words and the spill counter's copy (word 4) in one read into pinned memory: written there by a one-wave kernel
    // behind the step's kernels (a copy command costs a DMA round trip more than a launch)
    if (t.h_ret_dev) HIP_TRY(launch_words_to_host(t.d_err, t.h_ret_dev, 8, s));
    else HIP_TRY(hipMemcpyAsync(t.h_ret, t.d_err, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h_err[0] = t.h_ret[0];
    h_err[1] = t.h_ret[1];
    if (st.spill_counter32) h_spills = (uint32_t)t.h_ret[4];
    t.err_words_zero = true;
    for (int w = 0; w < 8; ++w) t.err_words_zero = t.err_words_zero && t.h_ret[w] == 0;
  } else {
    HIP_TRY(hipMemcpyAsync(h_err, t.d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
    if (st.spill_counter32) {
      HIP_TRY(hipMemcpyAsync(&h_spills, st.spill_counter32, sizeof(h_spills), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
  }
  st.spilled_rows = (int64_t)h_spills;
  if (h_err[1] && (t.kind == K_JOIN_PART || t.kind == K_JOIN_PROBE || t.kind == K_BASELINE_FAST)) t.recomputed = true;
  if (h_err[1] && t.trace) std::fprintf(stderr, "[mi355q] partitioned family gave up (code %d, spills %u): re-running with the direct kernel\n", h_err[1], h_spills);
  if (h_err[1] && t.kind == K_JOIN_PART) {
    // the partitioned probe ran out of spill space (extreme skew): redo with the direct probe
    HIP_TRY(hipMemsetAsync(t.d_err, 0, 64, s));
    HIP_TRY(launch_init_buffer(t.res->buf, q.entry_count, make_row_init(q), s));
    HIP_TRY(launch_join_sum(d, fv, t.res->buf, t.n_cus, s, &st));
    HIP_TRY(hipMemcpyAsync(h_err, t.d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  if (h_err[1] && t.kind == K_JOIN_PROBE) {
    // the payload probe ran out of spill space (extreme skew): redo the step with the row kernel
    HIP_TRY(hipMemsetAsync(t.d_err, 0, 64, s));
    HIP_TRY(launch_init_buffer(t.res->buf, q.entry_count, make_row_init(q), s));
    HIP_TRY(launch_generic(d, q.idx_target_as_key, make_row_init(q), t.d_cols, t.d_rows, t.nf, t.max_frag_rows, t.res->buf,
                           t.d_err, t.n_cus, s));
    st.kernel_name = "k_generic";
    HIP_TRY(hipMemcpyAsync(h_err, t.d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  // more groups than an LDS replica holds (a baseline table: the group count is only known now): the caller plans
  // the step again without that member (packed route / partitioned family / row kernel, whatever applies)
  if (h_err[1] && t.kind == K_LDS_GROUPBY) return kRetryNoLds;
  if (h_err[1] && t.kind == K_IDX_PART) return kRetryNoIdx;
  if (h_err[1] && t.kind == K_BASELINE_FAST) {
    // the partitioned family ran out of spill space (extreme skew): redo the step with the
    // direct-atomic member of the same family
    HIP_TRY(hipMemsetAsync(t.d_err, 0, 64, s));
    HIP_TRY(launch_init_buffer(t.res->buf, q.entry_count, make_row_init(q), s));
    HIP_TRY(launch_baseline_fast(d, fv, t.res->buf, t.d_err, nullptr, 0, 0, 1, t.n_cus, s, &st));
    HIP_TRY(hipMemcpyAsync(h_err, t.d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  if (report) {
    std::memset(report, 0, sizeof(*report));
    std::snprintf(report->kernel_name, sizeof(report->kernel_name), "%s", st.kernel_name ? st.kernel_name : "");
    if (t.ev_start) (void)hipEventElapsedTime(&report->total_ms, t.ev_start, t.ev_stop);
    if (st.n_events_used > 0) {
      float acc = 0.f;
      for (int i = 0; i + 1 < st.n_events_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.ev_pool[i], t.ev_pool[i + 1]) == hipSuccess) acc += ms;
      }
      report->kernel_ms = acc;
    } else if (st.n_launches > 0 && t.nf > 0) {
      if (hipEventElapsedTime(&report->kernel_ms, st.k_start, st.k_stop) != hipSuccess) report->kernel_ms = 0.f;
    }
    if (!t.ev_start) report->total_ms = report->kernel_ms;  // (a one-kernel scan step: execute_impl)
    report->n_launches = st.n_launches;
    report->variant = st.variant;
    report->rows_scanned = t.total_rows;
    report->algorithmic_bytes = t.alg_bytes;
    report->spilled_rows = st.spilled_rows;
  }
  return h_err[0];
}

}  // namespace

struct mi355q_pending {
  TailState* tail = nullptr;  // null once the step has been finished
  int device_id = 0;
  int32_t code = MI355Q_OK;
  mi355q_exec_report rep{};
};

namespace {

// finishes the step a previous mi355q_execute_async left in flight on this device (ctx.mu held)
void drain_inflight(DeviceCtx& ctx) {
  mi355q_pending* p = ctx.inflight;
  if (!p) return;
  ctx.inflight = nullptr;
  if (p->tail) {
    p->code = finish_step(*p->tail, &p->rep);
    // the step was re-run after the call had returned: a consumer enqueued behind the first launches (pads, slices,
    // a collective) worked on the table the abandoned attempt left — tell the caller to redo it
    if (p->code == MI355Q_OK && p->tail->recomputed) p->code = MI355Q_STEP_RECOMPUTED;
    // the asynchronous call never selects the members that can ask for another plan (execute_impl: `!pend`), and the
    // result has been handed out, so planning again is not possible here: an internal code must not reach the caller
    if (p->code == kRetryNoLds || p->code == kRetryNoIdx || p->code == kNotTaken) p->code = MI355Q_ERR_UNSUPPORTED;
    delete p->tail;
    p->tail = nullptr;
  }
}

// reserved != nullptr: RESERVE mode — plan the step, size and allocate the per-device workspace it would use,
// launch nothing (mi355q_reserve_workspace); the column pointers are not looked at
int32_t execute_impl(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options* opts,
                     mi355q_result** out, mi355q_exec_report* report, mi355q_pending** pend,
                     int64_t* reserved = nullptr);

// Grouped steps whose aggregates read SEVERAL value columns, over inputs large enough for the partitioned / packed
// routes: one run per value column through the single-value families, zipped into the final layout
// (kernels_generic.hip k_zip_targets).  kNotTaken when the shape does not call for it.
int32_t execute_multi_value(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o,
                            const mi355q_qmd& q, const DevPlan& d, mi355q_result** out, mi355q_exec_report* report,
                            int64_t* reserved) {
  if (plan->n_group_cols < 1 || plan->join_outer_col >= 0 || q.slot_width != 8 || q.output_columnar || d.col0_key_quirk)
    return kNotTaken;
  // the value columns, in order of first use
  int vcols[MI355Q_MAX_TARGETS], n_v = 0;
  for (int t = 0; t < plan->n_targets; ++t) {
    const mi355q_target& tg = plan->targets[t];
    if (tg.agg == MI355Q_PROJECT_KEY || tg.col < 0) continue;
    if (tg.table != 0 || tg.agg == MI355Q_COUNT_IF || tg.agg == MI355Q_SUM_IF) return kNotTaken;
    bool seen = false;
    for (int k = 0; k < n_v; ++k) seen = seen || vcols[k] == tg.col;
    if (!seen) vcols[n_v++] = tg.col;
  }
  if (n_v < 2 || n_v > 4) return kNotTaken;
  int64_t total_rows = 0;
  for (int f = 0; f < in->n_frags; ++f) total_rows += in->num_rows[f];
  // small inputs: one pass of the row kernel is as good (kernel_variant 2 = "the large-input members", as for the
  // packed route: how the tests reach this route with small tables)
  if (o.kernel_variant != 2 && total_rows < ((int64_t)8 << 20)) return kNotTaken;

  if (reserved) {  // RESERVE / explain: the run over the first value column (every run has the same shape)
    char note[64];
    std::snprintf(note, sizeof(note), "%d runs (one per value column) + k_zip_targets, each", n_v);
    route_note(note);
    mi355q_plan sp = *plan;
    sp.n_targets = 0;
    for (int t = 0; t < plan->n_targets; ++t) {
      const mi355q_target& tg = plan->targets[t];
      if (tg.agg == MI355Q_PROJECT_KEY || tg.col < 0 || tg.col == vcols[0]) sp.targets[sp.n_targets++] = tg;
    }
    mi355q_exec_options o2 = o;
    o2.out_buffer = nullptr;
    return execute_impl(&sp, in, &o2, out, report, nullptr, reserved);
  }
  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  // the final table starts EMPTY (result_create_impl only allocates): k_zip_targets claims its keys with CAS
  HIP_TRY(launch_init_buffer(res->buf, q.entry_count, make_row_init(q), s));
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, 64));
  HIP_TRY(hipMemsetAsync(err.p, 0, 64, s));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  mi355q_exec_report acc{};
  for (int k = 0; k < n_v; ++k) {
    // run k: the targets over value column k; run 0 also carries COUNT(*) and the key projections
    mi355q_plan sp = *plan;
    sp.n_targets = 0;
    int orig_of[MI355Q_MAX_TARGETS];
    for (int t = 0; t < plan->n_targets; ++t) {
      const mi355q_target& tg = plan->targets[t];
      const bool valueless = tg.agg == MI355Q_PROJECT_KEY || tg.col < 0;
      if (valueless ? k == 0 : tg.col == vcols[k]) {
        orig_of[sp.n_targets] = t;
        sp.targets[sp.n_targets++] = tg;
      }
    }
    mi355q_exec_options o2 = o;
    o2.stream = s;
    o2.out_buffer = nullptr;
    mi355q_result* r2 = nullptr;
    mi355q_exec_report rep2{};
    if (int32_t e2 = mi355q_execute(&sp, in, &o2, &r2, &rep2)) return e2;
    struct R2Guard {
      mi355q_result* r;
      ~R2Guard() { if (r) mi355q_result_free(r); }
    } r2g{r2};
    const mi355q_qmd& q2 = r2->qmd;
    if (q2.desc_type != q.desc_type || q2.entry_count != q.entry_count || q2.slot_width != 8 || q2.output_columnar ||
        (q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH && q2.key_width != q.key_width))
      return MI355Q_ERR_UNSUPPORTED;
    int32_t src[MI355Q_MAX_SLOTS], dst[MI355Q_MAX_SLOTS];
    int n = 0;
    for (int t2 = 0; t2 < sp.n_targets; ++t2) {
      const int t = orig_of[t2];
      const int ss = q2.target_slot[t2], sf = q.target_slot[t];
      if ((ss < 0) != (sf < 0)) return MI355Q_ERR_UNSUPPORTED;   // (a projection read from the key columns on both sides)
      if (ss < 0) continue;
      const int ns = plan->targets[t].agg == MI355Q_AVG ? 2 : 1;
      for (int j = 0; j < ns; ++j) {
        if (q2.init_vals[ss + j] != q.init_vals[sf + j]) return MI355Q_ERR_UNSUPPORTED;
        src[n] = ss + j;
        dst[n] = sf + j;
        ++n;
      }
    }
    HIP_TRY(launch_zip_targets(res->dplan, r2->dplan, q2.idx_target_as_key, r2->buf, res->buf, src, dst, n, (int32_t*)err.p, s));
    HIP_TRY(hipStreamSynchronize(s));   // r2 is freed at the end of this iteration
    if (k == 0) {
      std::snprintf(acc.kernel_name, sizeof(acc.kernel_name), "%s", rep2.kernel_name);
      acc.variant = rep2.variant;
    }
    acc.kernel_ms += rep2.kernel_ms;
    acc.n_launches += rep2.n_launches;
    acc.spilled_rows += rep2.spilled_rows;
  }
  if (ev1) HIP_TRY(hipEventRecord(ev1, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(h_err), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_err) return h_err;
  if (report) {
    *report = acc;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}

// Grouped steps over fact JOIN dim (SURVEY f2 + GROUP BY: `SELECT f.g, SUM(d.w), COUNT(*) FROM f JOIN d ON f.k = d.k GROUP BY
// f.g`) with a ONE-TO-ONE join table over a large outer input.  In the row kernel every row probes the table AND updates its
// group with device atomics; here the probe is its own pass (k_join_gather): per outer row, the inner columns the aggregates
// read become dense temporary OUTER columns — the column's NULL where the row has no match — plus, for INNER joins, a 0 / 1
// "matched" column the derived step filters on.  The derived plan has no join, the same targets over those columns and the
// SAME layout (asserted: the descriptor of the derived plan must equal the stated one bit for bit), so its result is the
// step's result and every grouped family applies (LDS members, the partitioned families).  One-to-many tables change the
// row multiplicity and stay in the row kernel.  kNotTaken when the shape does not call for it.
int32_t execute_join_gather(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o, const mi355q_qmd& q,
                            const DevPlan& d, int n_cus, mi355q_result** out, mi355q_exec_report* report, int64_t* reserved) {
  if (plan->join_outer_col < 0 || plan->n_group_cols < 1 || plan->n_exprs != 0 || o.kernel_variant == 1 || o.force_generic ||
      (d.join_hash_type != 0 && d.join_hash_type != 1) || q.desc_type == MI355Q_NON_GROUPED_AGGREGATE)
    return kNotTaken;
  const int nc = plan->n_cols, nf = in->n_frags;
  const bool inner_join = plan->join_kind != MI355Q_JOIN_LEFT;
  int64_t total_rows = 0, max_frag_rows = 0;
  for (int f = 0; f < nf; ++f) {
    if (in->num_rows[f] < 0) return kNotTaken;
    total_rows += in->num_rows[f];
    max_frag_rows = std::max(max_frag_rows, in->num_rows[f]);
  }
  // (kernel_variant 2 = "the large-input members": how the tests reach this route with small tables)
  if (nf < 1 || (o.kernel_variant != 2 && total_rows < kIdxPartMinRows)) return kNotTaken;
  // a perfect-hash table of <= 64 KB is aggregated by the row kernel in a per-workgroup LDS copy (launch_generic): probe and
  // update in one pass, nothing gained by splitting them — measured at 1 B rows, 100 groups: 49 ms there, 57 ms here; 10 000
  // groups (global atomics there): 470 / 163 ms there (LEFT / INNER), 69 / 64 ms here (profiles/r04_grouped_join_1b_call16/17.jsonl)
  if (o.kernel_variant != 2 && q.desc_type == MI355Q_GROUP_BY_PERFECT_HASH && q.entry_count * (int64_t)q.row_size <= 64 * 1024)
    return kNotTaken;
  // the inner columns the targets read, in order of first use
  int32_t used[MI355Q_MAX_COLS], dst[MI355Q_MAX_COLS], width[MI355Q_MAX_COLS];
  int64_t null_pat[MI355Q_MAX_COLS];
  int n_used = 0;
  mi355q_plan p2 = *plan;
  for (int t = 0; t < plan->n_targets; ++t) {
    const mi355q_target& tg = plan->targets[t];
    if (tg.agg == MI355Q_PROJECT_KEY || tg.table == 0 || tg.col < 0) continue;
    if (tg.col >= plan->n_inner_cols) return kNotTaken;
    int j = 0;
    while (j < n_used && used[j] != tg.col) ++j;
    if (j == n_used) {
      const mi355q_col_desc& cd = plan->inner_cols[tg.col];
      if (cd.encoding != MI355Q_ENC_NONE || (cd.logical_type != 0 && cd.logical_type != cd.type)) return kNotTaken;
      if (nc + n_used + 1 + (inner_join ? 1 : 0) > MI355Q_MAX_COLS) return kNotTaken;
      used[n_used] = tg.col;
      dst[n_used] = nc + n_used;
      width[n_used] = plain_width(cd.type);
      null_pat[n_used] = cd.type == MI355Q_DOUBLE ? kNullDoubleBits
                         : cd.type == MI355Q_FLOAT ? (int64_t)(uint32_t)kNullFloatBits : plain_int_null(cd.type);
      p2.cols[nc + n_used] = cd;
      // (nullable under an outer join, resolve_targets' rule; the RANGE stays the column's own, as the layout decisions read it)
      if (!inner_join) p2.cols[nc + n_used].nullable = 1;
      p2.col_ranges[nc + n_used] = plan->inner_col_ranges[tg.col];
      ++n_used;
    }
    p2.targets[t].table = 0;
    p2.targets[t].col = nc + j;
  }
  int flag_col = -1;
  if (inner_join) {
    if (plan->n_quals >= MI355Q_MAX_QUALS) return kNotTaken;
    flag_col = nc + n_used;
    p2.cols[flag_col] = mi355q_col_desc{MI355Q_INT32, 0, MI355Q_ENC_NONE, 0};  // (INT32: the fast families' range filters read INT32 / INT64)
    p2.col_ranges[flag_col] = mi355q_range{1, 0, 0, 1, 0.0, 0.0, 0};
    mi355q_qual& mq = p2.quals[p2.n_quals++];
    mq = mi355q_qual{};
    mq.col = flag_col;
    mq.op = MI355Q_EQ;
    mq.ival = 1;
  }
  const int nc2 = nc + n_used + (inner_join ? 1 : 0);
  p2.n_cols = nc2;
  p2.join_outer_col = -1;
  p2.join_table = nullptr;
  p2.n_join_cols = 0;
  p2.join_kind = MI355Q_JOIN_INNER;
  p2.n_inner_cols = 0;
  mi355q_qmd q2;
  if (qmd_init(p2, &q2) != MI355Q_OK || std::memcmp(&q, &q2, sizeof(q)) != 0) return kNotTaken;  // the same layout, or not this way
  int64_t row_bytes = inner_join ? 4 : 0;
  for (int j = 0; j < n_used; ++j) row_bytes += width[j];
  mi355q_exec_options o2 = o;
  if (reserved) {
    route_note("k_join_gather (inner columns + matched flag as outer columns)");
    // the derived step's shape: the same fragments with nc2 columns (the pointers are not looked at in reserve mode)
    std::vector<const void*> fake((size_t)nf * nc2, nullptr);
    mi355q_inputs in2 = *in;
    in2.col_buffers = fake.data();
    in2.inner_col_buffers = nullptr;
    return execute_impl(&p2, &in2, &o2, out, report, nullptr, reserved);
  }
  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const int64_t budget = std::min<int64_t>((int64_t)16 << 30, ((int64_t)free_b + ctx.gather_bytes) / 3);
  const int64_t pad = 16 * (int64_t)(nc2 - nc);  // every (fragment, column) chunk starts on a 16-byte boundary
  int64_t pass_rows = std::max<int64_t>(budget / std::max<int64_t>(row_bytes, 1), max_frag_rows);
  if (o.pass_rows > 0) pass_rows = std::max<int64_t>(o.pass_rows, max_frag_rows);  // tests: several passes
  if (pass_rows > total_rows) pass_rows = total_rows;
  const int64_t tab_bytes = ((int64_t)sizeof(void*) * nf * nc2 + 255) & ~255ll;
  const int64_t rows_bytes = ((int64_t)sizeof(int64_t) * nf + 255) & ~255ll;
  const int64_t col_region = ((pass_rows * row_bytes + pad * nf) + 255) & ~255ll;
  const int64_t need = col_region + tab_bytes + rows_bytes + 256;
  if (ctx.gather_bytes < need) {
    if (ctx.gather) (void)hipFree(ctx.gather);
    ctx.gather = nullptr;
    ctx.gather_bytes = 0;
    if (hipMalloc(&ctx.gather, (size_t)need) != hipSuccess) {
      (void)hipGetLastError();
      return kNotTaken;  // (the row kernel needs no temporary columns)
    }
    ctx.gather_bytes = need;
  }
  char* base = (char*)ctx.gather;
  const int8_t** d_tab = (const int8_t**)(base + col_region);
  int64_t* d_rows = (int64_t*)(base + col_region + tab_bytes);
  HIP_TRY(hipMemcpyAsync(d_rows, in->num_rows, sizeof(int64_t) * (size_t)nf, hipMemcpyHostToDevice, s));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  std::vector<const void*> cols2((size_t)nf * nc2);
  mi355q_result* res = nullptr;
  struct ResGuard {
    mi355q_result*& r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  mi355q_exec_report acc{};
  int pass = 0, f = 0;
  while (f < nf) {
    int f1 = f;
    int64_t rows = 0, off = 0;
    while (f1 < nf && (f1 == f || rows + in->num_rows[f1] <= pass_rows)) {
      for (int c = 0; c < nc; ++c) cols2[(size_t)(f1 - f) * nc2 + c] = in->col_buffers[(size_t)f1 * nc + c];
      for (int k = nc; k < nc2; ++k) {
        cols2[(size_t)(f1 - f) * nc2 + k] = base + off;
        const int w = k == flag_col ? 4 : width[k - nc];
        off += (in->num_rows[f1] * w + 15) & ~15ll;
      }
      rows += in->num_rows[f1];
      ++f1;
    }
    if (off > col_region) return MI355Q_ERR_OUT_OF_GPU_MEM;  // (cannot happen: the region is sized for it)
    const int pnf = f1 - f;
    HIP_TRY(hipMemcpyAsync(d_tab, cols2.data(), sizeof(void*) * (size_t)pnf * nc2, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_join_gather(d, n_used, used, dst, width, null_pat, flag_col, nc2, d_tab, d_rows + f, pnf, max_frag_rows, n_cus, s));
    HIP_TRY(hipStreamSynchronize(s));  // (cols2 is re-used by the next pass; the step below synchronises anyway)
    mi355q_inputs in2 = *in;
    in2.n_frags = pnf;
    in2.col_buffers = cols2.data();
    in2.num_rows = in->num_rows + f;
    in2.inner_col_buffers = nullptr;
    o2 = o;
    o2.stream = s;
    o2.out_buffer = pass == 0 ? o.out_buffer : nullptr;
    mi355q_result* r2 = nullptr;
    mi355q_exec_report rep2{};
    if (int32_t e2 = mi355q_execute(&p2, &in2, &o2, &r2, &rep2)) return e2;
    if (pass == 0) {
      res = r2;
      std::snprintf(acc.kernel_name, sizeof(acc.kernel_name), "%s", rep2.kernel_name);
      acc.variant = rep2.variant;
    } else {
      const int32_t er = mi355q_result_reduce(res, r2, s);
      mi355q_result_free(r2);
      if (er) return er;
    }
    acc.kernel_ms += rep2.kernel_ms;
    acc.n_launches += rep2.n_launches + 1;
    acc.spilled_rows += rep2.spilled_rows;
    f = f1;
    ++pass;
  }
  if (ev1) HIP_TRY(hipEventRecord(ev1, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (report) {
    *report = acc;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  *out = res;
  res = nullptr;
  return MI355Q_OK;
}

// Baseline steps over 2 - 3 plain INT key columns that all have ranges — the reference's PerfectHashMultiCol / MultiStep
// shapes beyond g_baseline_groupby_threshold (PHM006, MSPHM005, MSPHM007: ~11 M combinations, Execute.cpp:113,
// GroupByAndAggregate.cpp:232-365) — over a large input: the product of the ranges still indexes a table the
// index-partitioned family (kernels_idx.hip) aggregates in one exchange, so the step runs on a library-owned PERFECT
// twin of the layout and its live entries are re-keyed into the baseline table of the stated plan
// (kernels_generic.hip k_perfect_twin_emit).  kNotTaken when the shape does not call for it.
constexpr int64_t kTwinMaxEntries = (int64_t)32 << 20;
int32_t execute_perfect_twin(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o, const mi355q_qmd& q,
                             int n_cus, mi355q_result** out, mi355q_exec_report* report, int64_t* reserved) {
  if (q.desc_type != MI355Q_GROUP_BY_BASELINE_HASH || plan->n_group_cols < 2 || plan->n_group_cols > 3 ||
      plan->join_outer_col >= 0 || plan->n_exprs != 0 || q.slot_width != 8 || q.output_columnar ||
      plan->output_columnar_hint != 0 || o.kernel_variant == 1 || o.force_generic)
    return kNotTaken;
  __int128 card = 1;
  int32_t translate[MI355Q_MAX_GROUP_COLS], key_type[MI355Q_MAX_GROUP_COLS];
  for (int g = 0; g < plan->n_group_cols; ++g) {
    const int c = plan->group_cols[g];
    if (c < 0 || c >= plan->n_cols) return kNotTaken;
    const mi355q_col_desc& cd = plan->cols[c];
    const mi355q_range& r = plan->col_ranges[c];
    if (cd.type != MI355Q_INT32 || cd.encoding != MI355Q_ENC_NONE || (cd.logical_type != 0 && cd.logical_type != cd.type) ||
        !r.valid || r.bucket != 0 || r.min > r.max || (r.has_nulls && !cd.nullable))
      return kNotTaken;
    card *= (__int128)r.max - (__int128)r.min + 1 + (r.has_nulls ? 1 : 0);
    if (card > (__int128)kTwinMaxEntries) return kNotTaken;
    translate[g] = cd.nullable && r.has_nulls;  // (build_dev_plan's rule for several key columns)
    key_type[g] = col_type_code(cd);
  }
  int64_t total_rows = 0, max_rows = 0;
  for (int f = 0; f < in->n_frags; ++f) {
    total_rows += in->num_rows[f];
    max_rows = std::max(max_rows, in->num_rows[f]);
  }
  // (kernel_variant 2 = "the large-input members": how the tests reach this route with small tables)
  if (o.kernel_variant != 2 && total_rows < kIdxPartMinRows) return kNotTaken;
  PerfectTwinScope twin(kTwinMaxEntries);
  mi355q_qmd q2;
  if (qmd_init(*plan, &q2) != MI355Q_OK) return kNotTaken;
  if (q2.desc_type != MI355Q_GROUP_BY_PERFECT_HASH || q2.slot_width != 8 || q2.output_columnar ||
      q2.entry_count * (int64_t)q2.row_size > ((int64_t)4 << 30))
    return kNotTaken;
  DevPlan d2;
  if (build_dev_plan(*plan, q2, &d2) != MI355Q_OK) return kNotTaken;
  {
    FragView fvh{nullptr, nullptr, in->col_buffers, in->num_rows, in->n_frags, plan->n_cols, total_rows, max_rows};
    if (!idx_part_eligible(d2, fvh, n_cus)) return kNotTaken;
  }
  for (int t = 0; t < plan->n_targets; ++t) {
    const int sf = q.target_slot[t], ss = q2.target_slot[t];
    if (plan->targets[t].agg == MI355Q_PROJECT_KEY) {
      if (sf >= 0) return kNotTaken;  // (baseline: projections are read from the key columns)
      continue;
    }
    if (sf < 0 || ss < 0) return kNotTaken;
    for (int j = 0; j < (plan->targets[t].agg == MI355Q_AVG ? 2 : 1); ++j)
      if (q.init_vals[sf + j] != q2.init_vals[ss + j]) return kNotTaken;
  }
  mi355q_exec_options o2 = o;
  o2.out_buffer = nullptr;
  if (reserved) {
    route_note("perfect-hash twin + k_perfect_twin_emit");
    return execute_impl(plan, in, &o2, out, report, nullptr, reserved);
  }
  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  o2.stream = s;
  mi355q_result* r2 = nullptr;
  mi355q_exec_report rep2{};
  if (int32_t e2 = mi355q_execute(plan, in, &o2, &r2, &rep2)) {
    if (e2 == MI355Q_ERR_UNSUPPORTED || e2 == MI355Q_ERR_OUT_OF_GPU_MEM || e2 < 0) return kNotTaken;
    return e2;
  }
  struct R2Guard {
    mi355q_result* r;
    ~R2Guard() { if (r) mi355q_result_free(r); }
  } r2g{r2};
  if (r2->qmd.desc_type != MI355Q_GROUP_BY_PERFECT_HASH || r2->qmd.slot_width != 8 || r2->qmd.output_columnar ||
      r2->qmd.entry_count != q2.entry_count)
    return kNotTaken;
  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  HIP_TRY(launch_init_buffer(res->buf, q.entry_count, make_row_init(q), s));
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, 64));
  HIP_TRY(hipMemsetAsync(err.p, 0, 64, s));
  HIP_TRY(launch_perfect_twin_emit(res->dplan, r2->dplan, r2->qmd.idx_target_as_key, plan->n_group_cols, translate, key_type,
                                   q2.group_min, q2.group_card, q2.group_null_key, r2->buf, res->buf, (int32_t*)err.p, s));
  if (ev1) HIP_TRY(hipEventRecord(ev1, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(h_err), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_err) return h_err;
  if (report) {
    *report = rep2;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->n_launches = rep2.n_launches + 1;
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}

// Baseline steps whose key columns lie on a LATTICE — key = min + stride x i — with few lattice points: the reference
// benchmark's BIGINT columns x10k_s10k / x100k_s10k / x1m_s10k (multiples of 10 000: BaselineHash/BH007-010,
// MultiStep/MSBS006-007).  Their RANGE is far too wide for a perfect hash (GroupByAndAggregate.cpp:232-365 sees max - min),
// so the reference — and the plain route here — hash 8-byte keys.  The stride is found on the first fragment (k_key_gcd), every
// row's key is then rewritten as its lattice index i in a dense INT32 column and CHECKED (k_affine_keys: a key off the
// lattice or outside the range gives the route up, nothing is assumed about the data), the step runs grouped by those
// columns on a library-owned perfect-hash twin (typed LDS members / the index-partitioned family: ONE exchange of 8- or
// 16-byte records instead of 16-byte records per value column), and k_affine_twin_emit re-keys the twin's entries into the
// baseline table of the stated plan.  mi355q_explain cannot see the data and names the plain route for these shapes.
int32_t execute_affine_twin(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o, const mi355q_qmd& q,
                            int n_cus, mi355q_result** out, mi355q_exec_report* report, int64_t* reserved) {
  if (reserved || q.desc_type != MI355Q_GROUP_BY_BASELINE_HASH || plan->n_group_cols < 1 || plan->n_group_cols > 3 ||
      plan->join_outer_col >= 0 || plan->n_exprs != 0 || q.slot_width != 8 || q.output_columnar ||
      plan->output_columnar_hint != 0 || o.kernel_variant == 1 || o.force_generic)
    return kNotTaken;
  const int nc = plan->n_cols, nf = in->n_frags, ng = plan->n_group_cols;
  int64_t total_rows = 0, max_frag_rows = 0;
  int first = -1;
  for (int f = 0; f < nf; ++f) {
    if (in->num_rows[f] < 0) return kNotTaken;
    total_rows += in->num_rows[f];
    max_frag_rows = std::max(max_frag_rows, in->num_rows[f]);
    if (first < 0 && in->num_rows[f] > 0) first = f;
  }
  // (kernel_variant 2 = "the large-input members": how the tests reach this route with small tables)
  if (first < 0 || (o.kernel_variant != 2 && total_rows < kIdxPartMinRows)) return kNotTaken;
  bool proj[MI355Q_MAX_GROUP_COLS];
  int n_proj = 0;
  for (int g = 0; g < ng; ++g) {
    const int c = plan->group_cols[g];
    if (c < 0 || c >= nc) return kNotTaken;
    const mi355q_col_desc& cd = plan->cols[c];
    const mi355q_range& r = plan->col_ranges[c];
    if ((cd.type != MI355Q_INT32 && cd.type != MI355Q_INT64) || cd.encoding != MI355Q_ENC_NONE ||
        (cd.logical_type != 0 && cd.logical_type != cd.type) || !r.valid || r.bucket != 0 || r.min > r.max ||
        (r.has_nulls && !cd.nullable))
      return kNotTaken;
    // BIGINT keys, and INT keys over a range too wide for the twin on their own, are looked at for a stride
    proj[g] = cd.type == MI355Q_INT64 || (__int128)r.max - (__int128)r.min >= ((__int128)1 << 20);
    n_proj += proj[g] ? 1 : 0;
  }
  if (n_proj == 0 || nc + n_proj > MI355Q_MAX_COLS) return kNotTaken;
  // only where the twin's step is the typed LDS members' or the index-partitioned family's work (no qual, plain INT value
  // columns: checked here before anything is launched, and on the derived plan below): every other baseline step — the
  // headline's filtered AVG(double) among them — keeps its own family and pays nothing for this route
  // (a compiled filter travelling beside the plan is a qual too: the `rest` plan of execute_bool_filter has n_quals = 0;
  // ADVICE r05)
  if (plan->n_quals != 0 || step_bool_filter()) return kNotTaken;
  for (int t = 0; t < plan->n_targets; ++t) {
    const mi355q_target& tg = plan->targets[t];
    if (tg.agg == MI355Q_PROJECT_KEY || tg.col < 0) continue;
    if (tg.table != 0 || tg.col >= nc || tg.agg == MI355Q_COUNT_IF || tg.agg == MI355Q_SUM_IF) return kNotTaken;
    const mi355q_col_desc& vd = plan->cols[tg.col];
    if (vd.type != MI355Q_INT32 || vd.encoding != MI355Q_ENC_NONE || (vd.logical_type != 0 && vd.logical_type != vd.type)) return kNotTaken;
  }

  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  // ---- the stride of every looked-at key column, from the first non-empty fragment
  // (a sample: the first 4 M rows — the stride it gives is verified on every row below, a coarser lattice only gives the route up)
  constexpr int kGcdBlocks = 256;  // partial strides per column handed to the host
  constexpr int64_t kGcdScratch = 64 * 256;
  DevWord gw;
  HIP_TRY(hipMalloc(&gw.p, sizeof(unsigned long long) * (kGcdScratch + kGcdBlocks * MI355Q_MAX_GROUP_COLS) + 64));
  unsigned long long* g_scr = (unsigned long long*)gw.p;
  unsigned long long* g_out = g_scr + kGcdScratch;
  std::vector<unsigned long long> h_g((size_t)kGcdBlocks * MI355Q_MAX_GROUP_COLS);
  for (int gcol = 0, k = 0; gcol < ng; ++gcol) {
    if (!proj[gcol]) continue;
    const int c = plan->group_cols[gcol];
    HIP_TRY(launch_key_gcd(in->col_buffers[(size_t)first * nc + c], plain_width(plan->cols[c].type),
                           std::min<int64_t>(in->num_rows[first], (int64_t)4 << 20), plan->col_ranges[c].min, plan->cols[c].nullable, g_scr,
                           g_out + (size_t)k * kGcdBlocks, s));
    ++k;
  }
  HIP_TRY(hipMemcpyAsync(h_g.data(), g_out, sizeof(unsigned long long) * kGcdBlocks * n_proj, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  // ---- the derived plan
  mi355q_plan p2 = *plan;
  int32_t src_col[MI355Q_MAX_GROUP_COLS], dst_col[MI355Q_MAX_GROUP_COLS], width[MI355Q_MAX_GROUP_COLS], nullable[MI355Q_MAX_GROUP_COLS];
  int64_t kmin[MI355Q_MAX_GROUP_COLS], stride[MI355Q_MAX_GROUP_COLS], card[MI355Q_MAX_GROUP_COLS];
  int64_t base_of[MI355Q_MAX_GROUP_COLS], stride_of[MI355Q_MAX_GROUP_COLS];
  int32_t translate[MI355Q_MAX_GROUP_COLS], key_type[MI355Q_MAX_GROUP_COLS];
  __int128 entries = 1;
  for (int gcol = 0, k = 0; gcol < ng; ++gcol) {
    const int c = plan->group_cols[gcol];
    const mi355q_col_desc& cd = plan->cols[c];
    const mi355q_range& r = plan->col_ranges[c];
    translate[gcol] = cd.nullable && (ng == 1 || r.has_nulls);  // (build_dev_plan's rule for the twin's keys)
    key_type[gcol] = col_type_code(cd);
    base_of[gcol] = r.min;
    stride_of[gcol] = 1;
    __int128 points = (__int128)r.max - (__int128)r.min + 1;
    if (proj[gcol]) {
      unsigned long long gg = 0;
      for (int b = 0; b < kGcdBlocks; ++b) {
        unsigned long long x = h_g[(size_t)k * kGcdBlocks + b], y = gg;
        while (y) {
          const unsigned long long t = x % y;
          x = y;
          y = t;
        }
        gg = x;
      }
      if (gg == 0) gg = 1;  // (every key of the fragment is the minimum, or NULL)
      if (gg > (unsigned long long)INT64_MAX) return kNotTaken;
      points = ((__int128)r.max - (__int128)r.min) / (__int128)gg + 1;
      if (points >= ((__int128)1 << 30)) return kNotTaken;
      stride_of[gcol] = (int64_t)gg;
      src_col[k] = c;
      dst_col[k] = nc + k;
      width[k] = plain_width(cd.type);
      nullable[k] = cd.nullable;
      kmin[k] = r.min;
      stride[k] = (int64_t)gg;
      card[k] = (int64_t)points;
      p2.cols[nc + k] = mi355q_col_desc{MI355Q_INT32, cd.nullable, MI355Q_ENC_NONE, 0};
      p2.col_ranges[nc + k] = mi355q_range{1, r.has_nulls, 0, (int64_t)points - 1, 0.0, 0.0, 0};
      p2.group_cols[gcol] = nc + k;
      ++k;
    } else if (points >= ((__int128)1 << 30) || r.min <= -((int64_t)1 << 30) || r.min >= ((int64_t)1 << 30)) {
      return kNotTaken;
    }
    entries *= points + (r.has_nulls ? 1 : 0);
    if (entries > (__int128)kTwinMaxEntries) return kNotTaken;
  }
  // one key, one value column, >= 512 K lattice points: the 16-byte partitioned family does this shape as well as the twin's
  // two extra passes allow (BH009 at 1 B rows: 8.6 ms there, 9.9 ms here; 100 K points, BH008: 12.5 -> 8.2 ms;
  // profiles/r04_refbench_lattice_keys_call21.jsonl)
  {
    int vcols[MI355Q_MAX_TARGETS], n_v = 0;
    for (int t = 0; t < plan->n_targets; ++t) {
      const mi355q_target& tg = plan->targets[t];
      if (tg.agg == MI355Q_PROJECT_KEY || tg.col < 0) continue;
      bool seen = false;
      for (int k = 0; k < n_v; ++k) seen = seen || vcols[k] == tg.col;
      if (!seen) vcols[n_v++] = tg.col;
    }
    if (ng == 1 && n_v <= 1 && entries >= ((__int128)1 << 19) && o.kernel_variant != 2) return kNotTaken;
  }
  const int nc2 = nc + n_proj;
  p2.n_cols = nc2;
  PerfectTwinScope twin(kTwinMaxEntries);
  mi355q_qmd q2;
  if (qmd_init(p2, &q2) != MI355Q_OK) return kNotTaken;
  if (q2.desc_type != MI355Q_GROUP_BY_PERFECT_HASH || q2.slot_width != 8 || q2.output_columnar ||
      q2.entry_count * (int64_t)q2.row_size > ((int64_t)4 << 30))
    return kNotTaken;
  for (int t = 0; t < plan->n_targets; ++t) {
    const int sf = q.target_slot[t], ss = q2.target_slot[t];
    if (plan->targets[t].agg == MI355Q_PROJECT_KEY) {
      if (sf >= 0) return kNotTaken;  // (baseline: projections are read from the key columns)
      continue;
    }
    if (sf < 0 || ss < 0) return kNotTaken;
    for (int j = 0; j < (plan->targets[t].agg == MI355Q_AVG ? 2 : 1); ++j)
      if (q.init_vals[sf + j] != q2.init_vals[ss + j]) return kNotTaken;
  }
  {  // the twin's step must be one of the two families this route is for
    DevPlan d2;
    if (build_dev_plan(p2, q2, &d2) != MI355Q_OK) return kNotTaken;
    std::vector<const void*> shape((size_t)nf * nc2);
    for (int f = 0; f < nf; ++f) {
      for (int c = 0; c < nc; ++c) shape[(size_t)f * nc2 + c] = in->col_buffers[(size_t)f * nc + c];
      for (int k = 0; k < n_proj; ++k) shape[(size_t)f * nc2 + nc + k] = (const void*)(uintptr_t)256;  // (16-byte aligned, like the real chunks)
    }
    FragView fvh{nullptr, nullptr, shape.data(), in->num_rows, nf, nc2, total_rows, max_frag_rows};
    if (!lds_groupby_eligible(d2, fvh, n_cus) && !idx_part_eligible(d2, fvh, n_cus)) return kNotTaken;
  }
  // ---- passes: lattice indices of a pass of fragments, the twin step on them
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const int64_t budget = std::min<int64_t>((int64_t)16 << 30, ((int64_t)free_b + ctx.lattice_bytes) / 3);
  const int64_t row_bytes = 4 * (int64_t)n_proj, pad = 16 * (int64_t)n_proj;
  int64_t pass_rows = std::max<int64_t>(budget / row_bytes, max_frag_rows);
  if (o.pass_rows > 0) pass_rows = std::max<int64_t>(o.pass_rows, max_frag_rows);  // tests: several passes
  if (pass_rows > total_rows) pass_rows = total_rows;
  const int64_t tab_bytes = ((int64_t)sizeof(void*) * nf * nc2 + 255) & ~255ll;
  const int64_t rows_bytes = ((int64_t)sizeof(int64_t) * nf + 255) & ~255ll;
  const int64_t col_region = ((pass_rows * row_bytes + pad * nf) + 255) & ~255ll;
  const int64_t need = col_region + tab_bytes + rows_bytes + 256;
  if (ctx.lattice_bytes < need) {
    if (ctx.lattice) (void)hipFree(ctx.lattice);
    ctx.lattice = nullptr;
    ctx.lattice_bytes = 0;
    if (hipMalloc(&ctx.lattice, (size_t)need) != hipSuccess) {
      (void)hipGetLastError();
      return kNotTaken;
    }
    ctx.lattice_bytes = need;
  }
  char* base = (char*)ctx.lattice;
  const int8_t** d_tab = (const int8_t**)(base + col_region);
  int64_t* d_rows = (int64_t*)(base + col_region + tab_bytes);
  int32_t* d_flag = (int32_t*)(base + col_region + tab_bytes + rows_bytes);
  HIP_TRY(hipMemsetAsync(d_flag, 0, 64, s));
  HIP_TRY(hipMemcpyAsync(d_rows, in->num_rows, sizeof(int64_t) * (size_t)nf, hipMemcpyHostToDevice, s));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  std::vector<const void*> cols2((size_t)nf * nc2);
  mi355q_result* tw = nullptr;  // the twin table (all passes folded)
  struct TwGuard {
    mi355q_result*& r;
    ~TwGuard() { if (r) mi355q_result_free(r); }
  } twg{tw};
  mi355q_exec_report acc{};
  int pass = 0, f = 0;
  while (f < nf) {
    int f1 = f;
    int64_t rows = 0, off = 0;
    while (f1 < nf && (f1 == f || rows + in->num_rows[f1] <= pass_rows)) {
      for (int c = 0; c < nc; ++c) cols2[(size_t)(f1 - f) * nc2 + c] = in->col_buffers[(size_t)f1 * nc + c];
      for (int k = 0; k < n_proj; ++k) {
        cols2[(size_t)(f1 - f) * nc2 + nc + k] = base + off;
        off += (in->num_rows[f1] * 4 + 15) & ~15ll;
      }
      rows += in->num_rows[f1];
      ++f1;
    }
    if (off > col_region) return MI355Q_ERR_OUT_OF_GPU_MEM;  // (cannot happen: the region is sized for it)
    const int pnf = f1 - f;
    HIP_TRY(hipMemcpyAsync(d_tab, cols2.data(), sizeof(void*) * (size_t)pnf * nc2, hipMemcpyHostToDevice, s));
    HIP_TRY(launch_affine_keys(n_proj, src_col, dst_col, width, nullable, kmin, stride, card, nc2, d_tab, d_rows + f, pnf, max_frag_rows,
                               d_flag, n_cus, s));
    int32_t h_flag = 0;
    HIP_TRY(hipMemcpyAsync(&h_flag, d_flag, sizeof(h_flag), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h_flag) return kNotTaken;  // a key off the lattice of the first fragment: the plain route
    mi355q_inputs in2 = *in;
    in2.n_frags = pnf;
    in2.col_buffers = cols2.data();
    in2.num_rows = in->num_rows + f;
    mi355q_exec_options o2 = o;
    o2.stream = s;
    o2.out_buffer = nullptr;
    mi355q_result* r2 = nullptr;
    mi355q_exec_report rep2{};
    if (int32_t e2 = mi355q_execute(&p2, &in2, &o2, &r2, &rep2)) {
      if (e2 == MI355Q_ERR_UNSUPPORTED || e2 == MI355Q_ERR_OUT_OF_GPU_MEM || e2 < 0 || e2 == MI355Q_ERR_OUT_OF_SLOTS) return kNotTaken;
      return e2;
    }
    if (r2->qmd.desc_type != MI355Q_GROUP_BY_PERFECT_HASH || r2->qmd.slot_width != 8 || r2->qmd.output_columnar ||
        r2->qmd.entry_count != q2.entry_count) {
      mi355q_result_free(r2);
      return kNotTaken;
    }
    if (pass == 0) {
      tw = r2;
      std::snprintf(acc.kernel_name, sizeof(acc.kernel_name), "%s", rep2.kernel_name);
      acc.variant = rep2.variant;
    } else {
      const int32_t er = mi355q_result_reduce(tw, r2, s);
      mi355q_result_free(r2);
      if (er) return er;
    }
    acc.kernel_ms += rep2.kernel_ms;
    acc.n_launches += rep2.n_launches + 1;
    acc.spilled_rows += rep2.spilled_rows;
    f = f1;
    ++pass;
  }
  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  HIP_TRY(launch_init_buffer(res->buf, q.entry_count, make_row_init(q), s));
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, 64));
  HIP_TRY(hipMemsetAsync(err.p, 0, 64, s));
  HIP_TRY(launch_affine_twin_emit(res->dplan, tw->dplan, tw->qmd.idx_target_as_key, ng, translate, key_type, q2.group_min, q2.group_card,
                                  q2.group_null_key, base_of, stride_of, tw->buf, res->buf, (int32_t*)err.p, s));
  if (ev1) HIP_TRY(hipEventRecord(ev1, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(h_err), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_err) return h_err;
  if (report) {
    *report = acc;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->n_launches = acc.n_launches + 1;
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}

// GROUP BY CAST(<plain integer column> AS DOUBLE | FLOAT) — the reference benchmark's BaselineHash and MultiStep
// BaselineHash shapes (Benchmarks/synthetic_benchmark/queries/BaselineHash/BH001-006.sql, MultiStep/MSBS001-005.sql).  A
// floating-point key always takes the baseline layout (GroupByAndAggregate.cpp:232-365: getExprRangeInfo is FloatingPoint),
// but the groups ARE the integer column's values: the step runs on a derived plan that groups by the integer column
// itself (perfect hash: the LDS / partitioned families, no key expression to project), and its entries are then re-keyed
// with the cast value and merged into the baseline table of the stated plan (kernels_generic.hip k_cast_key_emit).
// kNotTaken when the shape does not call for it.
// Does an expression of the plan read the value of another one (MI355Q_EX_COL with arg >= n_cols)?  The derived-plan routes
// below take expressions out of the plan and renumber the rest: they leave such plans to the projection.
bool exprs_read_exprs(const mi355q_plan& p) {
  for (int k = 0; k < p.n_exprs && k < MI355Q_MAX_EXPRS; ++k)
    for (int i = 0; i < p.exprs[k].n_nodes && i < MI355Q_MAX_EXPR_NODES; ++i)
      if (p.exprs[k].nodes[i].op == MI355Q_EX_COL && p.exprs[k].nodes[i].arg >= p.n_cols) return true;
  return false;
}

int32_t execute_cast_key(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o,
                         mi355q_result** out, mi355q_exec_report* report, int64_t* reserved) {
  if (plan->n_group_cols != 1 || plan->join_outer_col >= 0 || plan->output_columnar_hint != 0 || o.kernel_variant == 1 ||
      o.force_generic || exprs_read_exprs(*plan))
    return kNotTaken;
  const int nc = plan->n_cols, gc = plan->group_cols[0];
  if (gc < nc || gc >= nc + plan->n_exprs) return kNotTaken;
  const mi355q_expr& ex = plan->exprs[gc - nc];
  if (ex.n_nodes != 2 || ex.nodes[0].op != MI355Q_EX_COL || ex.nodes[1].op != MI355Q_EX_CAST) return kNotTaken;
  const int to = ex.nodes[1].type, c = ex.nodes[0].arg;
  if ((to != MI355Q_DOUBLE && to != MI355Q_FLOAT) || c < 0 || c >= nc) return kNotTaken;
  const mi355q_col_desc& cd = plan->cols[c];
  const mi355q_range& cr = plan->col_ranges[c];
  if ((cd.type != MI355Q_INT32 && cd.type != MI355Q_INT64) || cd.encoding != MI355Q_ENC_NONE ||
      (cd.logical_type != 0 && cd.logical_type != cd.type) || !cr.valid || cr.bucket != 0 || (cr.has_nulls && !cd.nullable))
    return kNotTaken;
  int64_t total_rows = 0;
  for (int f = 0; f < in->n_frags; ++f) total_rows += in->num_rows[f];
  // (kernel_variant 2 = "the large-input members": how the tests reach this route with small tables)
  if (o.kernel_variant != 2 && total_rows < ((int64_t)4 << 20)) return kNotTaken;
  // the derived plan: the key expression leaves, the expressions behind it move down one column
  mi355q_plan p2 = *plan;
  p2.group_cols[0] = c;
  for (int e = gc - nc; e + 1 < plan->n_exprs; ++e) p2.exprs[e] = plan->exprs[e + 1];
  p2.n_exprs = plan->n_exprs - 1;
  bool key_read = false;
  auto move = [&](int32_t& col) {
    key_read = key_read || col == gc;
    if (col > gc) --col;
  };
  for (int t = 0; t < p2.n_targets; ++t) {
    mi355q_target& tg = p2.targets[t];
    if (tg.agg == MI355Q_PROJECT_KEY) continue;
    if (tg.table == 0 && tg.col >= 0) move(tg.col);
    if (tg.agg == MI355Q_COUNT_IF || tg.agg == MI355Q_SUM_IF) move(tg.cond.col);
  }
  for (int i = 0; i < p2.n_quals; ++i) move(p2.quals[i].col);
  if (key_read) return kNotTaken;  // the cast value itself is aggregated or filtered on
  mi355q_qmd q, q2;
  if (qmd_init(*plan, &q) != MI355Q_OK || qmd_init(p2, &q2) != MI355Q_OK) return kNotTaken;
  if (q.desc_type != MI355Q_GROUP_BY_BASELINE_HASH || q.slot_width != 8 || q.key_width != 8 || q.output_columnar ||
      q2.desc_type != MI355Q_GROUP_BY_PERFECT_HASH || q2.slot_width != 8 || q2.output_columnar)
    return kNotTaken;
  for (int t = 0; t < plan->n_targets; ++t) {
    const int sf = q.target_slot[t], ss = q2.target_slot[t];
    if (plan->targets[t].agg == MI355Q_PROJECT_KEY) {
      if (sf >= 0) return kNotTaken;  // (baseline: projections are read from the key column)
      continue;
    }
    if (sf < 0 || ss < 0) return kNotTaken;
    for (int j = 0; j < (plan->targets[t].agg == MI355Q_AVG ? 2 : 1); ++j)
      if (q.init_vals[sf + j] != q2.init_vals[ss + j]) return kNotTaken;
  }
  mi355q_exec_options o2 = o;
  o2.out_buffer = nullptr;
  if (reserved) {
    route_note("the step grouped by the integer column + k_cast_key_emit");
    return execute_impl(&p2, in, &o2, out, report, nullptr, reserved);
  }
  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  o2.stream = s;
  mi355q_result* r2 = nullptr;
  mi355q_exec_report rep2{};
  if (int32_t e2 = mi355q_execute(&p2, in, &o2, &r2, &rep2)) {
    if (e2 == MI355Q_ERR_UNSUPPORTED || e2 == MI355Q_ERR_OUT_OF_GPU_MEM || e2 < 0) return kNotTaken;
    return e2;
  }
  struct R2Guard {
    mi355q_result* r;
    ~R2Guard() { if (r) mi355q_result_free(r); }
  } r2g{r2};
  if (r2->qmd.desc_type != MI355Q_GROUP_BY_PERFECT_HASH || r2->qmd.slot_width != 8 || r2->qmd.output_columnar ||
      r2->qmd.entry_count != q2.entry_count || r2->qmd.group_min[0] != q2.group_min[0])
    return kNotTaken;
  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  HIP_TRY(launch_init_buffer(res->buf, q.entry_count, make_row_init(q), s));
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, 64));
  HIP_TRY(hipMemsetAsync(err.p, 0, 64, s));
  // (NULL keys of a nullable column sit at max + 1: groupByColumnCodegen translate_null_val, as plan.cpp build_dev_plan)
  HIP_TRY(launch_cast_key_emit(res->dplan, r2->dplan, r2->qmd.idx_target_as_key, to == MI355Q_FLOAT ? 1 : 0, cd.nullable != 0,
                               q2.group_min[0], q2.group_null_key[0], r2->buf, res->buf, (int32_t*)err.p, s));
  if (ev1) HIP_TRY(hipEventRecord(ev1, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(h_err), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_err) return h_err;
  if (report) {
    *report = rep2;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->n_launches = rep2.n_launches + 1;
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}

// Aggregates whose argument is `plain INT column + / - literal` (the reference benchmark's MultiStep shapes:
// MAX(x10 + 1), SUM(x10 + 1) next to MAX(x10)).  The reference compiles the addition into the row function
// (ArithmeticIR.cpp:77-150); materialising it as a projected column costs a pass over the input and a value column of
// the exchange.  Where the column's range says the addition cannot overflow, the step runs on a derived plan that
// aggregates the COLUMN — MIN / MAX / SUM / AVG / COUNT of (x + L) are MIN(x) + L, MAX(x) + L, SUM(x) + L·COUNT(x),
// (SUM(x) + L·COUNT(x)) / COUNT(x), COUNT(x), with NULL rows skipped on both sides and wrapping 64-bit sums — and the
// literal is added while the derived table is copied into the stated layout (k_zip_targets).  The derived plan holds
// every distinct aggregate once, plus COUNT(x) where a sum needs it.  kNotTaken when the shape does not call for it.
int32_t execute_shifted_args(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o,
                             mi355q_result** out, mi355q_exec_report* report, int64_t* reserved) {
  if (plan->n_group_cols < 1 || plan->join_outer_col >= 0 || plan->output_columnar_hint != 0 || o.kernel_variant == 1 ||
      o.force_generic || exprs_read_exprs(*plan))
    return kNotTaken;
  const int nc = plan->n_cols, nx = plan->n_exprs;
  // which expressions are `column +- literal` that cannot overflow
  int src_col[MI355Q_MAX_EXPRS];
  int64_t shift[MI355Q_MAX_EXPRS];
  bool any = false;
  for (int k = 0; k < nx; ++k) {
    src_col[k] = -1;
    const mi355q_expr& ex = plan->exprs[k];
    if (ex.n_nodes != 3 || ex.nodes[0].op != MI355Q_EX_COL || ex.nodes[1].op != MI355Q_EX_LIT || ex.nodes[1].reserved != 0 ||
        (ex.nodes[2].op != MI355Q_EX_ADD && ex.nodes[2].op != MI355Q_EX_SUB))
      continue;
    const int c = ex.nodes[0].arg;
    if (c < 0 || c >= nc) continue;
    const mi355q_col_desc& cd = plan->cols[c];
    const mi355q_range& r = plan->col_ranges[c];
    if ((cd.type != MI355Q_INT32 && cd.type != MI355Q_INT64) || cd.encoding != MI355Q_ENC_NONE ||
        (cd.logical_type != 0 && cd.logical_type != cd.type) || ex.nodes[1].type != cd.type || ex.nodes[2].type != cd.type ||
        !r.valid || r.min > r.max)
      continue;
    const __int128 L = ex.nodes[2].op == MI355Q_EX_ADD ? (__int128)ex.nodes[1].ilit : -(__int128)ex.nodes[1].ilit;
    const __int128 tmax = cd.type == MI355Q_INT32 ? (__int128)INT32_MAX : (__int128)INT64_MAX;
    // (the type's minimum is its NULL; the literal itself must be a value of the type)
    if ((__int128)r.max + L > tmax || (__int128)r.min + L <= -tmax - 1 || L > tmax || L < -tmax) continue;
    src_col[k] = c;
    shift[k] = (int64_t)L;
    any = true;
  }
  if (!any) return kNotTaken;
  // an expression that is also a key, a filter column or a condition stays projected
  for (int g = 0; g < plan->n_group_cols; ++g)
    if (plan->group_cols[g] >= nc && plan->group_cols[g] < nc + nx) src_col[plan->group_cols[g] - nc] = -1;
  for (int i = 0; i < plan->n_quals; ++i)
    if (plan->quals[i].col >= nc && plan->quals[i].col < nc + nx) src_col[plan->quals[i].col - nc] = -1;
  for (int t = 0; t < plan->n_targets; ++t) {
    const mi355q_target& tg = plan->targets[t];
    if ((tg.agg == MI355Q_COUNT_IF || tg.agg == MI355Q_SUM_IF) && tg.cond.col >= nc && tg.cond.col < nc + nx)
      src_col[tg.cond.col - nc] = -1;
    if (tg.agg != MI355Q_PROJECT_KEY && tg.table == 0 && tg.col >= nc && tg.col < nc + nx &&
        (tg.agg == MI355Q_COUNT_IF || tg.agg == MI355Q_SUM_IF))
      src_col[tg.col - nc] = -1;
  }
  any = false;
  for (int k = 0; k < nx; ++k) any = any || src_col[k] >= 0;
  if (!any) return kNotTaken;
  int64_t total_rows = 0;
  for (int f = 0; f < in->n_frags; ++f) total_rows += in->num_rows[f];
  // (kernel_variant 2 = "the large-input members": how the tests reach this route with small tables)
  if (o.kernel_variant != 2 && total_rows < ((int64_t)4 << 20)) return kNotTaken;

  // the derived plan: the shifted expressions leave (the others move down), every distinct aggregate once
  mi355q_plan p2 = *plan;
  int new_col[MI355Q_MAX_EXPRS];  // column index of a kept expression in the derived plan
  p2.n_exprs = 0;
  for (int k = 0; k < nx; ++k) {
    if (src_col[k] >= 0) {
      new_col[k] = -1;
    } else {
      new_col[k] = nc + p2.n_exprs;
      p2.exprs[p2.n_exprs++] = plan->exprs[k];
    }
  }
  auto moved = [&](int col) { return col >= nc && col < nc + nx ? new_col[col - nc] : col; };
  for (int g = 0; g < p2.n_group_cols; ++g) p2.group_cols[g] = moved(plan->group_cols[g]);
  for (int i = 0; i < p2.n_quals; ++i) p2.quals[i].col = moved(plan->quals[i].col);
  p2.n_targets = 0;
  int t2_of[MI355Q_MAX_TARGETS], cnt_t2[MI355Q_MAX_TARGETS];  // final target -> derived target; -> the derived COUNT(x)
  int64_t lit_of[MI355Q_MAX_TARGETS];
  auto same = [](const mi355q_target& a, const mi355q_target& b) {
    return a.agg == b.agg && a.col == b.col && a.table == b.table && a.cond.col == b.cond.col && a.cond.op == b.cond.op &&
           a.cond.ival == b.cond.ival && a.cond.fval == b.cond.fval;
  };
  auto add_target = [&](const mi355q_target& tg) {
    for (int i = 0; i < p2.n_targets; ++i)
      if (same(p2.targets[i], tg)) return i;
    if (p2.n_targets >= MI355Q_MAX_TARGETS) return -1;
    p2.targets[p2.n_targets] = tg;
    return p2.n_targets++;
  };
  for (int t = 0; t < plan->n_targets; ++t) {
    mi355q_target tg = plan->targets[t];
    lit_of[t] = 0;
    cnt_t2[t] = -1;
    if (tg.agg != MI355Q_PROJECT_KEY && tg.table == 0 && tg.col >= nc && tg.col < nc + nx) {
      const int k = tg.col - nc;
      if (src_col[k] >= 0) {
        tg.col = src_col[k];
        lit_of[t] = shift[k];
      } else {
        tg.col = new_col[k];
      }
    }
    if (tg.agg == MI355Q_COUNT_IF || tg.agg == MI355Q_SUM_IF) tg.cond.col = moved(tg.cond.col);
    t2_of[t] = tg.agg == MI355Q_PROJECT_KEY ? (p2.n_targets < MI355Q_MAX_TARGETS ? (p2.targets[p2.n_targets] = tg, p2.n_targets++) : -1)
                                            : add_target(tg);
    if (t2_of[t] < 0) return kNotTaken;
  }
  for (int t = 0; t < plan->n_targets; ++t) {  // the counts the shifted MIN / MAX / SUM targets need
    const mi355q_target& tg = plan->targets[t];
    if (!lit_of[t] || tg.agg == MI355Q_COUNT || tg.agg == MI355Q_AVG) continue;  // (AVG carries its own count)
    mi355q_target cnt{};
    cnt.agg = MI355Q_COUNT;
    cnt.col = p2.targets[t2_of[t]].col;
    cnt.table = 0;
    cnt_t2[t] = add_target(cnt);
    if (cnt_t2[t] < 0) return kNotTaken;
  }
  mi355q_qmd q, q2;
  if (qmd_init(*plan, &q) != MI355Q_OK || qmd_init(p2, &q2) != MI355Q_OK) return kNotTaken;
  if (q.slot_width != 8 || q2.slot_width != 8 || q.output_columnar || q2.output_columnar || q.desc_type != q2.desc_type ||
      q.entry_count != q2.entry_count || q.desc_type == MI355Q_NON_GROUPED_AGGREGATE ||
      (q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH && q.key_width != q2.key_width))
    return kNotTaken;
  int32_t src[MI355Q_MAX_SLOTS], dst[MI355Q_MAX_SLOTS], kind[MI355Q_MAX_SLOTS], cnt_src[MI355Q_MAX_SLOTS];
  int64_t lit[MI355Q_MAX_SLOTS];
  int n = 0;
  for (int t = 0; t < plan->n_targets; ++t) {
    const int sf = q.target_slot[t], ss = q2.target_slot[t2_of[t]];
    if ((sf < 0) != (ss < 0)) return kNotTaken;  // (a projection read from the key columns on both sides)
    if (sf < 0) continue;
    const int agg = plan->targets[t].agg;
    for (int j = 0; j < (agg == MI355Q_AVG ? 2 : 1); ++j) {
      if (q.init_vals[sf + j] != q2.init_vals[ss + j] || n >= MI355Q_MAX_SLOTS) return kNotTaken;
      src[n] = ss + j;
      dst[n] = sf + j;
      kind[n] = 0;
      cnt_src[n] = 0;
      lit[n] = 0;
      if (lit_of[t] && j == 0 && agg != MI355Q_COUNT) {
        kind[n] = (agg == MI355Q_SUM || agg == MI355Q_AVG) ? 2 : 1;
        cnt_src[n] = agg == MI355Q_AVG ? ss + 1 : q2.target_slot[cnt_t2[t]];
        if (cnt_src[n] < 0) return kNotTaken;
        lit[n] = lit_of[t];
      }
      ++n;
    }
  }
  mi355q_exec_options o2 = o;
  o2.out_buffer = nullptr;
  if (reserved) {
    route_note("aggregates of column + literal from the column's aggregates + k_zip_targets");
    return execute_impl(&p2, in, &o2, out, report, nullptr, reserved);
  }
  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  o2.stream = s;
  mi355q_result* r2 = nullptr;
  mi355q_exec_report rep2{};
  if (int32_t e2 = mi355q_execute(&p2, in, &o2, &r2, &rep2)) {
    if (e2 == MI355Q_ERR_UNSUPPORTED || e2 == MI355Q_ERR_OUT_OF_GPU_MEM) return kNotTaken;
    return e2;  // (out of slots included: the derived table has the stated table's entry count)
  }
  struct R2Guard {
    mi355q_result* r;
    ~R2Guard() { if (r) mi355q_result_free(r); }
  } r2g{r2};
  if (r2->qmd.desc_type != q.desc_type || r2->qmd.entry_count != q.entry_count || r2->qmd.slot_width != 8 ||
      r2->qmd.output_columnar)
    return kNotTaken;
  for (int t = 0; t < plan->n_targets; ++t)
    if (r2->qmd.target_slot[t2_of[t]] != q2.target_slot[t2_of[t]]) return kNotTaken;
  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  HIP_TRY(launch_init_buffer(res->buf, q.entry_count, make_row_init(q), s));
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, 64));
  HIP_TRY(hipMemsetAsync(err.p, 0, 64, s));
  HIP_TRY(launch_zip_targets(res->dplan, r2->dplan, r2->qmd.idx_target_as_key, r2->buf, res->buf, src, dst, n, (int32_t*)err.p, s,
                             kind, cnt_src, lit, /*into_empty_table=*/true));
  if (ev1) HIP_TRY(hipEventRecord(ev1, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(h_err), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (h_err) return h_err;
  if (report) {
    *report = rep2;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->n_launches = rep2.n_launches + 1;
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}

// Plans with projected expressions (mi355q_expr): scan / filter / PROJECT.  The expressions of a pass of
// fragments are evaluated into dense temporary columns (k_project), the step runs on the lowered plan — where
// those columns are ordinary inputs, so every kernel family applies — and the passes' results are folded with
// the reduce rule (ResultSetStorage::reduce, as for the reference's per-fragment kernels).  One pass when the
// temporary columns fit a third of the free memory (<= 16 GB).
int32_t execute_projected(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o,
                          mi355q_result** out, mi355q_exec_report* report) {
  mi355q_plan lp;
  DevExprSet xs;
  if (int32_t e = lower_exprs(*plan, &lp, &xs, true)) return e;
  mi355q_qmd q;
  if (int32_t e = qmd_init(*plan, &q)) return e;
  DevPlan d;
  if (int32_t e = build_dev_plan(lp, q, &d)) return e;
  if (int32_t e = attach_join(lp, in, &d)) return e;
  const int nf = in->n_frags, nc = plan->n_cols, nx = plan->n_exprs, nc2 = nc + nx;
  if (nf == 0) return mi355q_execute(&lp, in, &o, out, report);
  const uint32_t qual_expr_mask = expr_qual_mask(*plan);
  int64_t total_rows = 0, max_frag_rows = 0;
  for (int f = 0; f < nf; ++f) {
    if (in->num_rows[f] < 0) return MI355Q_ERR_INVALID_PLAN;
    total_rows += in->num_rows[f];
    max_frag_rows = std::max(max_frag_rows, in->num_rows[f]);
  }
  int64_t row_bytes = 0;
  for (int k = 0; k < nx; ++k) row_bytes += plain_width(xs.e[k].store_type);

  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  const int n_cus = cu_count_of(in->device_id);
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const int64_t budget = std::min<int64_t>((int64_t)16 << 30, ((int64_t)free_b + ctx.proj_bytes) / 3);
  // every (fragment, expression) chunk starts on a 16-byte boundary: the fast families want aligned columns
  const int64_t pad = 16 * (int64_t)nx;
  int64_t pass_rows = std::max<int64_t>(budget / std::max<int64_t>(row_bytes, 1), max_frag_rows);
  if (o.pass_rows > 0) pass_rows = std::max<int64_t>(o.pass_rows, max_frag_rows);  // tests: several passes
  if (pass_rows > total_rows) pass_rows = total_rows;
  const int64_t tab_bytes = ((int64_t)sizeof(void*) * nf * nc2 + 255) & ~255ll;
  const int64_t rows_bytes = ((int64_t)sizeof(int64_t) * nf + 255) & ~255ll;
  // (worst case: every fragment of a pass pads every expression chunk once)
  const int64_t col_region = ((pass_rows * row_bytes + pad * nf) + 255) & ~255ll;
  const int64_t xs_bytes = ((int64_t)sizeof(DevExprSet) + 255) & ~255ll;  // the lowered programs, read by k_project from device memory
  const int64_t need = col_region + tab_bytes + rows_bytes + 256 + xs_bytes;
  if (ctx.proj_bytes < need) {
    if (ctx.proj) (void)hipFree(ctx.proj);
    ctx.proj = nullptr;
    ctx.proj_bytes = 0;
    hipError_t he = hipMalloc(&ctx.proj, (size_t)need);
    if (he != hipSuccess) {
      last_hip_error = he;
      (void)hipGetLastError();
      return MI355Q_ERR_OUT_OF_GPU_MEM;
    }
    ctx.proj_bytes = need;
  }
  char* base = (char*)ctx.proj;
  const int8_t** d_tab = (const int8_t**)(base + col_region);
  int64_t* d_rows = (int64_t*)(base + col_region + tab_bytes);
  int32_t* d_err = (int32_t*)(base + col_region + tab_bytes + rows_bytes);
  DevExprSet* d_xs = (DevExprSet*)(base + col_region + tab_bytes + rows_bytes + 256);
  HIP_TRY(hipMemsetAsync(d_err, 0, 64, s));
  HIP_TRY(hipMemcpyAsync(d_rows, in->num_rows, sizeof(int64_t) * (size_t)nf, hipMemcpyHostToDevice, s));

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};

  std::vector<const void*> cols2((size_t)nf * nc2);
  mi355q_result* res = nullptr;
  struct ResGuard {
    mi355q_result*& r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  mi355q_exec_report acc{};
  int pass = 0, f = 0;
  while (f < nf) {
    int f1 = f;
    int64_t rows = 0, off = 0;
    while (f1 < nf && (f1 == f || rows + in->num_rows[f1] <= pass_rows)) {
      for (int c = 0; c < nc; ++c) cols2[(size_t)(f1 - f) * nc2 + c] = in->col_buffers[(size_t)f1 * nc + c];
      for (int k = 0; k < nx; ++k) {
        cols2[(size_t)(f1 - f) * nc2 + nc + k] = base + off;
        off += (in->num_rows[f1] * plain_width(xs.e[k].store_type) + 15) & ~15ll;
      }
      rows += in->num_rows[f1];
      ++f1;
    }
    if (off > col_region) return MI355Q_ERR_OUT_OF_GPU_MEM;  // (cannot happen: the region is sized for it)
    const int pnf = f1 - f;
    HIP_TRY(hipMemcpyAsync(d_tab, cols2.data(), sizeof(void*) * (size_t)pnf * nc2, hipMemcpyHostToDevice, s));
    // one-operation expressions over aligned plain columns: the vectorised members (k_project_simple); an overflow there
    // only raises word 2 and the interpreter (which knows whether the offending row counts) runs after all
    bool simple = project_simple_shapes(xs) && !o.force_generic;
    for (size_t i = 0; simple && i < (size_t)pnf * nc2; ++i) simple = ((uintptr_t)cols2[i] & 15) == 0;
    HIP_TRY(launch_project(xs, d_xs, d, qual_expr_mask, d_tab, d_rows + f, pnf, max_frag_rows, d_err, n_cus, s, simple));
    int32_t h_err3[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h_err3, d_err, sizeof(h_err3), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));  // (cols2 is re-used by the next pass; the step below synchronises anyway)
    if (simple && h_err3[2]) {
      HIP_TRY(hipMemsetAsync(d_err, 0, 64, s));
      HIP_TRY(launch_project(xs, d_xs, d, qual_expr_mask, d_tab, d_rows + f, pnf, max_frag_rows, d_err, n_cus, s, false));
      HIP_TRY(hipMemcpyAsync(h_err3, d_err, sizeof(h_err3), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    const int32_t h_err = h_err3[0];
    if (h_err) return h_err;
    mi355q_inputs in2 = *in;
    in2.n_frags = pnf;
    in2.col_buffers = cols2.data();
    in2.num_rows = in->num_rows + f;
    mi355q_exec_options o2 = o;
    o2.stream = s;
    o2.out_buffer = pass == 0 ? o.out_buffer : nullptr;
    mi355q_result* r2 = nullptr;
    mi355q_exec_report rep2{};
    if (int32_t e2 = mi355q_execute(&lp, &in2, &o2, &r2, &rep2)) return e2;
    if (pass == 0) {
      res = r2;
      std::snprintf(acc.kernel_name, sizeof(acc.kernel_name), "%s", rep2.kernel_name);
      acc.variant = rep2.variant;
    } else {
      const int32_t er = mi355q_result_reduce(res, r2, s);
      mi355q_result_free(r2);
      if (er) return er;
    }
    acc.kernel_ms += rep2.kernel_ms;
    acc.n_launches += rep2.n_launches;
    acc.spilled_rows += rep2.spilled_rows;
    f = f1;
    ++pass;
  }
  if (ev1) {
    HIP_TRY(hipEventRecord(ev1, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  if (report) {
    *report = acc;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  *out = res;
  res = nullptr;
  return MI355Q_OK;
}


// A filter compiled at plan time whose atoms include PROGRAMS (boolfilter.h, regprog.h: `b <> 0 AND a / b > 3`, `x + y > 100`,
// `a < b`, DOUBLE leaves): the pre-pass k_filter_mask streams the filter's columns once and leaves one byte per row, the step
// proper is `rest` (the stated plan without quals and expressions) with ONE more column — the mask, INT8 NOT NULL in [0, 1] —
// and the qual `mask = 1`, which every typed family loads as one 4-byte word per quad.  Errors (7 / 1) surface from the
// pre-pass: every row evaluates the filter's expressions, as the row function does ahead of its quals.
// kNotTaken: unaligned filter columns, no room for one more column (the caller falls through to the interpreter pass).
int32_t execute_masked(const mi355q_plan* plan, const mi355q_plan& rest, const BoolFilterHost& bfh, const mi355q_inputs* in,
                       const mi355q_exec_options& o, mi355q_result** out, mi355q_exec_report* report, int64_t* reserved) {
  const int nc = rest.n_cols, nc2 = nc + 1, nf = in->n_frags;
  if (nc2 > MI355Q_MAX_COLS || rest.n_quals != 0 || rest.n_exprs != 0) return kNotTaken;
  mi355q_plan mp = rest;
  mp.n_cols = nc2;
  std::memset(&mp.cols[nc], 0, sizeof(mp.cols[nc]));
  mp.cols[nc].type = MI355Q_INT8;
  std::memset(&mp.col_ranges[nc], 0, sizeof(mp.col_ranges[nc]));
  mp.col_ranges[nc].valid = 1;
  mp.col_ranges[nc].min = 0;
  mp.col_ranges[nc].max = 1;
  mp.n_quals = 1;
  std::memset(&mp.quals[0], 0, sizeof(mp.quals[0]));
  mp.quals[0].col = nc;
  mp.quals[0].op = MI355Q_EQ;
  mp.quals[0].ival = 1;
  int64_t total_rows = 0, max_frag_rows = 0;
  for (int f = 0; f < nf; ++f) {
    if (in->num_rows[f] < 0) return MI355Q_ERR_INVALID_PLAN;
    total_rows += in->num_rows[f];
    max_frag_rows = std::max(max_frag_rows, in->num_rows[f]);
  }
  if (reserved) {  // mi355q_reserve_workspace / mi355q_explain: nothing is launched
    route_note("k_filter_mask (program atoms + truth table -> 1 B/row)");
    return execute_impl(&mp, in, &o, out, report, nullptr, reserved);
  }
  if (nf == 0) return mi355q_execute(&mp, in, &o, out, report);
  {
    FragView hv{nullptr, nullptr, in->col_buffers, in->num_rows, nf, nc, total_rows, max_frag_rows};
    if (!filter_mask_eligible(bfh.bf, hv)) return kNotTaken;
  }
  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  const int n_cus = o.tune_cus > 0 ? std::min(o.tune_cus, cu_count_of(in->device_id)) : cu_count_of(in->device_id);
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  hipStream_t s = (hipStream_t)o.stream;
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  // one pass of fragments = as many as the mask region holds (1 B/row: 16 GB of it cover 16 G rows)
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const int64_t budget = std::min<int64_t>((int64_t)16 << 30, ((int64_t)free_b + ctx.maskws_bytes) / 3);
  int64_t pass_rows = std::max<int64_t>(budget, max_frag_rows);
  if (o.pass_rows > 0) pass_rows = std::max<int64_t>(o.pass_rows, max_frag_rows);  // tests: several passes
  if (pass_rows > total_rows) pass_rows = total_rows;
  const int64_t tab_bytes = ((int64_t)sizeof(void*) * nf * nc + 255) & ~255ll;    // the pre-pass's view: the stated columns
  const int64_t mtab_bytes = ((int64_t)sizeof(void*) * nf + 255) & ~255ll;        // per fragment: its mask chunk
  const int64_t rows_bytes = ((int64_t)sizeof(int64_t) * nf + 255) & ~255ll;
  const int64_t col_region = (pass_rows + 32 * (int64_t)nf + 255) & ~255ll;
  const int64_t need = col_region + tab_bytes + mtab_bytes + rows_bytes + 256;
  if (ctx.maskws_bytes < need) {
    if (ctx.maskws) (void)hipFree(ctx.maskws);
    ctx.maskws = nullptr;
    ctx.maskws_bytes = 0;
    hipError_t he = hipMalloc(&ctx.maskws, (size_t)need);
    if (he != hipSuccess) {
      last_hip_error = he;
      (void)hipGetLastError();
      return MI355Q_ERR_OUT_OF_GPU_MEM;
    }
    ctx.maskws_bytes = need;
  }
  if (!ctx.bf_table) HIP_TRY(hipMalloc(&ctx.bf_table, sizeof(BoolFilter)));
  HIP_TRY(hipMemcpy(ctx.bf_table, &bfh.bf, sizeof(BoolFilter), hipMemcpyHostToDevice));  // (synchronous: out of the caller's frame)
  char* base = (char*)ctx.maskws;
  const int8_t** d_tab = (const int8_t**)(base + col_region);
  int8_t** d_mtab = (int8_t**)(base + col_region + tab_bytes);
  int64_t* d_rows = (int64_t*)(base + col_region + tab_bytes + mtab_bytes);
  int32_t* d_err = (int32_t*)(base + col_region + tab_bytes + mtab_bytes + rows_bytes);
  HIP_TRY(hipMemsetAsync(d_err, 0, 64, s));
  HIP_TRY(hipMemcpyAsync(d_rows, in->num_rows, sizeof(int64_t) * (size_t)nf, hipMemcpyHostToDevice, s));

  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (report) {
    HIP_TRY(hipEventCreate(&ev0));
    HIP_TRY(hipEventCreate(&ev1));
    HIP_TRY(hipEventRecord(ev0, s));
  }
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } evg{ev0, ev1};
  std::vector<const void*> cols2((size_t)nf * nc2);
  std::vector<void*> masks((size_t)nf);
  mi355q_result* res = nullptr;
  struct ResGuard {
    mi355q_result*& r;
    ~ResGuard() { if (r) mi355q_result_free(r); }
  } rg{res};
  mi355q_exec_report acc{};
  int pass = 0, f = 0;
  while (f < nf) {
    int f1 = f;
    int64_t rows = 0, off = 0;
    while (f1 < nf && (f1 == f || rows + in->num_rows[f1] <= pass_rows)) {
      for (int c = 0; c < nc; ++c) cols2[(size_t)(f1 - f) * nc2 + c] = in->col_buffers[(size_t)f1 * nc + c];
      cols2[(size_t)(f1 - f) * nc2 + nc] = base + off;
      masks[(size_t)(f1 - f)] = base + off;
      off += filter_mask_chunk_bytes(in->num_rows[f1]);
      rows += in->num_rows[f1];
      ++f1;
    }
    if (off > col_region) return MI355Q_ERR_OUT_OF_GPU_MEM;  // (cannot happen: the region is sized for it)
    const int pnf = f1 - f;
    HIP_TRY(hipMemcpyAsync(d_tab, in->col_buffers + (size_t)f * nc, sizeof(void*) * (size_t)pnf * nc, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_mtab, masks.data(), sizeof(void*) * (size_t)pnf, hipMemcpyHostToDevice, s));
    int64_t prows = 0, pmax = 0;
    for (int i = f; i < f1; ++i) {
      prows += in->num_rows[i];
      pmax = std::max(pmax, in->num_rows[i]);
    }
    FragView fv{d_tab, d_rows + f, in->col_buffers + (size_t)f * nc, in->num_rows + f, pnf, nc, prows, pmax};
    HIP_TRY(launch_filter_mask(bfh.bf, (const BoolFilter*)ctx.bf_table, fv, d_mtab, d_err, n_cus, s));
    int32_t h_err = 0;
    if (bfh.bf.any_raise) {  // (a filter that cannot raise leaves nothing to look at: the step proper is enqueued right behind)
      HIP_TRY(hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (h_err) return h_err;
    } else {
      HIP_TRY(hipStreamSynchronize(s));  // (masks / cols2 are re-used by the next pass; the step below synchronises anyway)
    }
    mi355q_inputs in2 = *in;
    in2.n_frags = pnf;
    in2.col_buffers = cols2.data();
    in2.num_rows = in->num_rows + f;
    mi355q_exec_options o2 = o;
    o2.stream = s;
    o2.out_buffer = pass == 0 ? o.out_buffer : nullptr;
    mi355q_result* r2 = nullptr;
    mi355q_exec_report rep2{};
    if (int32_t e2 = mi355q_execute(&mp, &in2, &o2, &r2, &rep2)) return e2;
    if (pass == 0) {
      res = r2;
      std::snprintf(acc.kernel_name, sizeof(acc.kernel_name), "%s", rep2.kernel_name);
      acc.variant = rep2.variant;
    } else {
      const int32_t er = mi355q_result_reduce(res, r2, s);
      mi355q_result_free(r2);
      if (er) return er;
    }
    acc.kernel_ms += rep2.kernel_ms;
    acc.n_launches += rep2.n_launches + 1;
    acc.spilled_rows += rep2.spilled_rows;
    f = f1;
    ++pass;
  }
  if (ev1) {
    HIP_TRY(hipEventRecord(ev1, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  if (report) {
    *report = acc;
    (void)hipEventElapsedTime(&report->total_ms, ev0, ev1);
    report->rows_scanned = total_rows;
    report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
  }
  *out = res;
  res = nullptr;
  return MI355Q_OK;
}

}  // namespace

namespace {
// An INNER join on ONE key against a dense OneToOne perfect table (mi355q_join_table::dense) whose inner side no target reads:
// a row has a match exactly when its key lies in [min, max] (hash_join_idx, GroupByRuntime.cpp:287-297: in range ->
// the slot, and no slot of a dense table is -1; a NULL key matches nothing, hash_join_idx_nullable :311-318), so the step is
// the same step WITHOUT the join and with the two range quals on the key column — a plain scan for the non-grouped
// shapes (BASELINE cfg4, Query A on the dense dimension).  kNotTaken for everything else.
int32_t execute_dense_join_as_filter(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o,
                                     mi355q_result** out, mi355q_exec_report* report, int64_t* reserved) {
  const mi355q_join_table* jt = plan->join_table;
  if (plan->join_outer_col < 0 || !jt || !jt->dense || plan->join_kind != MI355Q_JOIN_INNER || plan->n_join_cols > 1 ||
      plan->n_exprs != 0 || plan->n_quals + 2 > MI355Q_MAX_QUALS || o.force_generic || o.kernel_variant != 0)
    return kNotTaken;
  for (int t = 0; t < plan->n_targets && t < MI355Q_MAX_TARGETS; ++t)
    if (plan->targets[t].table != 0 && plan->targets[t].agg != MI355Q_PROJECT_KEY) return kNotTaken;
  const int kc = plan->join_outer_col;
  if (kc >= plan->n_cols || col_type_code(plan->cols[kc]) < 0) return kNotTaken;
  const int lt = tc_logical(col_type_code(plan->cols[kc]));
  if (lt < MI355Q_INT8 || lt > MI355Q_INT64) return kNotTaken;
  mi355q_plan p2 = *plan;
  p2.join_outer_col = -1;
  p2.join_table = nullptr;
  p2.n_join_cols = 0;
  p2.n_inner_cols = 0;
  p2.quals[p2.n_quals++] = mi355q_qual{kc, MI355Q_GE, jt->min_key, 0.0};
  p2.quals[p2.n_quals++] = mi355q_qual{kc, MI355Q_LE, jt->max_key, 0.0};
  mi355q_qmd qa, qb;   // the layout does not look at the join or at the quals: the derived step's result IS the step's
  if (qmd_init(*plan, &qa) || qmd_init(p2, &qb) || std::memcmp(&qa, &qb, sizeof(qa)) != 0) return kNotTaken;
  route_note("join on a dense one-to-one table = range filter on the key");
  mi355q_inputs in2 = *in;
  in2.inner_col_buffers = nullptr;
  in2.inner_num_rows = 0;
  return execute_impl(&p2, &in2, &o, out, report, nullptr, reserved);
}
}  // namespace

// ------------------------------------------------------------------------------- execute
int32_t mi355q_execute(const mi355q_plan* plan, const mi355q_inputs* in,
                       const mi355q_exec_options* opts, mi355q_result** out,
                       mi355q_exec_report* report) {
  return execute_impl(plan, in, opts, out, report, nullptr);
}

int32_t mi355q_reserve_workspace(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options* opts,
                                 int64_t* reserved_bytes) {
  if (!plan || !in) return MI355Q_ERR_INVALID_PLAN;
  int64_t bytes = 0;
  // the planner looks at the pointer table for alignment only: a table of NULLs stands in for it
  std::vector<const void*> none((size_t)std::max(1, in->n_frags * std::max(plan->n_cols + plan->n_exprs + 1, 1)), nullptr);  // (+ 1: the row mask of a compiled filter, execute_masked)
  mi355q_inputs in2 = *in;
  in2.col_buffers = none.data();
  mi355q_result* r = nullptr;
  const int32_t e = execute_impl(plan, &in2, opts, &r, nullptr, nullptr, &bytes);
  if (r) mi355q_result_free(r);
  if (reserved_bytes) *reserved_bytes = bytes;
  return e;
}

// EXPLAIN for one step: the route mi355q_execute would take for this plan over inputs of this shape (fragment row
// counts; the pointer table is not looked at: 16-byte aligned chunks are assumed), as a " > "-separated chain of
// the derived-plan stages and the kernel family that runs the step, e.g.
//   "k_project > k_pack_keys (entry index, baseline temp) > k_part_scatter + k_part_aggregate > k_unpack_emit".
// Nothing is launched and no workspace is allocated; *scratch_bytes = the partition scratch the step would ask for.
int32_t mi355q_explain(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options* opts, char* route,
                       int64_t route_len, int64_t* scratch_bytes) {
  if (!plan || !in || (route_len > 0 && !route)) return MI355Q_ERR_INVALID_PLAN;
  int64_t bytes = 0;
  std::string text;
  int32_t e;
  try {
    std::vector<const void*> none((size_t)std::max(1, in->n_frags * std::max(plan->n_cols + plan->n_exprs + 1, 1)), nullptr);  // (+ 1: the row mask of a compiled filter, execute_masked)
    mi355q_inputs in2 = *in;
    in2.col_buffers = none.data();  // (16-byte aligned chunks assumed, as mi355q_reserve_workspace does)
    mi355q_result* r = nullptr;
    t_route = &text;
    t_plan_only = true;
    e = execute_impl(plan, &in2, opts, &r, nullptr, nullptr, &bytes);
    t_route = nullptr;
    t_plan_only = false;
    if (r) mi355q_result_free(r);
  } catch (...) {
    t_route = nullptr;
    t_plan_only = false;
    return MI355Q_ERR_OUT_OF_CPU_MEM;
  }
  if (route && route_len > 0) std::snprintf(route, (size_t)route_len, "%s", text.c_str());
  if (scratch_bytes) *scratch_bytes = bytes;
  return e;
}

// Stream-ordered form of mi355q_execute: every kernel of the step is enqueued on the stream and the call
// returns without waiting for the device.  *out is valid at once for STREAM-ORDERED use on the same stream
// (mi355q_shard_pads, a collective enqueued behind it, ...); whether the step succeeded — error code, the
// retry with the direct member after a spill overflow, the report — is only known after mi355q_wait.  One step
// per device may be in flight: the next call on the device finishes it first.  Routes that need the host in
// the middle of the step (projected expressions, packed multi-column keys, columnar / 4-byte-slot results,
// the first build of a join payload) run to completion inside this call; mi355q_wait then returns at once.
int32_t mi355q_execute_async(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options* opts,
                             mi355q_result** out, mi355q_pending** pending) {
  if (!pending) return MI355Q_ERR_INVALID_PLAN;
  *pending = nullptr;
  mi355q_pending* p = nullptr;
  mi355q_exec_report rep{};
  const int32_t e = execute_impl(plan, in, opts, out, &rep, &p);
  if (!p) {  // completed (or failed) synchronously
    if (e) return e;  // a failed call leaves no handle behind (*pending stays NULL)
    p = new (std::nothrow) mi355q_pending();
    if (!p) return e ? e : MI355Q_ERR_OUT_OF_CPU_MEM;
    p->device_id = in ? in->device_id : 0;
    p->code = e;
    p->rep = rep;
  }
  *pending = p;
  return p->tail ? MI355Q_OK : e;
}

int32_t mi355q_wait(mi355q_pending* p, mi355q_exec_report* report) {
  if (!p) return MI355Q_ERR_INVALID_PLAN;
  {
    DeviceCtx& ctx = ctx_of(p->device_id);
    std::lock_guard<std::recursive_mutex> lk(ctx.mu);
    if (ctx.inflight == p) {
      DeviceGuard g(p->device_id);
      drain_inflight(ctx);
    }
  }
  if (report) *report = p->rep;
  const int32_t code = p->code;
  delete p->tail;
  delete p;
  return code;
}

namespace {
int32_t execute_impl(const mi355q_plan* plan, const mi355q_inputs* in,
                     const mi355q_exec_options* opts, mi355q_result** out,
                     mi355q_exec_report* report, mi355q_pending** pend, int64_t* reserved) {
  if (!plan || !in || !out) return MI355Q_ERR_INVALID_PLAN;
  if (in->n_frags < 0 || (in->n_frags > 0 && (!in->col_buffers || !in->num_rows)))
    return MI355Q_ERR_INVALID_PLAN;
  *out = nullptr;
  {  // a step left in flight by mi355q_execute_async owns the device workspace: finish it first
    DeviceCtx& c0 = ctx_of(in->device_id);
    std::lock_guard<std::recursive_mutex> lk0(c0.mu);
    if (c0.inflight) {
      DeviceGuard g0(in->device_id);
      drain_inflight(c0);
    }
  }
  mi355q_exec_options o{};
  if (opts) o = *opts;
  for (int i = 0; i < plan->n_quals && i < MI355Q_MAX_QUALS; ++i)
    if (MI355Q_QUAL_OR_GROUP(plan->quals[i].op) != 0) o.force_generic = 1;  // a disjunction among the quals: the row kernel
  {
    TuneKnobs k;
    k.blocks_per_cu = o.tune_blocks_per_cu;
    k.probe_keyed_passes = o.probe_keyed_passes;
    k.pass_rows = o.pass_rows;
    k.flags = o.flags;
    k.overlap_cus = o.tune_overlap_cus;
    set_tune_knobs(k);
  }
  bool proj_step = plan->n_targets > 0;
  for (int i = 0; i < plan->n_targets && i < MI355Q_MAX_TARGETS; ++i) proj_step = proj_step && plan->targets[i].agg == MI355Q_PROJECT;
  // (a Projection evaluates its expressions — filters included — in the compaction kernel's registers: below)
  if (plan->n_exprs != 0 && !pend && !step_bool_filter() && !o.force_generic && !proj_step && !(o.flags & MI355Q_OPT_NO_COMPILED_FILTER)) {
    // A filter of comparisons with literals under AND / OR / NOT is compiled into atoms + a truth table and evaluated by
    // the consuming kernel on the values it holds in registers (boolfilter.h): no temporary column, no second pass.
    BoolFilterHost bfh;
    mi355q_plan rest;
    if (compile_bool_filter(*plan, &bfh, &rest)) {
      const size_t mark = t_route ? t_route->size() : 0;
      int32_t e = kNotTaken;
      const bool fused_progs = bfh.bf.n_progs == 0 || (bfh.bf.all_lean && bfh.bf.all_i32 && bfh.bf.n_progs <= kLdsFusedProgs &&
                                                       !(o.flags & MI355Q_OPT_FILTER_PREPASS));
      if (fused_progs) {
        DeviceGuard gb(in->device_id);
        DeviceCtx& cb = ctx_of(in->device_id);
        std::lock_guard<std::recursive_mutex> lb(cb.mu);
        bool have = true;
        if (!t_plan_only) {
          if (!cb.bf_table) have = gb.ok && hipMalloc(&cb.bf_table, sizeof(BoolFilter)) == hipSuccess;
          // (a synchronous copy out of this frame: atoms + truth table, 1.2 KB)
          have = have && hipMemcpy(cb.bf_table, &bfh.bf, sizeof(BoolFilter), hipMemcpyHostToDevice) == hipSuccess;
        }
        if (have) {
          struct BfScope {
            BfScope(const BoolFilter* b, const BoolFilter* d) { set_step_bool_filter(b, d); }
            ~BfScope() { set_step_bool_filter(nullptr, nullptr); }
          } scope(&bfh.bf, (const BoolFilter*)cb.bf_table);
          route_note("filter compiled (atoms + truth table)");
          e = execute_impl(&rest, in, &o, out, report, nullptr, reserved);
        } else {
          (void)hipGetLastError();
        }
      }
      if (e == kNotTaken && bfh.bf.n_progs != 0) {  // program atoms no family evaluates itself: the row-mask pre-pass, then `mask = 1`
        if (t_route) t_route->resize(mark);
        *out = nullptr;
        route_note("filter compiled (atoms + programs + truth table)");
        e = execute_masked(plan, rest, bfh, in, o, out, report, reserved);
      }
      if (e != kNotTaken) {
        if (e == MI355Q_OK && report && !reserved) report->algorithmic_bytes = algorithmic_bytes(*plan, *in);
        return e;
      }
      if (t_route) t_route->resize(mark);
      *out = nullptr;
    }
  }
  if (plan->n_exprs != 0 && proj_step && !pend && !step_bool_filter() && !o.force_generic && !(o.flags & MI355Q_OPT_NO_COMPILED_FILTER)) {
    // a Projection whose expressions all belong to the FILTER (`SELECT a, b FROM t WHERE x + y > 100`): the filter is compiled,
    // the row-mask pre-pass evaluates it, the Projection runs on `mask = 1` — in its fast member where the targets allow —
    // instead of evaluating the quals' expressions row by row in the general member.  With a LIMIT only where the filter
    // cannot raise: the pre-pass looks at every row, the reference's loop stops at the limit.
    BoolFilterHost bfh;
    mi355q_plan rest;
    if (compile_bool_filter(*plan, &bfh, &rest) && (!bfh.bf.any_raise || plan->scan_limit == 0)) {
      const size_t mark = t_route ? t_route->size() : 0;
      route_note("filter compiled (atoms + programs + truth table)");
      const int32_t e = execute_masked(plan, rest, bfh, in, o, out, report, reserved);
      if (e != kNotTaken) return e;
      if (t_route) t_route->resize(mark);
      *out = nullptr;
    }
  }
  if (plan->n_exprs != 0) {
    {  // a Projection evaluates its expressions in the compaction kernel's registers: no k_project pass
      if (proj_step) return execute_projection(plan, in, o, out, report, reserved);
    }
    if (!pend) {
      const size_t mark = t_route ? t_route->size() : 0;
      int32_t e = execute_shifted_args(plan, in, o, out, report, reserved);
      if (e != kNotTaken) return e;
      if (t_route) t_route->resize(mark);
      e = execute_cast_key(plan, in, o, out, report, reserved);
      if (e != kNotTaken) return e;
      if (t_route) t_route->resize(mark);
    }
    if (reserved) {  // the step proper runs on the lowered plan: reserve for that
      route_note("k_project");
      mi355q_plan lp;
      if (int32_t e = lower_exprs(*plan, &lp, nullptr, true)) return e;
      return execute_impl(&lp, in, &o, out, report, nullptr, reserved);
    }
    return execute_projected(plan, in, o, out, report);
  }

  Trace tr((o.flags & MI355Q_OPT_TRACE) != 0);
  mi355q_qmd q;
  if (int32_t e = qmd_init(*plan, &q)) return e;
  if (q.desc_type == MI355Q_PROJECTION) return execute_projection(plan, in, o, out, report, reserved);
  DevPlan d;
  if (int32_t e = build_dev_plan(*plan, q, &d)) return e;
  if (int32_t e = attach_join(*plan, in, &d)) return e;
  // a compiled filter travels beside the plan (its quals are gone): only the families that take one may run the step —
  // no derived plans, no row kernel; kNotTaken sends the caller back to the projection pass
  d.bf_active = step_bool_filter() != nullptr;
  const bool bf_step = d.bf_active != 0;

  DeviceGuard g(in->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  const int n_cus = o.tune_cus > 0 ? std::min(o.tune_cus, cu_count_of(in->device_id)) : cu_count_of(in->device_id);

  if (q.output_columnar) {
    // Columnar output: the step runs on the row-wise form of the same decisions (same entry
    // count, 8-byte key components, same hash, same slots) into a library-owned table, and the
    // finished table is moved column by column into the caller-visible buffer.
    mi355q_plan pr = *plan;
    pr.output_columnar_hint = MI355Q_OUTPUT_ROWWISE_COLUMNAR_DECISIONS;
    mi355q_exec_options orw = o;
    orw.out_buffer = nullptr;
    if (reserved) {
      route_note("row-wise twin + k_rows_to_columns");
      return execute_impl(&pr, in, &orw, out, report, nullptr, reserved);
    }
    RowTwin t;
    if (int32_t e = mi355q_execute(&pr, in, &orw, &t.tw, report)) return e;
    const mi355q_qmd& qr = t.tw->qmd;
    if (qr.entry_count != q.entry_count || qr.row_size != q.row_size || qr.key_bytes != q.key_bytes ||
        qr.slot_count != q.slot_count || qr.slot_width != q.slot_width || qr.output_columnar)
      return MI355Q_ERR_UNSUPPORTED;
    mi355q_result* rc = nullptr;
    if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &rc)) return e;
    if (int32_t e = store_row_twin(t, rc, (hipStream_t)o.stream)) {
      mi355q_result_free(rc);
      return e;
    }
    *out = rc;
    return MI355Q_OK;
  }

  if (q.slot_width == 4) {
    // Compact layout (4-byte slots): the step runs on the 8-byte layout of the same plan — same
    // entry count, same key bytes, same hash — and the finished table is narrowed row by row.
    int64_t rows = 0;
    for (int f = 0; f < in->n_frags; ++f) rows += in->num_rows[f];
    if (rows > (int64_t)UINT32_MAX) return MI355Q_ERR_INVALID_PLAN;  // a 32-bit COUNT would wrap
    mi355q_plan p8 = *plan;
    p8.bigint_count = 1;  // pick_target_compact_width: g_bigint_count -> 8-byte slots
    mi355q_qmd q8;
    if (int32_t e = qmd_init(p8, &q8)) return e;
    if (q8.slot_width != 8 || q8.entry_count != q.entry_count || q8.key_bytes != q.key_bytes ||
        q8.slot_count != q.slot_count)
      return MI355Q_ERR_UNSUPPORTED;
    if (reserved) {
      route_note("8-byte-slot twin + k_narrow_slots");
      return execute_impl(&p8, in, &o, out, report, nullptr, reserved);
    }
    DeviceCtx& ctx = ctx_of(in->device_id);
    std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
    const int64_t need = q8.entry_count * (int64_t)q8.row_size;
    if (ctx.wide_bytes < need) {
      if (ctx.wide) (void)hipFree(ctx.wide);
      ctx.wide = nullptr;
      ctx.wide_bytes = 0;
      hipError_t he = hipMalloc(&ctx.wide, (size_t)need);
      if (he != hipSuccess) {
        last_hip_error = he;
        return MI355Q_ERR_OUT_OF_GPU_MEM;
      }
      ctx.wide_bytes = need;
    }
    hipStream_t s4 = (hipStream_t)o.stream;
    if (!s4) {
      if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
      s4 = ctx.stream;
    }
    mi355q_exec_options o8 = o;
    o8.stream = s4;
    o8.out_buffer = ctx.wide;
    mi355q_result* r8 = nullptr;
    if (int32_t e = mi355q_execute(&p8, in, &o8, &r8, report)) return e;
    mi355q_result_free(r8);
    mi355q_result* res4 = nullptr;
    if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res4)) return e;
    hipError_t he = launch_narrow_slots((const int64_t*)ctx.wide, q8.row_size / 8, q.key_bytes / 8, q.slot_count,
                                        q.row_size / 8, q.entry_count, res4->buf, s4);
    if (he == hipSuccess) he = hipStreamSynchronize(s4);
    if (he != hipSuccess) {
      last_hip_error = he;
      mi355q_result_free(res4);
      return MI355Q_ERR_HIP;
    }
    *out = res4;
    return MI355Q_OK;
  }

  // (small multi-column perfect-hash tables: the LDS group-by computes the entry index from the key columns itself,
  // no packed index column needed)
  bool lds_direct = false, idx_direct = false;
  if (!o.force_generic && in->n_frags > 0 && (o.kernel_variant == 0 || o.kernel_variant == 2) &&
      (d.desc_type == MI355Q_GROUP_BY_PERFECT_HASH ||
       (d.desc_type == MI355Q_GROUP_BY_BASELINE_HASH && d.entry_count <= 65536 && !(o.flags & MI355Q_OPT_NO_LDS_BASELINE) && !pend))) {
    int64_t tr = 0, mr = 0;
    for (int f = 0; f < in->n_frags; ++f) {
      tr += in->num_rows[f];
      mr = std::max(mr, in->num_rows[f]);
    }
    FragView fvh{nullptr, nullptr, in->col_buffers, in->num_rows, in->n_frags, plan->n_cols, tr, mr};
    if (o.kernel_variant == 0) lds_direct = lds_groupby_eligible(d, fvh, n_cus);
    // perfect-hash tables too large for LDS over plain INT keys and values: partitioned by entry index with narrow
    // records (kernels_idx.hip) instead of packed keys + one exchange per value column (kernel_variant 2 = "the
    // large-input members" takes it whatever the input size: tests)
    if (!lds_direct && !pend && d.desc_type == MI355Q_GROUP_BY_PERFECT_HASH && (tr >= kIdxPartMinRows || o.kernel_variant == 2))
      lds_direct = idx_direct = idx_part_eligible(d, fvh, n_cus);
  }
  if (bf_step && in->n_frags <= 0) return kNotTaken;
  if (bf_step && q.desc_type != MI355Q_NON_GROUPED_AGGREGATE && (!lds_direct || idx_direct)) return kNotTaken;
  if (!pend && plan->join_outer_col >= 0 && !bf_step) {
    const size_t mark = t_route ? t_route->size() : 0;
    const int32_t e = execute_dense_join_as_filter(plan, in, o, out, report, reserved);
    if (e != kNotTaken) return e;
    if (t_route) t_route->resize(mark);
    *out = nullptr;
  }
  if (!o.force_generic && in->n_frags > 0 && !pend && plan->join_outer_col >= 0 && !bf_step) {
    const size_t mark = t_route ? t_route->size() : 0;
    const int32_t e = execute_join_gather(plan, in, o, q, d, n_cus, out, report, reserved);
    if (e != kNotTaken) return e;
    if (t_route) t_route->resize(mark);
    *out = nullptr;
  }
  if (!o.force_generic && in->n_frags > 0 && !lds_direct && !pend) {
    const size_t mark = t_route ? t_route->size() : 0;
    const int32_t e = execute_perfect_twin(plan, in, o, q, n_cus, out, report, reserved);
    if (e != kNotTaken) return e;
    if (t_route) t_route->resize(mark);
    *out = nullptr;
  }
  if (!o.force_generic && in->n_frags > 0 && !pend && !reserved) {
    // (before the LDS attempt on a small baseline table too: BH007's 10 K lattice points are a typed LDS member's work)
    const int32_t e = execute_affine_twin(plan, in, o, q, n_cus, out, report, reserved);
    if (e != kNotTaken) return e;
    *out = nullptr;
  }
  if (!o.force_generic && in->n_frags > 0 && !lds_direct) {
    const size_t mark = t_route ? t_route->size() : 0;
    const int32_t e = execute_packed_multi(plan, in, o, q, d, n_cus, out, report, reserved);
    if (e != kNotTaken) return e;
    if (t_route) t_route->resize(mark);
    *out = nullptr;
  }

  if (!o.force_generic && in->n_frags > 0 && !lds_direct && o.kernel_variant != 1 && d.n_group >= 1) {
    // several value columns over a large input, and the LDS group-by is not going to take the step (table too large,
    // a forced variant, or its attempts already failed): one run per value column, zipped
    const size_t mark = t_route ? t_route->size() : 0;
    const int32_t e = execute_multi_value(plan, in, o, q, d, out, report, reserved);
    if (e != kNotTaken && e != MI355Q_ERR_UNSUPPORTED) return e;
    if (t_route) t_route->resize(mark);
    *out = nullptr;
  }

  hipStream_t s = (hipStream_t)o.stream;  // caller's, or the device context's own (below)

  mi355q_result* res = nullptr;
  if (int32_t e = result_create_impl(&q, in->device_id, o.out_buffer, &res)) return e;
  struct ResGuard {
    mi355q_result* r;
    ~ResGuard() { mi355q_result_free(r); }
  } rg{res};

  // device copies of the fragment tables + error word, one allocation
  const int nf = in->n_frags, nc = plan->n_cols;
  int64_t total_rows = 0, max_frag_rows = 0;
  for (int f = 0; f < nf; ++f) {
    if (in->num_rows[f] < 0) return MI355Q_ERR_INVALID_PLAN;
    total_rows += in->num_rows[f];
    max_frag_rows = std::max(max_frag_rows, in->num_rows[f]);
  }
  const size_t ptr_bytes = sizeof(void*) * (size_t)std::max(1, nf * nc);
  const size_t rows_bytes = sizeof(int64_t) * (size_t)std::max(1, nf);
  const size_t meta_bytes = ptr_bytes + rows_bytes + 64;
  DeviceCtx& ctx = ctx_of(in->device_id);
  std::lock_guard<std::recursive_mutex> ctx_lock(ctx.mu);
  if (ctx.meta_bytes < meta_bytes) {
    if (ctx.meta) (void)hipFree(ctx.meta);
    ctx.meta = nullptr;
    ctx.meta_bytes = 0;
    if (ctx.h_meta) (void)hipHostFree(ctx.h_meta);
    ctx.h_meta = nullptr;
    ctx.h_ret_dev = nullptr;
    ctx.meta_shadow.clear();
    HIP_TRY(hipMalloc(&ctx.meta, meta_bytes * 2));
    HIP_TRY(hipHostMalloc((void**)&ctx.h_meta, meta_bytes * 2 + 64, hipHostMallocDefault));
    ctx.meta_bytes = meta_bytes * 2;
  }
  if (!ctx.h_ret_dev && ctx.h_meta) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, ctx.h_meta + ctx.meta_bytes, 0) == hipSuccess) ctx.h_ret_dev = (int32_t*)dp;
    else (void)hipGetLastError();
  }
  if (!s) {
    if (!ctx.stream) HIP_TRY(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
    s = ctx.stream;
  }
  char* mp = (char*)ctx.meta;
  const int8_t* const* d_cols = (const int8_t* const*)mp;
  const int64_t* d_rows = (const int64_t*)(mp + ptr_bytes);
  int32_t* d_err = (int32_t*)(mp + ptr_bytes + rows_bytes);
  {
    // one copy out of pinned memory: column table | row counts | zeroed error words (the synchronous step of a 60 us scan
    // used to spend ~90 us on a memset and two copies out of pageable memory, each staged by the driver)
    char* hm = ctx.h_meta;
    const size_t tab_bytes = ptr_bytes + rows_bytes;
    // the same table as the last upload, its error words untouched since: nothing to send
    bool same = !reserved && nf > 0 && ctx.meta_err_clean && ctx.meta_shadow.size() == tab_bytes &&
                !std::memcmp(ctx.meta_shadow.data(), in->col_buffers, sizeof(void*) * (size_t)(nf * nc)) &&
                !std::memcmp(ctx.meta_shadow.data() + ptr_bytes, in->num_rows, sizeof(int64_t) * (size_t)nf);
    if (!same && !reserved) {  // (reserve / explain launch nothing: no upload to leave in flight)
      if (nf > 0) {
        std::memcpy(hm, in->col_buffers, sizeof(void*) * (size_t)(nf * nc));
        std::memcpy(hm + ptr_bytes, in->num_rows, sizeof(int64_t) * (size_t)nf);
      }
      std::memset(hm + ptr_bytes + rows_bytes, 0, 64);
      const size_t lo = nf > 0 ? 0 : ptr_bytes + rows_bytes;
      HIP_TRY(hipMemcpyAsync(mp + lo, hm + lo, ptr_bytes + rows_bytes + 64 - lo, hipMemcpyHostToDevice, s));
      if (nf > 0) ctx.meta_shadow.assign(hm, hm + tab_bytes);
      else ctx.meta_shadow.clear();
    }
    ctx.meta_err_clean = false;  // (until this step's words have been read back as zeros: below)
  }
  // The upload reads the SHARED pinned block: a return that does not reach finish_step (an allocation that fails, a HIP
  // error, kNotTaken) must not leave it in flight — the next call on another stream rewrites h_meta and ctx.meta while
  // the stale copy could still land.  Disarmed where the step synchronises itself or stays in flight by
  // design (mi355q_execute_async: the next call drains it first).
  struct UploadGuard {
    hipStream_t s;
    bool armed;
    ~UploadGuard() {
      if (armed) (void)hipStreamSynchronize(s);
    }
  } upload_guard{s, !reserved};

  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  LaunchStats st;
  constexpr int kEvPool = 128;
  hipEvent_t* ev_pool = nullptr;
  if (report) {
    while ((int)ctx.events.size() < kEvPool + 4) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      ctx.events.push_back(e);
    }
    ev_start = ctx.events[0];
    ev_stop = ctx.events[1];
    st.k_start = ctx.events[2];
    st.k_stop = ctx.events[3];
    ev_pool = ctx.events.data() + 4;
    st.ev_pool = ev_pool;
    st.n_ev = kEvPool;
  }

  FragView fv{d_cols, d_rows, in->col_buffers, in->num_rows, nf, nc, total_rows, max_frag_rows};

  // ---- plan-time kernel selection (a fixed family; no JIT)
  StepKind kind = K_GENERIC;
  JoinPayloadView pay{};
  if (bf_step) {
    if (!o.force_generic && nf > 0 && scan_agg_eligible(d, fv)) kind = K_SCAN_AGG;
    else if (!o.force_generic && nf > 0 && o.kernel_variant == 0 && lds_groupby_eligible(d, fv, n_cus)) kind = K_LDS_GROUPBY;
  } else if (!o.force_generic && nf > 0) {
    if (scan_count_eligible(d, fv)) kind = K_SCAN_COUNT;
    else if (o.kernel_variant != 1 && scan_agg_eligible(d, fv)) kind = K_SCAN_AGG;
    // (a 1-byte filter column — the row mask of a compiled filter — in front of a few-groups step: the typed LDS member reads it
    // at 1.98 ms per 1 B rows, k_perfect_lds's 4-byte quad loads at ~3 ms: measured in round 6)
    else if (!(d.n_quals == 1 && d.quals[0].type == MI355Q_INT8 && o.kernel_variant == 0 && lds_groupby_typed_eligible(d, fv, n_cus)) &&
             perfect_lds_eligible(d, fv))
      kind = K_PERFECT_LDS;
    // few groups: the table replicated in every workgroup's LDS (perfect-hash layouts up to 64 K entries that fit;
    // baseline layouts whose entry guess says "small" — if the groups turn out to be too many the step is re-run)
    else if (o.kernel_variant == 0 && lds_groupby_eligible(d, fv, n_cus) &&
             (d.desc_type == MI355Q_GROUP_BY_PERFECT_HASH ||
              (d.entry_count <= 65536 && !(o.flags & MI355Q_OPT_NO_LDS_BASELINE) && !pend)))
      kind = K_LDS_GROUPBY;
    else if (idx_direct && idx_part_eligible(d, fv, n_cus)) kind = K_IDX_PART;
    else if (baseline_fast_eligible(d, fv)) kind = K_BASELINE_FAST;
    else if (join_sum_eligible(d, fv)) kind = K_JOIN_SUM;
    // semi-join + aggregate over a large fact table: radix-partitioned probe (bitmap slices in
    // LDS) instead of one random bitmap read per row; kernel_variant 1 / 2 force either member
    if (kind == K_JOIN_SUM && o.kernel_variant != 1 && join_part_supported(d, fv, n_cus) &&
        (o.kernel_variant == 2 || total_rows >= ((int64_t)64 << 20)))
      kind = K_JOIN_PART;
  }

  // SEVERAL plain quals in front of a family that filters on one column (the partitioned GROUP BY, the perfect-hash LDS
  // member, the index-partitioned family takes none): instead of the row kernel, the quals become the range atoms of a
  // compiled filter, the row-mask pre-pass evaluates them (k_filter_mask: one byte per row) and the step runs on
  // `mask = 1` — measured in round 6: `a < K AND b > L` over 10 M INT64 groups went to k_generic at 55 ms per 1 B rows
  if (kind == K_GENERIC && !bf_step && !o.force_generic && nf > 0 && !pend && plan->n_exprs == 0 && plan->n_quals >= 2 &&
      d.join_col < 0 && q.desc_type != MI355Q_PROJECTION && (total_rows >= ((int64_t)1 << 20) || o.kernel_variant == 2) &&
      !(o.flags & MI355Q_OPT_NO_COMPILED_FILTER)) {
    BoolFilterHost bfh;
    mi355q_plan rest;
    if (compile_bool_filter(*plan, &bfh, &rest)) {
      const size_t mark = t_route ? t_route->size() : 0;
      route_note("quals compiled (range atoms + truth table)");
      const int32_t e = execute_masked(plan, rest, bfh, in, o, out, report, reserved);
      if (e != kNotTaken) return e;
      if (t_route) t_route->resize(mark);
      *out = nullptr;
    }
  }

  // joins that read the inner side / one-to-many tables / LEFT joins over a large outer table: the
  // payload probe (per-key aggregated payload of the perfect table in LDS).  The semi-join shapes
  // keep their 1-bit-per-key member above.
  // The L2 members of that probe map workgroup b to XCD b % 8 and group b / 8 and walk the runs with stride
  // gridDim / 8: they are planned and launched with a multiple of 8 workgroups, or not at all (found by the host
  // simulation, which first ran them on "4 CUs": gridDim / 8 = 0 never advances — on the device a `tune_cus` below 8
  // would have hung the GPU, and one that is not a multiple of 8 would have read some runs twice).
  const int n_cus_probe = n_cus >= 8 ? (n_cus & ~7) : 0;
  if (!o.force_generic && nf > 0 && kind != K_JOIN_PART && o.kernel_variant != 1 && plan->join_table && !reserved &&
      d.join_col >= 0 && (o.kernel_variant == 3 || total_rows >= ((int64_t)16 << 20))) {
    int wcol = -1, l2 = 0;
    if (n_cus_probe && join_probe_wants(d, fv, n_cus_probe, &wcol, &l2)) {
      mi355q_join_table* jt = const_cast<mi355q_join_table*>(plan->join_table);
      const void* inner = wcol >= 0 ? (const void*)d.inner_cols[wcol] : nullptr;
      std::lock_guard<std::mutex> pl(jt->pay_mu);
      const int64_t entries = jt->entry_count;
      bool ok = !(jt->pay_refused && jt->pay_refused_col == inner && jt->pay_refused_rows == total_rows);
      // a cached payload is only as good as the column it was derived from: same address AND same generation
      const bool have = l2 ? (jt->pay16_built && (!inner || (jt->pay16_col == inner && jt->pay16_version == in->inner_version)))
                           : (jt->pay_col_built && (!inner || (jt->pay_col == inner && jt->pay_version == in->inner_version)));
      if (ok && !have) {
        // (re)build for this inner column, in the layout the chosen mode reads
        hipEvent_t b0 = nullptr, b1 = nullptr;
        (void)hipEventCreate(&b0);
        (void)hipEventCreate(&b1);
        DevWord flags;
        ok = hipMalloc(&flags.p, 64) == hipSuccess;
        if (l2) {
          if (ok && !jt->pay16) ok = hipMalloc(&jt->pay16, (size_t)entries * 16) == hipSuccess;
          // one-to-one tables: the 8-byte payload — interleaved with the slot's key for a keyed table
          if (ok && (jt->hash_type == 0 || jt->hash_type == 1) && !jt->pay8)
            ok = hipMalloc((void**)&jt->pay8, (size_t)entries * (l2 == 2 ? 16 : 8)) == hipSuccess;
          // (one spare key behind the end: the keyed probe reads the keys two at a time)
          if (ok && l2 == 2 && !jt->pay_kkeys) ok = hipMalloc((void**)&jt->pay_kkeys, (size_t)entries * 8 + 16) == hipSuccess;
        } else {
          if (ok && !jt->pay_cnt) ok = hipMalloc((void**)&jt->pay_cnt, (size_t)entries * 4) == hipSuccess;
          if (ok && inner && !jt->pay_wsum) ok = hipMalloc((void**)&jt->pay_wsum, (size_t)entries * 8) == hipSuccess;
          if (ok && inner && !jt->pay_wnn) ok = hipMalloc((void**)&jt->pay_wnn, (size_t)entries * 4) == hipSuccess;
        }
        if (ok) {
          (void)hipMemsetAsync(flags.p, 0, 64, s);
          if (b0) (void)hipEventRecord(b0, s);
          ok = (l2 == 2 ? launch_join_payload_keyed_build(jt->buf, jt->hash_type, entries, inner, jt->pay_kkeys, jt->pay16,
                                                          jt->pay8, (int32_t*)flags.p, n_cus, s)
                        : launch_join_payload_build(jt->buf, jt->hash_type, entries, inner, jt->pay_cnt, jt->pay_wsum,
                                                    jt->pay_wnn, l2 ? jt->pay16 : nullptr, l2 ? jt->pay8 : nullptr,
                                                    (int32_t*)flags.p, n_cus, s)) == hipSuccess;
          if (b1) (void)hipEventRecord(b1, s);
          int32_t h_flags = 0;
          ok = ok && hipMemcpyAsync(&h_flags, flags.p, 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
               hipStreamSynchronize(s) == hipSuccess;
          if (ok) {
            if (l2) {
              jt->pay16_built = true;
              jt->pay16_col = inner;
              jt->pay16_version = in->inner_version;
              jt->pay16_has_nulls = h_flags & 1;
            } else {
              jt->pay_col_built = true;
              jt->pay_col = inner;
              jt->pay_version = in->inner_version;
              jt->pay_has_nulls = h_flags & 1;
            }
            if (b0 && b1) (void)hipEventElapsedTime(&jt->pay_build_ms, b0, b1);
          }
        }
        if (b0) (void)hipEventDestroy(b0);
        if (b1) (void)hipEventDestroy(b1);
        if (!ok) (void)hipGetLastError();
      }
      if (ok) {
        pay.cnt_k = l2 ? nullptr : jt->pay_cnt;
        pay.wsum_k = (!l2 && inner) ? jt->pay_wsum : nullptr;
        pay.wnn_k = (!l2 && inner) ? jt->pay_wnn : nullptr;
        pay.pay16 = l2 ? jt->pay16 : nullptr;
        pay.pay8 = l2 ? jt->pay8 : nullptr;
        pay.kkeys = l2 == 2 ? jt->pay_kkeys : nullptr;
        pay.inner_col = inner;
        pay.entries = entries;
        pay.has_nulls = l2 ? jt->pay16_has_nulls : jt->pay_has_nulls;
        if (join_probe_supported(d, fv, pay, n_cus_probe)) {
          kind = K_JOIN_PROBE;
        } else {
          // the probe plan does not take this payload after all: entries x 16 B of device memory are not kept for a
          // member that will not run (the step falls back to k_join_sum / the row kernel below)
          auto drop = [](auto*& ptr) {
            if (ptr) (void)hipFree((void*)ptr);
            ptr = nullptr;
          };
          if (l2) {
            drop(jt->pay16);
            drop(jt->pay8);
            drop(jt->pay_kkeys);
            jt->pay16_built = false;
            jt->pay16_col = nullptr;
          } else {
            drop(jt->pay_cnt);
            drop(jt->pay_wsum);
            drop(jt->pay_wnn);
            jt->pay_col_built = false;
            jt->pay_col = nullptr;
          }
          pay = JoinPayloadView{};
          jt->pay_refused = true;
          jt->pay_refused_col = inner;
          jt->pay_refused_rows = total_rows;
        }
      }
    }
  }

  if (bf_step && kind != K_LDS_GROUPBY && kind != K_SCAN_AGG) return kNotTaken;
  tr.mark("setup done");
  int64_t scratch_bytes = 0;
  int64_t scratch_cap = o.scratch_bytes;
  if (kind == K_BASELINE_FAST || kind == K_JOIN_PART || kind == K_JOIN_PROBE || kind == K_IDX_PART) {
    // default cap: kDefaultScratchCap, but never more than 70 % of what the device has free right now
    // (counting the scratch this context already holds), halved while the device cannot provide it
    // (the planner then cuts the input into more chunks)
    if (scratch_cap <= 0) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const int64_t spare = (int64_t)((double)((int64_t)free_b + ctx.scratch_bytes) * 0.7);
        scratch_cap = std::max<int64_t>(std::min<int64_t>(kDefaultScratchCap, spare), (int64_t)1 << 30);
      }
    }
    for (;;) {
      scratch_bytes = kind == K_JOIN_PART    ? join_part_scratch_bytes(d, fv, n_cus, scratch_cap)
                      : kind == K_IDX_PART   ? idx_part_scratch_bytes(d, fv, n_cus, scratch_cap)
                      : kind == K_JOIN_PROBE ? join_probe_scratch_bytes(d, fv, pay, n_cus_probe, scratch_cap)
                                             : baseline_fast_scratch_bytes(d, fv, o.kernel_variant, scratch_cap, n_cus);
      if (kind == K_JOIN_PROBE && scratch_bytes == 0) {  // no plan within this cap: the row kernel
        kind = join_sum_eligible(d, fv) ? K_JOIN_SUM : K_GENERIC;
        break;
      }
      if (kind == K_IDX_PART && scratch_bytes == 0) return MI355Q_ERR_OUT_OF_GPU_MEM;  // (eligible means it plans)
      if (kind == K_JOIN_PART && scratch_bytes == 0) {  // no plan within this cap: direct probe
        kind = K_JOIN_SUM;
        break;
      }
      if (scratch_bytes <= ctx.scratch_bytes || t_plan_only) break;
      if (ctx.scratch) (void)hipFree(ctx.scratch);
      ctx.scratch = nullptr;
      ctx.scratch_bytes = 0;
      hipError_t e = hipMalloc(&ctx.scratch, (size_t)scratch_bytes);
      if (e == hipSuccess) {
        ctx.scratch_bytes = scratch_bytes;
        break;
      }
      (void)hipGetLastError();
      const int64_t cur_cap = scratch_cap > 0 ? scratch_cap : kDefaultScratchCap;
      if (cur_cap <= ((int64_t)1 << 30)) {
        last_hip_error = e;
        return MI355Q_ERR_OUT_OF_GPU_MEM;
      }
      scratch_cap = cur_cap / 2;
    }
  }

  tr.mark("scratch allocated");
  if (reserved) {  // mi355q_reserve_workspace / mi355q_explain: nothing is launched
    if (t_route) {
      char note[96];
      const char* name = kind == K_SCAN_COUNT ? "k_scan_count" : kind == K_SCAN_AGG ? "k_scan_agg"
                         : kind == K_PERFECT_LDS ? "k_perfect_lds" : kind == K_LDS_GROUPBY ? "k_groupby_lds"
                         : kind == K_JOIN_SUM ? "k_join_sum" : kind == K_JOIN_PART ? "k_part_scatter + k_part_join"
                         : kind == K_JOIN_PROBE ? "k_part_scatter + k_part_probe"
                         : kind == K_IDX_PART ? "k_idx_scatter + k_idx_aggregate"
                         : kind == K_BASELINE_FAST
                             ? (nf > 0 && baseline_fast_variant(d, fv, o.kernel_variant, n_cus) == 2 ? "k_part_scatter + k_part_aggregate"
                                                                                                      : "k_baseline_direct")
                             : "k_generic";
      std::snprintf(note, sizeof(note), "%s", name);
      route_note(note);
    }
    *reserved = t_plan_only ? scratch_bytes : ctx.scratch_bytes;
    return MI355Q_OK;
  }
  // a scan step is ONE kernel of tens of microseconds behind a one-row initialisation: its own event pair (k_start / k_stop,
  // recorded by the launch function) is the step's time as well — two event commands fewer on the stream, ~3 us each
  const bool one_kernel_step = nf > 0 && (kind == K_SCAN_COUNT || kind == K_SCAN_AGG);
  if (one_kernel_step) ev_start = ev_stop = nullptr;
  if (ev_start) HIP_TRY(hipEventRecord(ev_start, s));
  // the partitioned member writes every row of the table itself (empty rows included)
  const bool self_init = kind == K_BASELINE_FAST && nf > 0 &&
                         baseline_fast_variant(d, fv, o.kernel_variant, n_cus) == 2;
  if (!self_init) HIP_TRY(launch_init_buffer(res->buf, q.entry_count, make_row_init(q), s));

  if (nf > 0) {
    switch (kind) {
      case K_SCAN_COUNT:
        HIP_TRY(launch_scan_count(d, fv, res->buf, n_cus, s, &st));
        break;
      case K_SCAN_AGG:
        HIP_TRY(launch_scan_agg(d, fv, res->buf, n_cus, s, &st));
        break;
      case K_PERFECT_LDS:
        HIP_TRY(launch_perfect_lds(d, fv, res->buf, d_err, n_cus, s, &st));
        break;
      case K_LDS_GROUPBY:
        HIP_TRY(launch_lds_groupby(d, fv, res->buf, d_err, n_cus, s, &st));
        break;
      case K_BASELINE_FAST:
        HIP_TRY(launch_baseline_fast(d, fv, res->buf, d_err, ctx.scratch, ctx.scratch_bytes,
                                     scratch_cap, o.kernel_variant, n_cus, s, &st));
        break;
      case K_IDX_PART:
        HIP_TRY(launch_idx_partitioned(d, fv, res->buf, d_err, ctx.scratch, ctx.scratch_bytes, scratch_cap, n_cus, s, &st));
        break;
      case K_JOIN_SUM:
        HIP_TRY(launch_join_sum(d, fv, res->buf, n_cus, s, &st));
        break;
      case K_JOIN_PART:
        HIP_TRY(launch_join_partitioned(d, fv, res->buf, d_err, ctx.scratch, ctx.scratch_bytes, scratch_cap, n_cus,
                                        s, &st));
        break;
      case K_JOIN_PROBE:
        HIP_TRY(launch_join_probe(d, fv, pay, res->buf, d_err, ctx.scratch, ctx.scratch_bytes, scratch_cap, n_cus_probe, s,
                                  &st));
        break;
      default:
        st.kernel_name = "k_generic";
        st.n_launches = 1;
        if (st.k_start) HIP_TRY(hipEventRecord(st.k_start, s));
        HIP_TRY(launch_generic(d, q.idx_target_as_key, make_row_init(q), d_cols, d_rows, nf, max_frag_rows,
                               res->buf, d_err, n_cus, s));
        if (st.k_stop) HIP_TRY(hipEventRecord(st.k_stop, s));
    }
  }
  if (st.spill_counter32) {
    // the counter lives in the partition scratch, which mi355q_shard_merge_slices (or the next call) may reuse or
    // reallocate before mi355q_wait looks at it: keep a copy next to the error words
    HIP_TRY(hipMemcpyAsync(d_err + 4, st.spill_counter32, 4, hipMemcpyDeviceToDevice, s));
    st.spill_counter32 = (uint32_t*)(d_err + 4);
  }
  if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, s));
  tr.mark("launched");

  TailState* tail = new (std::nothrow) TailState();
  if (!tail) return MI355Q_ERR_OUT_OF_CPU_MEM;
  tail->q = q;
  tail->d = d;
  tail->kind = kind;
  tail->st = st;
  tail->res = res;
  tail->d_err = d_err;
  tail->d_cols = d_cols;
  tail->d_rows = d_rows;
  tail->h_cols.assign(in->col_buffers, in->col_buffers + (size_t)nf * nc);
  tail->h_rows.assign(in->num_rows, in->num_rows + nf);
  tail->nf = nf;
  tail->nc = nc;
  tail->n_cus = n_cus;
  tail->total_rows = total_rows;
  tail->max_frag_rows = max_frag_rows;
  tail->alg_bytes = algorithmic_bytes(*plan, *in);
  tail->s = s;
  tail->ev_start = ev_start;
  tail->ev_stop = ev_stop;
  tail->ev_pool = ev_pool;
  tail->trace = tr.on;
  tail->knobs = tune_knobs();
  tail->h_ret = (int32_t*)(ctx.h_meta + ctx.meta_bytes);
  tail->h_ret_dev = ctx.h_ret_dev;
  upload_guard.armed = false;  // (finish_step synchronises the stream; a pending step is drained before the next call)
  if (pend) {  // mi355q_execute_async: the rest runs in mi355q_wait (or before the next call on this device)
    mi355q_pending* p = new (std::nothrow) mi355q_pending();
    if (!p) {
      delete tail;
      return MI355Q_ERR_OUT_OF_CPU_MEM;
    }
    p->tail = tail;
    p->device_id = in->device_id;
    ctx.inflight = p;
    *pend = p;
    rg.r = nullptr;
    *out = res;
    return MI355Q_OK;
  }
  const int32_t code = finish_step(*tail, report);
  ctx.meta_err_clean = code == MI355Q_OK && tail->err_words_zero;
  delete tail;
  tr.mark("synchronized");
  if (code == kRetryNoIdx) {
    mi355q_result_free(res);
    rg.r = nullptr;
    mi355q_exec_options o2 = o;
    o2.flags |= MI355Q_OPT_NO_IDX_PART;
    return execute_impl(plan, in, &o2, out, report, nullptr, nullptr);
  }
  if (code == kRetryNoLds) {
    mi355q_result_free(res);
    rg.r = nullptr;
    mi355q_exec_options o2 = o;
    // small replicas did not hold the groups: the largest replica next, then eight windows of it, then another family
    // (skipping the windows over large inputs was measured and is worse: with <= 65536 entries the families behind them
    // are the direct global-atomic members — BH007 at 1 B rows: 29.6 ms in eight windows, 638 ms without,
    // profiles/r04_refbench_1b_call14.jsonl)
    o2.flags |= !(o.flags & MI355Q_OPT_LDS_BASELINE_LARGE)     ? MI355Q_OPT_LDS_BASELINE_LARGE
                : !(o.flags & MI355Q_OPT_LDS_BASELINE_WINDOWS) ? MI355Q_OPT_LDS_BASELINE_WINDOWS
                                                               : MI355Q_OPT_NO_LDS_BASELINE;
    return execute_impl(plan, in, &o2, out, report, nullptr, nullptr);
  }
  if (code) return code;
  rg.r = nullptr;
  *out = res;
  return MI355Q_OK;
}
}  // namespace

// ------------------------------------------------------------------------------- synth
int32_t mi355q_generate_column(int32_t device_id, void* dst, int64_t n_rows, int64_t row_offset,
                               int32_t kind, uint64_t seed, int64_t a, int64_t b, int64_t c,
                               double a_f, int32_t null_every, void* stream) {
  if (!dst || n_rows < 0 || kind < MI355Q_GEN_I32_UNIFORM31 || kind > MI355Q_GEN_F64_UNIT)
    return MI355Q_ERR_INVALID_PLAN;
  if ((kind == MI355Q_GEN_I32_MOD || kind == MI355Q_GEN_I64_MOD || kind == MI355Q_GEN_I64_MOD_MUL) &&
      a <= 0)
    return MI355Q_ERR_INVALID_PLAN;
  DeviceGuard g(device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  HIP_TRY(launch_generate(dst, n_rows, row_offset, kind, seed, a, b, c, a_f, null_every,
                          (hipStream_t)stream));
  return MI355Q_OK;
}

}  // extern "C"
