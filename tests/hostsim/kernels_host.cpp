// TEST INFRASTRUCTURE (tests/hostsim): stand-ins for the kernel families that cannot be simulated block by block
// (kernels_fast.hip, kernels_lds.hip, kernels_part.hip, kernels_sort.hip: wave-level pipelines, workgroups that wait
// for each other, rocPRIM).  What the host library (api.cpp) sees of them is kept: who is eligible for which plan
// shape, what a launch leaves in the output table (computed here by the product's own row function, rowfunc.h, one
// row at a time), and the ways they hand a step back — an LDS replica that runs out of room (d_err[1]), a spill list
// that overflows (d_err[1]) — which tests switch on through hostsim_configure().  The point of the simulation is
// api.cpp's control flow (routes, derived plans, retries, workspaces, result layouts), which no CPU test reached
// before; the fast kernels themselves are tested on the device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "kernels.h"
#include "rowfunc.h"

namespace {
struct SimConfig {
  uint32_t routes = 0xffffffffu;   // bit per family, see hostsim_configure
  int32_t lds_small_groups = 128;  // groups a "small" / "large" baseline LDS replica holds
  int32_t lds_large_groups = 2048;
  int32_t part_overflows = 0;      // the next N partitioned launches report a spill overflow
  int32_t launches[16] = {};       // per family, since the last configure
} g_cfg;
enum Family { F_SCAN_COUNT = 0, F_SCAN_AGG, F_PERFECT_LDS, F_LDS_GROUPBY, F_BASELINE_DIRECT, F_BASELINE_PART,
              F_JOIN_SUM };
bool on(Family f) { return (g_cfg.routes >> f) & 1u; }
}  // namespace

extern "C" {
// routes: bit i enables family i (scan_count, scan_agg, perfect_lds, lds_groupby, baseline direct, baseline
// partitioned, join_sum); part_overflows: that many partitioned launches end with "spill list overflowed"
void hostsim_configure(uint32_t routes, int32_t lds_small_groups, int32_t lds_large_groups, int32_t part_overflows) {
  g_cfg = SimConfig();
  g_cfg.routes = routes;
  if (lds_small_groups > 0) g_cfg.lds_small_groups = lds_small_groups;
  if (lds_large_groups > 0) g_cfg.lds_large_groups = lds_large_groups;
  g_cfg.part_overflows = part_overflows;
}
int32_t hostsim_launches(int32_t family) { return family >= 0 && family < 16 ? g_cfg.launches[family] : -1; }
}

#ifndef HOSTSIM_REAL_FAST   // (HOSTSIM_REAL_FAST: the real kernels_fast / _lds / _part / _sort .hip are compiled in instead)
namespace mq {

namespace {

// the whole step through the row function, the way k_generic does it without its grid
int32_t run_rows(const DevPlan& p, const FragView& fv, int64_t* out) {
  const bool ng = p.desc_type == MI355Q_NON_GROUPED_AGGREGATE;
  int64_t loc[MI355Q_MAX_SLOTS + 1];
  for (int s = 0; s < p.slot_count; ++s) loc[s] = p.init_vals[s];
  for (int f = 0; f < fv.n_frags; ++f) {
    const int8_t* const* cols = fv.d_cols + (size_t)f * fv.n_cols;
    for (int64_t pos = 0; pos < fv.d_num_rows[f]; ++pos) {
      const int32_t e = ng ? process_row<false>(p, cols, pos, out, loc) : process_row<true>(p, cols, pos, out, nullptr);
      if (e) return e;
    }
  }
  if (ng) {
    for (int ti = 0; ti < p.n_targets; ++ti) {
      const DevTarget& t = p.targets[ti];
      if (t.slot < 0) continue;
      reduce_target<true>(t, p.init_vals, out, loc);
    }
  }
  return 0;
}

void finish(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, LaunchStats* st, const char* name,
            int variant, Family f) {
  const int32_t e = run_rows(p, fv, out);
  if (e && d_err) atomicCAS(d_err, 0, e);
  if (st) {
    st->kernel_name = name;
    st->n_launches = 1;
    st->variant = variant;
  }
  ++g_cfg.launches[f];
}

bool plain_int_or_double(int code) { return code == MI355Q_INT64 || code == MI355Q_INT32 || code == MI355Q_DOUBLE; }

// the single-value shapes of the partitioned / direct / perfect-LDS families (fast_common.h grouped_fast_shape):
// one group column, at most one integer qual, no join, every value aggregate over ONE plain column
bool one_value_shape(const DevPlan& p) {
  if (p.join_col >= 0 || p.n_quals > 1 || p.n_group != 1 || p.col0_key_quirk) return false;
  if (p.group_nullable && p.desc_type != MI355Q_GROUP_BY_BASELINE_HASH) return false;
  if (p.n_quals == 1 && p.quals[0].type != MI355Q_INT32 && p.quals[0].type != MI355Q_INT64) return false;
  int vcol = -1;
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    if (t.table != 0) return false;
    if (t.agg == MI355Q_PROJECT_KEY) continue;
    if (t.agg == MI355Q_COUNT && (t.col < 0 || !t.skip_null)) continue;
    if (t.col < 0 || !plain_int_or_double(t.arg_type)) return false;
    if (vcol >= 0 && vcol != t.col) return false;
    vcol = t.col;
    if (t.agg != MI355Q_COUNT && t.agg != MI355Q_SUM && t.agg != MI355Q_AVG && t.agg != MI355Q_MIN && t.agg != MI355Q_MAX)
      return false;
  }
  return true;
}

bool plain_aggs(const DevPlan& p, int max_value_cols) {
  int cols[MI355Q_MAX_TARGETS], n = 0;
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    if (t.agg == MI355Q_PROJECT_KEY) continue;
    if (t.table != 0 || t.arg_f32) return false;
    if (t.agg == MI355Q_COUNT && t.col < 0) continue;
    if (t.agg != MI355Q_COUNT && t.agg != MI355Q_SUM && t.agg != MI355Q_MIN && t.agg != MI355Q_MAX && t.agg != MI355Q_AVG) return false;
    if (t.col < 0 || !plain_int_or_double(t.arg_type)) return false;
    bool seen = false;
    for (int k = 0; k < n; ++k) seen = seen || cols[k] == t.col;
    if (!seen) cols[n++] = t.col;
  }
  return n <= max_value_cols;
}

int64_t live_entries(const DevPlan& p, const int64_t* out) {
  int64_t n = 0;
  for (int64_t e = 0; e < p.entry_count; ++e) n += out[e * p.row_quad] != kEmptyKey64;
  return n;
}

}  // namespace

// ---- non-grouped scans
bool scan_count_eligible(const DevPlan& p, const FragView&) {
  return on(F_SCAN_COUNT) && p.desc_type == MI355Q_NON_GROUPED_AGGREGATE && p.join_col < 0 && p.n_targets == 1 &&
         p.targets[0].agg == MI355Q_COUNT && p.targets[0].col < 0 && p.n_quals == 1 && p.slot_width == 8;
}
hipError_t launch_scan_count(const DevPlan& p, const FragView& fv, int64_t* out, int, hipStream_t, LaunchStats* st) {
  finish(p, fv, out, nullptr, st, "k_scan_count", 0, F_SCAN_COUNT);
  return hipSuccess;
}
bool scan_agg_eligible(const DevPlan& p, const FragView&) {
  if (p.bf_active) return false;  // (the stand-in runs the row function on the plan's quals: it takes no compiled filter)
  if (!on(F_SCAN_AGG) || p.desc_type != MI355Q_NON_GROUPED_AGGREGATE || p.join_col >= 0 || p.slot_width != 8) return false;
  for (int i = 0; i < p.n_quals; ++i)
    if (p.quals[i].type != MI355Q_INT32 && p.quals[i].type != MI355Q_INT64) return false;
  return plain_aggs(p, 8);
}
hipError_t launch_scan_agg(const DevPlan& p, const FragView& fv, int64_t* out, int, hipStream_t, LaunchStats* st) {
  finish(p, fv, out, nullptr, st, "k_scan_agg", 0, F_SCAN_AGG);
  return hipSuccess;
}

// ---- small tables in LDS
bool perfect_lds_eligible(const DevPlan& p, const FragView&) {
  return on(F_PERFECT_LDS) && p.desc_type == MI355Q_GROUP_BY_PERFECT_HASH && p.slot_width == 8 && !p.group_nullable &&
         (p.group_type == MI355Q_INT32 || p.group_type == MI355Q_INT64) && p.group_bucket[0] == 0 &&
         p.entry_count * (int64_t)p.row_quad * 8 <= 64 * 1024 && one_value_shape(p);
}
hipError_t launch_perfect_lds(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, int, hipStream_t,
                              LaunchStats* st) {
  finish(p, fv, out, d_err, st, "k_perfect_lds", 0, F_PERFECT_LDS);
  return hipSuccess;
}
bool lds_groupby_eligible(const DevPlan& p, const FragView&, int) {
  if (p.bf_active) return false;  // (the stand-in runs the row function on the plan's quals: it takes no compiled filter)
  if (!on(F_LDS_GROUPBY) || p.join_col >= 0 || p.col0_key_quirk || p.slot_width != 8 || p.n_quals > MI355Q_MAX_QUALS) return false;
  for (int i = 0; i < p.n_quals; ++i)
    if (p.quals[i].type != MI355Q_INT32 && p.quals[i].type != MI355Q_INT64) return false;
  if (p.desc_type == MI355Q_GROUP_BY_PERFECT_HASH) {
    if (p.n_group < 1 || p.n_group > 3 || p.entry_count > 8 * 4096) return false;   // "fits eight windows of a replica"
    for (int g = 0; g < p.n_group; ++g)
      if ((p.group_types[g] != MI355Q_INT32 && p.group_types[g] != MI355Q_INT64) || p.group_bucket[g]) return false;
  } else if (p.desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
    if (p.n_group != 1) return false;
    const int kt = p.group_types[0];
    if (kt != MI355Q_INT64 && kt != MI355Q_DOUBLE && kt != MI355Q_INT32 && kt != MI355Q_FLOAT) return false;
  } else {
    return false;
  }
  return plain_aggs(p, 3);
}
bool lds_groupby_typed_eligible(const DevPlan&, const FragView&, int) { return false; }  // (the stand-in has no typed members)
hipError_t launch_lds_groupby(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, int, hipStream_t,
                              LaunchStats* st) {
  finish(p, fv, out, d_err, st, "k_groupby_lds", 4, F_LDS_GROUPBY);
  if (p.desc_type == MI355Q_GROUP_BY_BASELINE_HASH) {
    const uint32_t fl = tune_knobs().flags;
    const int64_t cap = (fl & MI355Q_OPT_LDS_BASELINE_WINDOWS) ? 8 * (int64_t)g_cfg.lds_large_groups
                        : (fl & MI355Q_OPT_LDS_BASELINE_LARGE) ? g_cfg.lds_large_groups : g_cfg.lds_small_groups;
    if (live_entries(p, out) > cap) {
      // a replica ran out of room: what the table holds now is not a result
      for (int64_t i = 0; i < p.entry_count * p.row_quad; ++i) out[i] = 0x6b6b6b6b6b6b6b6bll;
      atomicExch(d_err + 1, 1);
    }
  }
  return hipSuccess;
}

// ---- baseline hash, one int64 / double key, one value column
bool baseline_fast_eligible(const DevPlan& p, const FragView&) {
  if (!(on(F_BASELINE_DIRECT) || on(F_BASELINE_PART))) return false;
  if (p.desc_type != MI355Q_GROUP_BY_BASELINE_HASH || p.key_width != 8 || p.slot_width != 8) return false;
  if (p.group_type != MI355Q_INT64 && p.group_type != MI355Q_DOUBLE) return false;
  return one_value_shape(p);
}
bool part_supported(const DevPlan& p, const FragView& fv, int) { return on(F_BASELINE_PART) && baseline_fast_eligible(p, fv); }
int64_t part_scratch_bytes(const DevPlan&, const FragView& fv, int, int64_t cap_bytes) {
  const int64_t want = fv.total_rows * 16 + 4096;
  return cap_bytes > 0 ? std::min(want, cap_bytes) : want;
}
int baseline_fast_variant(const DevPlan& p, const FragView& fv, int requested, int n_cus) {
  if (requested == 1) return 1;
  const bool can_part = part_supported(p, fv, n_cus);
  if (requested == 2) return can_part ? 2 : 1;
  if (!can_part || fv.total_rows < (int64_t)8 << 20 || p.entry_count < 65536) return 1;
  return 2;
}
int64_t baseline_fast_scratch_bytes(const DevPlan& p, const FragView& fv, int variant, int64_t cap_bytes, int n_cus) {
  return baseline_fast_variant(p, fv, variant, n_cus) == 1 ? 0 : part_scratch_bytes(p, fv, n_cus, cap_bytes);
}
hipError_t launch_baseline_partitioned(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, void* scratch,
                                       int64_t scratch_bytes, int64_t, int, hipStream_t, LaunchStats* st) {
  if (!scratch || scratch_bytes < 4096) return hipErrorInvalidValue;   // the caller sized and passed the workspace
  std::memset(scratch, 0x11, (size_t)std::min<int64_t>(scratch_bytes, 1 << 20));  // and it is writable
  // this member writes every row of the table itself, empty rows included (the caller skips its init pass)
  for (int64_t e = 0; e < p.entry_count; ++e) {
    int64_t* row = out + e * p.row_quad;
    for (int k = 0; k < p.key_quad; ++k) row[k] = kEmptyKey64;
    for (int j = 0; j < p.slot_count; ++j) row[p.key_quad + j] = p.init_vals[j];
  }
  finish(p, fv, out, d_err, st, "k_part_scatter", 2, F_BASELINE_PART);
  if (g_cfg.part_overflows > 0) {
    --g_cfg.part_overflows;
    atomicExch(d_err + 1, 1);
  }
  return hipSuccess;
}
hipError_t launch_baseline_fast(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, void* scratch,
                                int64_t scratch_bytes, int64_t cap_bytes, int variant, int n_cus, hipStream_t s,
                                LaunchStats* st) {
  if (baseline_fast_variant(p, fv, variant, n_cus) != 1)
    return launch_baseline_partitioned(p, fv, out, d_err, scratch, scratch_bytes, cap_bytes, n_cus, s, st);
  finish(p, fv, out, d_err, st, "k_baseline_direct", 1, F_BASELINE_DIRECT);
  return hipSuccess;
}

// ---- joins: the one-to-one probe + SUM / COUNT member; the partitioned and payload probes are not simulated
bool join_sum_eligible(const DevPlan& p, const FragView&) {
  return on(F_JOIN_SUM) && p.desc_type == MI355Q_NON_GROUPED_AGGREGATE && p.join_col >= 0 && p.join_n_keys == 1 &&
         p.join_kind == MI355Q_JOIN_INNER && (p.join_hash_type == 0 || p.join_hash_type == 1) && p.n_quals == 0 &&
         plain_aggs(p, 1);
}
hipError_t launch_join_sum(const DevPlan& p, const FragView& fv, int64_t* out, int, hipStream_t, LaunchStats* st) {
  finish(p, fv, out, nullptr, st, "k_join_sum", 0, F_JOIN_SUM);
  return hipSuccess;
}
bool join_part_supported(const DevPlan&, const FragView&, int) { return false; }
int64_t join_part_scratch_bytes(const DevPlan&, const FragView&, int, int64_t) { return 0; }
hipError_t launch_join_partitioned(const DevPlan&, const FragView&, int64_t*, int32_t*, void*, int64_t, int64_t, int,
                                   hipStream_t, LaunchStats*) { return hipErrorNotSupported; }
bool join_probe_wants(const DevPlan&, const FragView&, int, int*, int*) { return false; }
bool join_probe_supported(const DevPlan&, const FragView&, const JoinPayloadView&, int) { return false; }
int64_t join_probe_scratch_bytes(const DevPlan&, const FragView&, const JoinPayloadView&, int, int64_t) { return 0; }
hipError_t launch_join_probe(const DevPlan&, const FragView&, const JoinPayloadView&, int64_t*, int32_t*, void*, int64_t,
                             int64_t, int, hipStream_t, LaunchStats*) { return hipErrorNotSupported; }
hipError_t launch_join_payload_build(const void*, int, int64_t, const void*, uint32_t*, int64_t*, uint32_t*, void*,
                                     int64_t*, int32_t*, int, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_join_payload_keyed_build(const void*, int, int64_t, const void*, int64_t*, void*, int64_t*, int32_t*,
                                           int, hipStream_t) { return hipErrorNotSupported; }

// ---- multi-device slice merge, ORDER BY: not simulated (the callers report UNSUPPORTED / fall back)
int64_t slice_merge_scratch_bytes(const DevPlan&, int64_t, int64_t, int) { return 0; }
hipError_t launch_slice_merge(const DevPlan&, int64_t*, const int64_t* const*, const int64_t* const*, int, int, int64_t,
                              int64_t, int32_t*, void*, int64_t, int, hipStream_t) { return hipErrorNotSupported; }
int64_t topk_scratch_bytes(int64_t) { return 256; }
int topk_max_k() { return 0; }
hipError_t launch_topk(const DevPlan&, int, int, int64_t, bool, bool, bool, const int64_t*, int64_t, void*, int64_t*,
                       int64_t*, hipStream_t) { return hipErrorNotSupported; }
int64_t sort_scratch_bytes(int64_t) { return 256; }
hipError_t launch_sort(const DevPlan&, int, const SortOrderEntry*, int, const int64_t*, int64_t, int64_t, void*, int64_t*,
                       int64_t*, hipStream_t) { return hipErrorNotSupported; }

// ---- perfect-hash GROUP BY partitioned by entry index (kernels_idx.hip): no stand-in — the family is simply absent from
// this build (the packed route keeps those shapes); the real kernels run in the HOSTSIM_REAL_FAST build
bool idx_part_eligible(const DevPlan&, const FragView&, int) { return false; }
int64_t idx_part_scratch_bytes(const DevPlan&, const FragView&, int, int64_t) { return 0; }
hipError_t launch_idx_partitioned(const DevPlan&, const FragView&, int64_t*, int32_t*, void*, int64_t, int64_t, int, hipStream_t,
                                  LaunchStats*) {
  return hipErrorInvalidValue;
}

}  // namespace mq
#endif  // HOSTSIM_REAL_FAST
