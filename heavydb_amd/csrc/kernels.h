// kernels.h — host-callable launchers of the HIP kernel family (internal to libmi355q).
#pragma once

#include <hip/hip_runtime_api.h>

#include "dev_common.h"

namespace mq {

// Partition scratch of the partitioned families when the caller sets no cap: large enough for the
// record-index limit of one chunk (2^32 records x 16 B) plus its spill list, so that 10 B rows are
// three chunks; mi355q_execute lowers it to what the device can actually spare (api.cpp).
constexpr int64_t kDefaultScratchCap = (int64_t)76 << 30;


struct RowInit {           // one output row image: key quads then slot init values
  int64_t quad[MI355Q_MAX_GROUP_COLS + MI355Q_MAX_SLOTS];
  int32_t row_quad;
};

struct LaunchStats {
  const char* kernel_name = "";
  int n_launches = 0;
  int variant = 0;
  int64_t spilled_rows = 0;
  // events bracketing the dominant kernel(s) only (subset of the whole call)
  hipEvent_t k_start = nullptr, k_stop = nullptr;
  // multi-launch families: pairs (start, stop) taken from this pool, one per dominant launch
  hipEvent_t* ev_pool = nullptr;
  int n_ev = 0;
  int n_events_used = 0;
  uint32_t* spill_counter32 = nullptr;  // device word (read back by the caller)
};

// testing / tuning knobs of one mi355q_execute call (mi355q_exec_options; thread-local while the call runs —
// derived plans re-enter mi355q_execute with the same options).  Nothing reads the environment.
struct TuneKnobs {
  int blocks_per_cu = 0;
  int probe_keyed_passes = 0;
  int64_t pass_rows = 0;
  uint32_t flags = 0;  // MI355Q_OPT_*
  int overlap_cus = 0;  // partitioned GROUP BY: CUs of phase 1 while phase 2 of the previous chunk runs on the rest
};
// n 32-bit words of device memory into host memory the device can address (hipHostGetDevicePointer), one wave
hipError_t launch_words_to_host(const int32_t* d_src, int32_t* h_dst_dev, int n, hipStream_t s);
const TuneKnobs& tune_knobs();
void set_tune_knobs(const TuneKnobs& k);
// the compiled filter of the step being planned / launched on this thread (boolfilter.h; table already in device memory),
// or null: the step's filter is its plan's quals
struct BoolFilter;
const BoolFilter* step_bool_filter();      // the host copy (columns, atom counts: what eligibility looks at)
const BoolFilter* step_bool_filter_dev();  // the same filter in device memory (what a kernel is handed)
void set_step_bool_filter(const BoolFilter* bf, const BoolFilter* dev);

// ---- generic family (kernels_generic.hip)
hipError_t launch_init_buffer(int64_t* buf, int64_t entry_count, const RowInit& init,
                              hipStream_t s);
// perfect-hash tables that fit LDS are aggregated in a per-workgroup LDS copy and folded into
// `out` with the reduce rule; everything else updates `out` directly
hipError_t launch_generic(const DevPlan& p, int idx_target_as_key, const RowInit& init,
                          const int8_t* const* d_cols, const int64_t* d_num_rows, int n_frags,
                          int64_t max_frag_rows, int64_t* out, int32_t* d_err, int n_cus,
                          hipStream_t s);
hipError_t launch_reduce(const DevPlan& p, int idx_target_as_key, int64_t* this_buf,
                         const int64_t* that_rows, int64_t that_entries, int32_t* d_err,
                         hipStream_t s);
hipError_t launch_count_nonempty(const DevPlan& p, int idx_target_as_key, const int64_t* buf,
                                 unsigned long long* d_count, hipStream_t s);
// multi-device merge by home-slot slices (kernels_generic.hip)
hipError_t launch_shard_pads(const DevPlan& p, const int64_t* buf, int world, int pad_rows, int64_t* out_pads,
                             int32_t* d_ok, hipStream_t s);
hipError_t launch_reduce_range(const DevPlan& p, int idx_target_as_key, int64_t* this_buf, const int64_t* that_rows,
                               int64_t that_entries, int64_t home_lo, int64_t home_hi, int32_t* d_err, hipStream_t s);
// LDS fold of the slice exchange (kernels_part.hip): rows [lo, hi) of n_src tables + their pads into rows
// [lo, hi) of `out`; 0 scratch bytes = the layout is not one the partitioned family takes
int64_t slice_merge_scratch_bytes(const DevPlan& p, int64_t lo, int64_t hi, int n_cus);
hipError_t launch_slice_merge(const DevPlan& p, int64_t* out, const int64_t* const* src, const int64_t* const* pads,
                              int n_src, int pad_rows, int64_t lo, int64_t hi, int32_t* d_err, void* scratch,
                              int64_t scratch_bytes, int n_cus, hipStream_t s);

hipError_t launch_shard_partition(const DevPlan& p, int idx_target_as_key, const int64_t* buf,
                                  int n_parts, int64_t* out_rows, int64_t* d_part_counts,
                                  int64_t* d_cursors /* n_parts scratch */, hipStream_t s);
hipError_t launch_join_fill_perfect(const int8_t* keys, int type, int nullable, int64_t n,
                                    int64_t min_key, int64_t max_key, int32_t* buf,
                                    int32_t* d_err, hipStream_t s);
// bit i = (table[i] >= 0): the semi-join view of a perfect table, 32x smaller than the table
hipError_t launch_join_presence_bitmap(const int32_t* table, int64_t entries, uint32_t* bitmap,
                                       hipStream_t s);
// ---- multi-column keys through the single-key fast families: the group columns of a row are
// packed into one int64 (per column: (key - min), or the last code for NULL, at its bit offset),
// the step runs on the packed column, and the finished table is re-emitted with the real key
// components (kernels_generic.hip: k_pack_keys / k_unpack_emit)
struct PackSpec {
  int32_t n;
  int32_t cols[MI355Q_MAX_GROUP_COLS], types[MI355Q_MAX_GROUP_COLS];  // column index, type code
  int32_t nullable[MI355Q_MAX_GROUP_COLS], shift[MI355Q_MAX_GROUP_COLS];
  int64_t min[MI355Q_MAX_GROUP_COLS];
  uint64_t card[MI355Q_MAX_GROUP_COLS];  // codes 0 .. card-1 (card-1 = NULL for nullable columns)
  uint64_t mask[MI355Q_MAX_GROUP_COLS];
  // perfect-hash layouts (mode 1): the code is the ENTRY INDEX sum_i (key_i - min_i) * mul_i with
  // NULL keys translated to null_key first, exactly as the row function computes it
  int32_t mode;  // 0 bit-packed (baseline), 1 entry index into a baseline temp table, 2 entry index
                 // into an index-aligned (perfect) temp table
  int32_t raw_f32;  // mode 0, ONE FLOAT key: the "packed" key is the bit pattern of the double the key widens to — what
                    // the table stores for a FLOAT key anyway (castToTypeIn(group_key, 64), IRCodegen.cpp:1505-1507) —
                    // so the partitioned family, which reads 8-byte keys, takes the step
  int32_t tmp_idx_target;  // mode 2: emptiness of a keyless temp row = slot tmp_idx_target == tmp_init
  int64_t tmp_init;
  int32_t translate[MI355Q_MAX_GROUP_COLS];
  int64_t mul[MI355Q_MAX_GROUP_COLS], null_key[MI355Q_MAX_GROUP_COLS];
  int64_t bucket[MI355Q_MAX_GROUP_COLS];  // perfect-hash modes: 0, or the bucket the index divides by (DATE in days)
  // final slot <- temp slot (>= 0) or <- original value of key component -(1 + k)
  int32_t slot_src[MI355Q_MAX_SLOTS];
};
// packed_cols: device array [n_frags] of output pointers
hipError_t launch_pack_keys(const PackSpec& ps, const int8_t* const* d_cols, const int64_t* d_num_rows,
                            int n_frags, int n_cols, int64_t max_frag_rows, int64_t* const* packed_cols,
                            int32_t* d_err, int n_cus, hipStream_t s);
// copies slots src[i] of every live row of `sub` (layout ps) to slots dst[i] of the same group's row in `fin`
// (layout pf, same group columns): see k_zip_targets
// a baseline step over integer keys with ranges, run as a perfect hash over the product of the ranges: entries re-keyed into the baseline table
hipError_t launch_perfect_twin_emit(const DevPlan& pf, const DevPlan& ps, int idx_key_s, int n_keys, const int32_t* translate,
                                    const int32_t* key_type, const int64_t* key_min, const int64_t* key_card,
                                    const int64_t* null_key, const int64_t* sub, int64_t* fin, int32_t* d_err, hipStream_t s);
// grouped joins over a one-to-one table: the inner columns the aggregates read + a "matched" column as dense outer columns
hipError_t launch_join_gather(const DevPlan& p, int n_inner, const int32_t* inner_col, const int32_t* dst_col, const int32_t* width,
                              const int64_t* null_pat, int flag_col, int nc2, const int8_t* const* d_cols, const int64_t* d_num_rows,
                              int n_frags, int64_t max_frag_rows, int n_cus, hipStream_t s);
// baseline keys on a lattice (key = min + stride x i): the stride of a column's first fragment, i as dense INT32 columns (verified row
// by row: *d_flag raised for a key off the lattice), the twin's entries re-keyed into the baseline table
// (scratch: 64 * 256 words; out256: 256 partial strides the host folds)
hipError_t launch_key_gcd(const void* col, int width, int64_t n, int64_t kmin, int nullable, unsigned long long* scratch,
                          unsigned long long* out256, hipStream_t s);
hipError_t launch_affine_keys(int n, const int32_t* src_col, const int32_t* dst_col, const int32_t* width, const int32_t* nullable,
                              const int64_t* kmin, const int64_t* stride, const int64_t* card, int nc2, const int8_t* const* d_cols,
                              const int64_t* d_num_rows, int n_frags, int64_t max_frag_rows, int32_t* d_flag, int n_cus, hipStream_t s);
hipError_t launch_affine_twin_emit(const DevPlan& pf, const DevPlan& ps, int idx_key_s, int n_keys, const int32_t* translate,
                                   const int32_t* key_type, const int64_t* twin_min, const int64_t* twin_card, const int64_t* twin_null,
                                   const int64_t* base, const int64_t* stride, const int64_t* sub, int64_t* fin, int32_t* d_err,
                                   hipStream_t s);
// GROUP BY CAST(int column AS DOUBLE | FLOAT): entries of the integer-keyed perfect table re-keyed and merged into the baseline table
hipError_t launch_cast_key_emit(const DevPlan& pf, const DevPlan& ps, int idx_key_s, int cast_to_float, int translate,
                                int64_t key_min, int64_t null_key, const int64_t* sub, int64_t* fin, int32_t* d_err,
                                hipStream_t s);
// (kind / cnt_src / lit: per copy, the `aggregate of column + literal` fix-up of ZipMap; null = plain copies)
hipError_t launch_zip_targets(const DevPlan& pf, const DevPlan& ps, int idx_key_s, const int64_t* sub, int64_t* fin,
                              const int32_t* src, const int32_t* dst, int n, int32_t* d_err, hipStream_t s,
                              const int32_t* kind = nullptr, const int32_t* cnt_src = nullptr, const int64_t* lit = nullptr,
                              bool into_empty_table = false);
// projected expressions: d_cols is the pass's extended fragment table [frag][xs.n_cols + xs.n]; the last xs.n
// pointers of every fragment are the output columns (dense, of each expression's result type).  p = the
// device plan of the LOWERED plan (quals, join): it decides whether a row's overflow counts.
// d_xs_area: sizeof(DevExprSet) bytes of device memory the lowered programs are uploaded to (they are too large for the
// kernel-argument segment); the caller keeps `xs` alive until it has synchronised the stream
hipError_t launch_project(const DevExprSet& xs, DevExprSet* d_xs_area, const DevPlan& p, uint32_t qual_expr_mask,
                          const int8_t* const* d_cols, const int64_t* d_num_rows, int n_frags, int64_t max_frag_rows, int32_t* d_err,
                          int n_cus, hipStream_t s, bool simple = false);
// every expression is CAST(plain INT / BIGINT column AS DOUBLE | FLOAT) or column + - * literal (k_project_simple)
bool project_simple_shapes(const DevExprSet& xs);
// tmp: table of the packed single-key step (rows = packed key + slot_count slots); out: the
// initialised final table described by p
hipError_t launch_unpack_emit(const PackSpec& ps, const DevPlan& p, const int64_t* tmp, int64_t tmp_entries,
                              int tmp_quad, int tmp_key_quad, int64_t* out, int32_t* d_err, hipStream_t s);

// dense columns of the non-empty entries (ColumnarResults): flags/offsets are int32[entries] scratch,
// tile_scratch holds (entries / 2048 + 2) int64; cols = device array of n_targets column pointers
struct ColumnarSpec {
  int64_t null_pat[MI355Q_MAX_TARGETS];
  int32_t is_fp[MI355Q_MAX_TARGETS];
};
hipError_t launch_to_columns(const DevPlan& p, int idx_target_as_key, const ColumnarSpec& cs, const int64_t* buf,
                             int32_t* flags, int32_t* offsets, int64_t* tile_scratch, int64_t* const* cols,
                             hipStream_t s);

// compact COUNT(*)-only layouts: the finished 8-byte-slot table -> its 4-byte-slot image
hipError_t launch_narrow_slots(const int64_t* wide, int wide_quad, int key_quad, int slot_count,
                               int narrow_quad, int64_t entries, int64_t* out, hipStream_t s);

// row-wise <-> columnar result buffers (ColLayout: dev_common.h); init = initColumnarGroups
hipError_t launch_rows_to_columns(const ColLayout& L, const int64_t* rows, void* cols, hipStream_t s);
hipError_t launch_columns_to_rows(const ColLayout& L, const void* cols, int64_t* rows, hipStream_t s);
hipError_t launch_init_columns(const ColLayout& L, const RowInit& init, void* cols, hipStream_t s);

// inner key columns of a join table build
struct JoinKeyCols {
  const int8_t* col[MI355Q_MAX_GROUP_COLS];
  int32_t type[MI355Q_MAX_GROUP_COLS], nullable[MI355Q_MAX_GROUP_COLS];
  int32_t n, width;  // components and their width in the table (4 / 8)
};
hipError_t launch_join_init_keyed(void* tab, int64_t entries, int n_keys, int stride, int width,
                                  hipStream_t s);
hipError_t launch_join_fill_keyed(const JoinKeyCols& kc, int64_t n, void* tab, int64_t entries, int stride,
                                  bool with_payload, int32_t* d_err, hipStream_t s);
hipError_t launch_join_one_to_many(const JoinKeyCols& kc, int64_t n, int hash_type, const void* tab,
                                   int64_t entries, int64_t min_key, int64_t max_key, int32_t* offsets,
                                   int32_t* counts, int32_t* payloads, int64_t* tile_scratch,
                                   int32_t* d_err, hipStream_t s);
hipError_t launch_generate(void* dst, int64_t n_rows, int64_t row_offset, int kind,
                           uint64_t seed, int64_t a, int64_t b, int64_t c, double a_f,
                           int null_every, hipStream_t s);

// ---- fast families (kernels_fast.hip); each returns hipErrorNotSupported-free bool
// eligibility via the *_eligible functions, decided at plan time.
struct FragView {
  const int8_t* const* d_cols;  // device array [n_frags * n_cols]
  const int64_t* d_num_rows;    // device array [n_frags]
  const void* const* h_cols;    // host copy of the pointer table (alignment checks)
  const int64_t* h_num_rows;
  int n_frags;
  int n_cols;
  int64_t total_rows;
  int64_t max_frag_rows;
};

// non-grouped aggregates over up to 8 plain int32 / int64 / double columns with up to 4 integer range quals
bool scan_agg_eligible(const DevPlan& p, const FragView& fv);
hipError_t launch_scan_agg(const DevPlan& p, const FragView& fv, int64_t* out, int n_cus, hipStream_t s,
                           LaunchStats* st);
// GROUP BY with few groups, whole table replicated in every workgroup's LDS (kernels_lds.hip): perfect-hash layouts
// of 1-3 integer key columns with <= 65536 entries, or a baseline layout over one 8-byte / int32 key with at most a
// few thousand groups (more: d_err[1] is set and the caller re-runs the step with another family); 1-3 value
// columns, any aggregate kinds, NULL-aware or not, up to 4 integer range quals
bool lds_groupby_eligible(const DevPlan& p, const FragView& fv, int n_cus);
// ---- perfect-hash GROUP BY on a table too large for LDS: partition by entry index, narrow records (kernels_idx.hip)
bool idx_part_eligible(const DevPlan& p, const FragView& fv, int n_cus);
int64_t idx_part_scratch_bytes(const DevPlan& p, const FragView& fv, int n_cus, int64_t cap_bytes);
hipError_t launch_idx_partitioned(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, void* scratch,
                                  int64_t scratch_bytes, int64_t cap_bytes, int n_cus, hipStream_t s, LaunchStats* st);
bool lds_groupby_typed_eligible(const DevPlan& p, const FragView& fv, int n_cus);  // ... and a TYPED member (roles compiled in) would run it
hipError_t launch_lds_groupby(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, int n_cus,
                              hipStream_t s, LaunchStats* st);
bool scan_count_eligible(const DevPlan& p, const FragView& fv);
hipError_t launch_scan_count(const DevPlan& p, const FragView& fv, int64_t* out, int n_cus,
                             hipStream_t s, LaunchStats* st);

bool perfect_lds_eligible(const DevPlan& p, const FragView& fv);
hipError_t launch_perfect_lds(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err,
                              int n_cus, hipStream_t s, LaunchStats* st);

bool baseline_fast_eligible(const DevPlan& p, const FragView& fv);
// scratch: device workspace of scratch_bytes; variant selects direct-atomic (1) or
// partitioned (2); 0 = plan-time choice.
hipError_t launch_baseline_fast(const DevPlan& p, const FragView& fv, int64_t* out,
                                int32_t* d_err, void* scratch, int64_t scratch_bytes,
                                int64_t cap_bytes, int variant, int n_cus, hipStream_t s,
                                LaunchStats* st);
int64_t baseline_fast_scratch_bytes(const DevPlan& p, const FragView& fv, int variant,
                                    int64_t cap_bytes, int n_cus);
// variant resolution: 1 direct atomics, 2 partition-then-aggregate
int baseline_fast_variant(const DevPlan& p, const FragView& fv, int requested, int n_cus);

// kernels_part.hip
bool part_supported(const DevPlan& p, const FragView& fv, int n_cus);
int64_t part_scratch_bytes(const DevPlan& p, const FragView& fv, int n_cus, int64_t cap_bytes);
// d_err[0]: reference error code; d_err[1]: set when the spill list overflowed (the caller
// re-runs the step with the direct kernel)
hipError_t launch_baseline_partitioned(const DevPlan& p, const FragView& fv, int64_t* out,
                                       int32_t* d_err, void* scratch, int64_t scratch_bytes,
                                       int64_t cap_bytes, int n_cus, hipStream_t s,
                                       LaunchStats* st);

// radix join probe (kernels_part.hip): non-grouped INNER semi-join + COUNT(*) / SUM(fact col) with
// the fact rows partitioned by key range so that every partition probes a bitmap slice in LDS
bool join_part_supported(const DevPlan& p, const FragView& fv, int n_cus);
int64_t join_part_scratch_bytes(const DevPlan& p, const FragView& fv, int n_cus, int64_t cap_bytes);
hipError_t launch_join_partitioned(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err,
                                   void* scratch, int64_t scratch_bytes, int64_t cap_bytes, int n_cus,
                                   hipStream_t s, LaunchStats* st);

// ---- ORDER BY one target LIMIT k on the device (kernels_sort.hip)
int64_t topk_scratch_bytes(int64_t entry_count);
int topk_max_k();
hipError_t launch_topk(const DevPlan& p, int idx_target_as_key, int target, int64_t null_pattern,
                       bool fp_result, bool desc, bool nulls_first, const int64_t* buf, int64_t k,
                       void* scratch, int64_t* out_rows, int64_t* d_n_out, hipStream_t s);

// full ORDER BY (kernels_sort.hip): one entry per order column, most significant first
struct SortOrderEntry {
  int32_t target;
  int32_t desc, nulls_first, fp_result;
  int64_t null_pattern;
};
int64_t sort_scratch_bytes(int64_t entry_count);
hipError_t launch_sort(const DevPlan& p, int idx_target_as_key, const SortOrderEntry* order, int n_order,
                       const int64_t* buf, int64_t offset, int64_t limit, void* scratch, int64_t* out_rows,
                       int64_t* d_n_out, hipStream_t s);

// payload probe (kernels_part.hip): joins whose non-grouped targets read the inner side / one-to-many
// tables / LEFT joins, through per-key aggregated payload arrays of a perfect-hash table kept in LDS
struct JoinPayloadView {
  const uint32_t* cnt_k;   // [entries] rows per key
  const int64_t* wsum_k;   // [entries] sum of the inner column over the key's rows (NULLs skipped)
  const uint32_t* wnn_k;   // [entries] non-NULL values among them
  const void* pay16;       // [entries] {wsum, cnt, wnn} as one 16-byte entry per key (L2 mode), or null
  const int64_t* pay8;     // [entries] one-to-one tables: the inner value, INT64_MIN where absent (L2 mode), or null
  const int64_t* kkeys;    // keyed tables: [entries] the key of every slot (EMPTY_KEY_64 where free); pay16 / pay8
                           // are then indexed by slot
  const void* inner_col;   // the inner column wsum / wnn were built for (nullptr: counts only)
  int64_t entries;
  int32_t has_nulls;       // some matching inner value is NULL: wnn != cnt somewhere
};
hipError_t launch_join_payload_build(const void* table, int hash_type, int64_t entries, const void* inner_col,
                                     uint32_t* cnt_k, int64_t* wsum_k, uint32_t* wnn_k, void* pay16, int64_t* pay8,
                                     int32_t* d_flags, int n_cus, hipStream_t s);
hipError_t launch_join_payload_keyed_build(const void* table, int hash_type, int64_t entries, const void* inner_col,
                                           int64_t* kkeys, void* pay16, int64_t* pay8, int32_t* d_flags, int n_cus,
                                           hipStream_t s);
// l2_mode: 0 = LDS slices of a perfect table, 1 = L2 slices of a perfect table, 2 = keyed table (L2 slices by slot)
bool join_probe_wants(const DevPlan& p, const FragView& fv, int n_cus, int* inner_col, int* l2_mode);
bool join_probe_supported(const DevPlan& p, const FragView& fv, const JoinPayloadView& pay, int n_cus);
int64_t join_probe_scratch_bytes(const DevPlan& p, const FragView& fv, const JoinPayloadView& pay, int n_cus,
                                 int64_t cap_bytes);
hipError_t launch_join_probe(const DevPlan& p, const FragView& fv, const JoinPayloadView& pay, int64_t* out,
                             int32_t* d_err, void* scratch, int64_t scratch_bytes, int64_t cap_bytes, int n_cus,
                             hipStream_t s, LaunchStats* st);

// ---- PROJECTION family (kernels_proj.hip): order-preserving stream compaction of the rows that pass the quals
enum ProjKind : int32_t { PROJ_INT = 0, PROJ_F64 = 1, PROJ_F32 = 2, PROJ_F32_TO_F64 = 3 };
constexpr int32_t kProjInnerCol = 64;  // ProjTarget::col from here on: column (col - 64) of the join's inner table
struct ProjTarget {
  int32_t col;     // source: a physical column (< n_phys_cols) or expression col - n_phys_cols
  int32_t code;    // type code of the physical column; the result type of an expression
  int32_t kind;    // ProjKind: how the value is stored
  int32_t width;   // bytes of the target's slot in the buffer (8 row-wise; the logical width in a columnar buffer)
  int64_t col_off; // columnar: byte offset of the slot column in the buffer
};
struct ProjSpec {
  int32_t n_targets, columnar;
  int32_t row_quad;       // row-wise: 1 + n_targets
  int32_t n_phys_cols;    // physical columns; indices from here on name the plan's expressions
  int32_t n_cols_table;   // pointers per fragment in the column table
  int32_t x_info;         // expressions (uploaded with their handlers, expr.h xh_label_programs): deepest stack | count << 8
  int64_t entry_count;
  ProjTarget t[MI355Q_MAX_TARGETS];
};
// A Projection target that is `[CAST](column) <op> literal` (or a cast alone) over a plain INT32 / INT64 / DOUBLE column,
// <op> one of + - * : evaluated by the FAST member on the quad it has loaded — one ex_cast / ex_arith (expr.h) per row
// behind wave-uniform branches — instead of sending the whole step to the general member's interpreter (round 6; the
// reference compiles target expressions into the row function, NativeCodegen.cpp:3455, ArithmeticIR.cpp:39-431)
struct ProjForm {
  int32_t on;                   // 1: this target is a form
  int32_t src_col, src_code;    // the physical column it reads
  int32_t cast_to, cast_flags;  // 0: no cast; else ex_cast from the column's type to this one
  int32_t op, type, flags;      // 0: no operation; else MI355Q_EX_ADD / _SUB / _MUL at `type` (EXF_* of the node)
  int32_t lit_first, can_raise; // literal <op> value; the form can raise error 7
  int64_t lit;                  // the literal's pattern (ex_lit)
};
struct ProjForms {
  int32_t ok, pad_;
  ProjForm f[MI355Q_MAX_TARGETS];
};
// the forms of a Projection step's expression targets; ok = 0: some expression is not a form (or a qual reads one)
void projection_forms(const DevExprSet& xs, const ProjSpec& ps, uint32_t qual_expr_mask, ProjForms* out);
int64_t projection_tile_rows();
int64_t projection_scratch_bytes(int n_frags, const int64_t* h_num_rows);
// p: quals of the (lowered) plan; d_xs: the lowered expressions in DEVICE memory (or null); *d_total: device word that
// holds the number of matching rows once the stream has drained
hipError_t launch_projection(const DevPlan& p, const ProjSpec& ps, const DevExprSet* d_xs, uint32_t qual_expr_mask,
                             const FragView& fv, void* scratch, void* out, int32_t* d_err, unsigned long long** d_total,
                             int n_cus, hipStream_t s, LaunchStats* st, const ProjForms* forms = nullptr);
hipError_t launch_projection_count_live(const int64_t* keys, int64_t stride_quads, int64_t entries, unsigned long long* d_count,
                                        hipStream_t s);

// the row mask of a compiled filter with program atoms (kernels_filter.hip): one byte per row, 1 = the row passes
int64_t filter_mask_chunk_bytes(int64_t n_rows);  // bytes of one fragment's mask chunk (16-byte multiple)
bool filter_mask_eligible(const BoolFilter& bf, const FragView& fv);
hipError_t launch_filter_mask(const BoolFilter& bf, const BoolFilter* d_bf, const FragView& fv, int8_t* const* d_mask, int32_t* d_err,
                              int n_cus, hipStream_t s);

bool join_sum_eligible(const DevPlan& p, const FragView& fv);
hipError_t launch_join_sum(const DevPlan& p, const FragView& fv, int64_t* out, int n_cus,
                           hipStream_t s, LaunchStats* st);

}  // namespace mq
