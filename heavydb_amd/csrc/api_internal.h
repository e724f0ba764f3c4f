// api_internal.h — what the translation units of the C-ABI share (api.cpp, api_projection.cpp): the opaque handle
// types, the per-device workspace, and the small host-side helpers.  Internal to libmi355q.
#pragma once

#include <hip/hip_runtime_api.h>

#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"
#include "plan.h"

struct mi355q_join_table {
  int device_id = 0;
  int hash_type = 0;  // 0 perfect 1:1, 1 keyed 1:1, 2 perfect 1:N, 3 keyed 1:N (mi355q.h)
  int key_type = MI355Q_INT64;
  int n_keys = 1, width = 8;  // key components / component width of keyed tables
  int64_t entry_count = 0;
  int64_t min_key = 0, max_key = 0;
  void* buf = nullptr;
  int64_t bytes = 0;
  void* bitmap = nullptr;  // perfect tables: presence bitmap (1 bit per slot), for probes that
                           // only need to know WHETHER a key matches (no inner column read)
  float build_ms = 0.f;
  // OneToOne perfect table in which EVERY slot holds a row id: the inner key column is NOT NULL, the build met no duplicate
  // and rows == max - min + 1.  hash_join_idx (GroupByRuntime.cpp:287-297) then answers >= 0 exactly for min <= key <= max.
  bool dense = false;
  // perfect tables: per-key aggregated payload for the payload probe (kernels_part.hip), built on
  // first use for one inner column and kept with the table (the inner table does not change under
  // a join table): rows per key, sum of the inner column over them, non-NULL values among them
  std::mutex pay_mu;
  uint32_t* pay_cnt = nullptr;
  int64_t* pay_wsum = nullptr;
  uint32_t* pay_wnn = nullptr;
  void* pay16 = nullptr;         // the same as 16-byte entries (L2 mode of the probe)
  int64_t* pay8 = nullptr;       // one-to-one tables, L2 mode: the inner value per key slot, INT64_MIN = absent
  int64_t* pay_kkeys = nullptr;  // keyed tables: the key of every slot (pay16 / pay8 are then per slot)
  const void* pay16_col = nullptr;
  bool pay16_built = false, pay_col_built = false;
  int pay16_has_nulls = 0;
  const void* pay_col = nullptr;
  int pay_has_nulls = 0;
  float pay_build_ms = 0.f;
  int64_t pay_version = 0, pay16_version = 0;  // mi355q_inputs.inner_version the payloads were built for
  // a payload the probe plan then refused (built, dropped): not built again for the same column and step shape
  bool pay_refused = false;
  const void* pay_refused_col = nullptr;
  int64_t pay_refused_rows = 0;
};

struct mi355q_result {
  mi355q_qmd qmd{};
  mq::DevPlan dplan{};  // layout + targets for reduce / iteration kernels
  int device_id = 0;
  int64_t* buf = nullptr;
  int64_t bytes = 0;
  bool owns_buf = false;
  int64_t total_matched = -1;  // Projection results: rows that passed the quals; -1 = not known to the host (a wrapped buffer)
  int64_t live_rows = -1;      // Projection results put together by mi355q_result_append: the rows at the front of the buffer
};

namespace mq {
namespace api {

// mi355q_explain: the route of a step, written down while execute_impl plans it in RESERVE mode (nothing is launched,
// nothing is allocated: t_plan_only).  Every derived route (projection, row-wise / 8-byte twins, packed keys, one run
// per value column) notes itself and plans its derived step the same way.
extern thread_local std::string* t_route;
extern thread_local bool t_plan_only;
void route_note(const char* what);


#define HIP_TRY(expr)                                     \
  do {                                                    \
    hipError_t _e = (expr);                               \
    if (_e != hipSuccess) {                               \
      mq::api::last_hip_error = _e;                                \
      return _e == hipErrorOutOfMemory ? MI355Q_ERR_OUT_OF_GPU_MEM : MI355Q_ERR_HIP; \
    }                                                     \
  } while (0)

extern thread_local hipError_t last_hip_error;

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
    // hipGetLastError() is per thread and sticky: another runtime user in this process (torch)
    // may have left an unrelated error behind, which the launch checks would then report
    (void)hipGetLastError();
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int cu_count_of(int dev);

// Per-device workspace kept between calls: the partition scratch (tens of GB for the headline
// workload — hipMalloc/hipFree of that size costs up to a second per call), the fragment
// pointer tables and the timing events.  Calls on one device are serialised by `mu`, like
// the reference's per-device gpu_exec_mutex_ (ExecutionKernel.cpp:216-220).
struct DeviceCtx {
  std::recursive_mutex mu;  // the packed multi-column path re-enters mi355q_execute
  void* aux = nullptr;      // packed key column + temporary tables of that path
  int64_t aux_bytes = 0;
  void* wide = nullptr;     // 8-byte-slot table of a step whose result layout has 4-byte slots
  int64_t wide_bytes = 0;
  void* proj = nullptr;     // dense temporary columns of projected expressions (one pass of fragments)
  int64_t proj_bytes = 0;
  void* gather = nullptr;   // dense temporary columns of a grouped join's inner side (execute_join_gather; may nest inside proj's step)
  int64_t gather_bytes = 0;
  void* lattice = nullptr;  // dense INT32 key columns of a lattice-keyed step (execute_affine_twin; may nest inside both)
  int64_t lattice_bytes = 0;
  void* bf_table = nullptr; // a compiled filter in device memory (boolfilter.h BoolFilter: atoms + truth table)
  void* maskws = nullptr;   // the row mask of a compiled filter with program atoms (execute_masked: one pass of fragments)
  int64_t maskws_bytes = 0;
  void* projws = nullptr;   // Projection family: lowered expressions, ticket / total counters, tile table, tile descriptors
  int64_t projws_bytes = 0;
  void* scratch = nullptr;
  int64_t scratch_bytes = 0;
  void* meta = nullptr;
  size_t meta_bytes = 0;
  // pinned host mirror of `meta` (column table, row counts, zeroed error words go to the device as ONE copy that does not
  // stage through a driver buffer) + 64 bytes the error words and the spill counter come back into
  char* h_meta = nullptr;
  // what `meta` holds on the device: the table bytes of the last upload, and whether its error words are still the zeros
  // that upload laid down (a prepared step over resident columns sends the same table every time: the copy — a DMA command
  // ahead of the kernel, ~8 us of a 60 us scan — is then skipped)
  std::vector<char> meta_shadow;
  bool meta_err_clean = false;
  int32_t* h_ret_dev = nullptr;  // the device's address of the 64 return bytes behind h_meta (k_words_to_host writes them)
  std::vector<hipEvent_t> events;
  hipStream_t stream = nullptr;  // library-owned launch stream (when the caller passes none)
  struct mi355q_pending* inflight = nullptr;  // a step enqueued by mi355q_execute_async and not yet waited for
};
DeviceCtx& ctx_of(int dev);


// Small pinned-free device scratch for the error word / counters, one per call.
struct DevWord {
  void* p = nullptr;
  ~DevWord() {
    if (p) (void)hipFree(p);
  }
};

// a result handle over `qmd` (allocates the buffer unless one is given; no initialisation)
int32_t result_create_impl(const mi355q_qmd* qmd, int32_t device_id, void* device_buffer, mi355q_result** out);
// column bytes a plan must read (mi355q_exec_report.algorithmic_bytes)
int64_t algorithmic_bytes(const mi355q_plan& p, const mi355q_inputs& in);
// the Projection family's step (api_projection.cpp); reserved != null: plan only
int32_t execute_projection(const mi355q_plan* plan, const mi355q_inputs* in, const mi355q_exec_options& o, mi355q_result** out,
                           mi355q_exec_report* report, int64_t* reserved);
// result accessors of a Projection buffer
// ---- shared between api.cpp (the step executor), api_result.cpp (result objects, shards) and api_join.cpp
RowInit make_row_init(const mi355q_qmd& q);
ColLayout col_layout_of(const mi355q_qmd& q);
// the device operations that walk rows run on a row-wise twin of a columnar buffer (same entries, same values)
struct RowTwin {
  mi355q_result* tw = nullptr;
  ~RowTwin() {
    if (tw) mi355q_result_free(tw);
  }
};
int32_t make_row_twin(const mi355q_result* r, hipStream_t s, RowTwin* out);
int32_t store_row_twin(const RowTwin& t, mi355q_result* r, hipStream_t s);
int32_t run_reduce(mi355q_result* dst, const int64_t* rows, int64_t n_rows, void* stream);
int32_t attach_join(const mi355q_plan& p, const mi355q_inputs* in, DevPlan* d);  // (api.cpp) the join table and the inner columns
int64_t projection_row_count(const mi355q_result* r);
int32_t projection_append(mi355q_result* this_rs, const mi355q_result* that_rs, hipStream_t s);
int32_t projection_fetch_rows(const mi355q_result* r, int64_t max_rows, int64_t* ival, double* dval, int8_t* is_null, int64_t* n_rows);

}  // namespace api
}  // namespace mq
