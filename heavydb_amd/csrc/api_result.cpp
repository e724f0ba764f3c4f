// api_result.cpp — the result objects of the C-ABI (include/mi355q.h): creation / wrapping of ResultSetStorage-layout
// buffers, reduce and append, iteration, columnar export, Arrow export, top-k and sort, and the shard operations of the
// multi-device merge (heavydb_amd/multi_gpu.py).  Split out of api.cpp in round 5; the step executor stays there.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "api_internal.h"

using namespace mq;
using namespace mq::api;

RowInit mq::api::make_row_init(const mi355q_qmd& q) {
  RowInit r{};
  r.row_quad = q.row_size / 8;
  row_init_image(q, r.quad);
  return r;
}

extern "C" {

// ---- columnar results (output_columnar_): the device operations that walk rows run on a
// row-wise twin of the buffer (same entries, same values; rowfunc.h entry_to_columns)
extern "C++" {
ColLayout mq::api::col_layout_of(const mi355q_qmd& q) {
  ColLayout L{};
  L.entry_count = q.entry_count;
  L.slot_col_bytes = ((int64_t)q.slot_width * q.entry_count + 7) & ~(int64_t)7;
  L.key_quads = q.key_bytes / 8;
  L.slot_count = q.slot_count;
  L.slot_width = q.slot_width;
  L.row_quad = q.row_size / 8;
  return L;
}
}  // extern "C++"
extern "C++" {
int32_t mq::api::make_row_twin(const mi355q_result* r, hipStream_t s, RowTwin* out) {
  mi355q_qmd rq = r->qmd;
  rq.output_columnar = 0;
  if (int32_t e = result_create_impl(&rq, r->device_id, nullptr, &out->tw)) return e;
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  HIP_TRY(launch_columns_to_rows(col_layout_of(r->qmd), r->buf, out->tw->buf, s));
  HIP_TRY(hipStreamSynchronize(s));
  return MI355Q_OK;
}
}  // extern "C++"
extern "C++" {
int32_t mq::api::store_row_twin(const RowTwin& t, mi355q_result* r, hipStream_t s) {
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  HIP_TRY(launch_rows_to_columns(col_layout_of(r->qmd), t.tw->buf, r->buf, s));
  HIP_TRY(hipStreamSynchronize(s));
  return MI355Q_OK;
}
}  // extern "C++"

int32_t mi355q_result_create(const mi355q_qmd* qmd, int32_t device_id, void* device_buffer,
                             mi355q_result** out) {
  if (int32_t e = result_create_impl(qmd, device_id, device_buffer, out)) return e;
  DeviceGuard g(device_id);
  hipError_t he = qmd->output_columnar
                      ? launch_init_columns(col_layout_of(*qmd), make_row_init(*qmd), (*out)->buf, nullptr)
                      : launch_init_buffer((*out)->buf, qmd->entry_count, make_row_init(*qmd), nullptr);
  if (he == hipSuccess) he = hipStreamSynchronize(nullptr);
  if (he != hipSuccess) {
    last_hip_error = he;
    mi355q_result_free(*out);
    *out = nullptr;
    return MI355Q_ERR_HIP;
  }
  return MI355Q_OK;
}

int32_t mi355q_result_wrap(const mi355q_qmd* qmd, int32_t device_id, void* device_buffer,
                           mi355q_result** out) {
  if (!device_buffer) return MI355Q_ERR_INVALID_PLAN;
  return result_create_impl(qmd, device_id, device_buffer, out);
}

}  // extern "C"
int32_t mq::api::result_create_impl(const mi355q_qmd* qmd, int32_t device_id, void* device_buffer, mi355q_result** out) {
  if (!qmd || !out || qmd->row_size <= 0 || qmd->entry_count <= 0) return MI355Q_ERR_INVALID_PLAN;
  DeviceGuard g(device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  auto* r = new (std::nothrow) mi355q_result();
  if (!r) return MI355Q_ERR_OUT_OF_CPU_MEM;
  r->qmd = *qmd;
  r->device_id = device_id;
  r->bytes = qmd_buffer_bytes(*qmd);
  if (r->bytes <= 0) {
    delete r;
    return MI355Q_ERR_INVALID_PLAN;
  }
  // layout-only device plan (targets for reduce/iteration)
  DevPlan& d = r->dplan;
  std::memset(&d, 0, sizeof(d));
  layout_from_qmd(*qmd, &d);
  for (int i = 0; i < qmd->n_targets; ++i) d.targets[i].col = -1;
  if (device_buffer) {
    r->buf = (int64_t*)device_buffer;
  } else if (t_plan_only) {
    r->buf = (int64_t*)(uintptr_t)4096;  // mi355q_explain: nothing is launched, the table is never touched
  } else {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)r->bytes);
    if (e != hipSuccess) {
      last_hip_error = e;
      delete r;
      return MI355Q_ERR_OUT_OF_GPU_MEM;
    }
    r->buf = (int64_t*)p;
    r->owns_buf = true;
  }
  *out = r;
  return MI355Q_OK;
}

extern "C" {

void mi355q_result_free(mi355q_result* r) {
  if (!r) return;
  if (r->owns_buf && r->buf) {
    DeviceGuard g(r->device_id);
    (void)hipFree(r->buf);
  }
  delete r;
}

int32_t mi355q_result_qmd(const mi355q_result* r, mi355q_qmd* out) {
  if (!r || !out) return MI355Q_ERR_INVALID_PLAN;
  *out = r->qmd;
  return MI355Q_OK;
}
void* mi355q_result_device_ptr(const mi355q_result* r) { return r ? r->buf : nullptr; }
int64_t mi355q_result_bytes(const mi355q_result* r) { return r ? r->bytes : 0; }

int32_t mi355q_result_copy_to_host(const mi355q_result* r, void* dst, int64_t dst_bytes) {
  if (!r || !dst || dst_bytes < r->bytes) return MI355Q_ERR_INVALID_PLAN;
  DeviceGuard g(r->device_id);
  HIP_TRY(hipMemcpy(dst, r->buf, (size_t)r->bytes, hipMemcpyDeviceToHost));
  return MI355Q_OK;
}

extern "C++" {
int32_t mq::api::run_reduce(mi355q_result* dst, const int64_t* rows, int64_t n_rows, void* stream) {
  DeviceGuard g(dst->device_id);
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, sizeof(int32_t)));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(err.p, 0, sizeof(int32_t), s));
  HIP_TRY(launch_reduce(dst->dplan, dst->qmd.idx_target_as_key, dst->buf, rows, n_rows,
                        (int32_t*)err.p, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return h_err;
}
}  // extern "C++"

int32_t mi355q_result_reduce(mi355q_result* this_rs, const mi355q_result* that_rs, void* stream) {
  if (!this_rs || !that_rs) return MI355Q_ERR_INVALID_PLAN;
  const mi355q_qmd& a = this_rs->qmd;
  const mi355q_qmd& b = that_rs->qmd;
  // projections are not reduced but appended (Executor::resultsUnion, Execute.cpp:1670-1694)
  if (a.desc_type == MI355Q_PROJECTION && b.desc_type == MI355Q_PROJECTION) return projection_append(this_rs, that_rs, (hipStream_t)stream);
  if (a.desc_type != b.desc_type || a.row_size != b.row_size || a.slot_count != b.slot_count ||
      a.keyless != b.keyless || a.key_width != b.key_width || a.output_columnar != b.output_columnar ||
      this_rs->device_id != that_rs->device_id) {
    return MI355Q_ERR_INVALID_PLAN;
  }
  if (a.desc_type != MI355Q_GROUP_BY_BASELINE_HASH && a.entry_count != b.entry_count)
    return MI355Q_ERR_INVALID_PLAN;
  // same geometry is not enough: the slots are merged with THIS result's aggregate ops, so the
  // two descriptors must agree on what every slot holds (ResultSetStorage::reduce CHECKs the
  // descriptors' compatibility the same way, ResultSetReduction.cpp:203-215)
  if (a.n_targets != b.n_targets || a.slot_width != b.slot_width || a.idx_target_as_key != b.idx_target_as_key)
    return MI355Q_ERR_INVALID_PLAN;
  for (int t = 0; t < a.n_targets && t < MI355Q_MAX_TARGETS; ++t) {
    if (a.target_agg[t] != b.target_agg[t] || a.target_slot[t] != b.target_slot[t] ||
        a.target_skip_null[t] != b.target_skip_null[t] || a.target_arg_is_fp[t] != b.target_arg_is_fp[t] ||
        a.target_arg_is_f32[t] != b.target_arg_is_f32[t])
      return MI355Q_ERR_INVALID_PLAN;
  }
  for (int j = 0; j < a.slot_count && j < MI355Q_MAX_SLOTS; ++j)
    if (a.init_vals[j] != b.init_vals[j]) return MI355Q_ERR_INVALID_PLAN;
  if (a.output_columnar) {
    RowTwin ta, tb;
    if (int32_t e = make_row_twin(this_rs, (hipStream_t)stream, &ta)) return e;
    if (int32_t e = make_row_twin(that_rs, (hipStream_t)stream, &tb)) return e;
    if (int32_t e = mi355q_result_reduce(ta.tw, tb.tw, stream)) return e;
    return store_row_twin(ta, this_rs, (hipStream_t)stream);
  }
  return run_reduce(this_rs, that_rs->buf, b.entry_count, stream);
}

int32_t mi355q_result_append(mi355q_result* this_rs, const mi355q_result* that_rs, void* stream) {
  if (!this_rs || !that_rs) return MI355Q_ERR_INVALID_PLAN;
  return projection_append(this_rs, that_rs, (hipStream_t)stream);
}

int64_t mi355q_result_total_matched(const mi355q_result* r) {
  return r && r->qmd.desc_type == MI355Q_PROJECTION ? r->total_matched : -1;
}

int64_t mi355q_result_row_count(const mi355q_result* r) {
  if (!r) return -1;
  if (r->qmd.desc_type == MI355Q_NON_GROUPED_AGGREGATE) return 1;
  if (r->qmd.desc_type == MI355Q_PROJECTION) return projection_row_count(r);
  if (r->qmd.output_columnar) {
    RowTwin t;
    if (make_row_twin(r, nullptr, &t)) return -1;
    return mi355q_result_row_count(t.tw);
  }
  DeviceGuard g(r->device_id);
  DevWord cnt;
  if (hipMalloc(&cnt.p, sizeof(unsigned long long)) != hipSuccess) return -1;
  if (launch_count_nonempty(r->dplan, r->qmd.idx_target_as_key, r->buf,
                            (unsigned long long*)cnt.p, nullptr) != hipSuccess)
    return -1;
  unsigned long long h = 0;
  if (hipMemcpy(&h, cnt.p, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)h;
}

int32_t mi355q_result_to_columns(const mi355q_result* r, void* const* cols_dev, int32_t n_cols, int64_t* n_rows,
                                 void* stream) {
  if (!r || !cols_dev || !n_rows || n_cols != r->qmd.n_targets) return MI355Q_ERR_INVALID_PLAN;
  if (r->qmd.desc_type == MI355Q_NON_GROUPED_AGGREGATE) return MI355Q_ERR_UNSUPPORTED;  // one row: fetch_rows
  if (r->qmd.output_columnar) {
    RowTwin t;
    if (int32_t e = make_row_twin(r, (hipStream_t)stream, &t)) return e;
    return mi355q_result_to_columns(t.tw, cols_dev, n_cols, n_rows, stream);
  }
  const mi355q_qmd& q = r->qmd;
  if (q.entry_count >= ((int64_t)1 << 31)) return MI355Q_ERR_UNSUPPORTED;
  for (int t = 0; t < n_cols; ++t)
    if (!cols_dev[t]) return MI355Q_ERR_INVALID_PLAN;
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  const int64_t rows = mi355q_result_row_count(r);
  if (rows < 0) return MI355Q_ERR_HIP;
  *n_rows = rows;
  if (rows == 0) return MI355Q_OK;
  ColumnarSpec cs{};
  for (int t = 0; t < q.n_targets; ++t) {
    cs.null_pat[t] = q.target_null[t];
    cs.is_fp[t] = q.target_is_fp[t];
  }
  DevWord scratch;
  const size_t flag_bytes = ((size_t)q.entry_count * 4 + 255) & ~(size_t)255;
  const size_t tile_bytes = ((size_t)(q.entry_count / 2048 + 2) * 8 + 255) & ~(size_t)255;
  const size_t tab_bytes = sizeof(void*) * (size_t)q.n_targets;
  HIP_TRY(hipMalloc(&scratch.p, 2 * flag_bytes + tile_bytes + tab_bytes));
  char* base = (char*)scratch.p;
  int64_t** d_cols = (int64_t**)(base + 2 * flag_bytes + tile_bytes);
  HIP_TRY(hipMemcpyAsync(d_cols, cols_dev, tab_bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(launch_to_columns(r->dplan, q.idx_target_as_key, cs, r->buf, (int32_t*)base, (int32_t*)(base + flag_bytes),
                            (int64_t*)(base + 2 * flag_bytes), d_cols, s));
  HIP_TRY(hipStreamSynchronize(s));
  return MI355Q_OK;
}

// ---- Arrow C Data Interface export
namespace {

struct ArrowColumnOwner {          // private_data of one child array: its two host buffers
  std::vector<uint8_t> validity;   // empty when the column has no NULL
  std::vector<int64_t> values;
  const void* buffers[2];
};
struct ArrowBatchOwner {           // private_data of the struct array
  std::vector<ArrowArray> child_storage;
  std::vector<ArrowArray*> child_ptrs;
  const void* buffers[1];
};
struct ArrowSchemaOwner {          // private_data of the struct schema: the child structs (not their names)
  std::vector<ArrowSchema> child_storage;
  std::vector<ArrowSchema*> child_ptrs;
};

void release_child_array(ArrowArray* a) {
  if (!a || !a->release) return;
  delete static_cast<ArrowColumnOwner*>(a->private_data);
  a->release = nullptr;
}
void release_batch_array(ArrowArray* a) {
  if (!a || !a->release) return;
  ArrowBatchOwner* o = static_cast<ArrowBatchOwner*>(a->private_data);
  for (ArrowArray* c : o->child_ptrs)
    if (c->release) c->release(c);
  delete o;
  a->release = nullptr;
}
// a child schema owns its name: a consumer may move a child out of the parent (copy the struct, mark the source
// released) and release the parent first — the moved child's name must outlive the parent (Arrow C Data Interface,
// "moving child arrays")
void release_child_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  delete static_cast<std::string*>(s->private_data);
  s->private_data = nullptr;
  s->release = nullptr;
}
void release_batch_schema(ArrowSchema* s) {
  if (!s || !s->release) return;
  ArrowSchemaOwner* o = static_cast<ArrowSchemaOwner*>(s->private_data);
  for (ArrowSchema* c : o->child_ptrs)
    if (c->release) c->release(c);
  delete o;
  s->release = nullptr;
}

}  // namespace

static int32_t export_arrow_impl(const mi355q_result* r, const char* const* names, struct ArrowSchema* out_schema,
                                 struct ArrowArray* out_array, void* stream);

int32_t mi355q_result_export_arrow(const mi355q_result* r, const char* const* names, struct ArrowSchema* out_schema,
                                   struct ArrowArray* out_array, void* stream) {
  if (!r || !out_schema || !out_array) return MI355Q_ERR_INVALID_PLAN;
  *out_schema = ArrowSchema{};
  *out_array = ArrowArray{};
  // no C++ exception crosses the C boundary: an allocation failure is an error code, and what was built is released
  try {
    return export_arrow_impl(r, names, out_schema, out_array, stream);
  } catch (const std::bad_alloc&) {
  } catch (...) {
  }
  if (out_schema->release) out_schema->release(out_schema);
  if (out_array->release) out_array->release(out_array);
  *out_schema = ArrowSchema{};
  *out_array = ArrowArray{};
  return MI355Q_ERR_OUT_OF_CPU_MEM;
}

static int32_t export_arrow_impl(const mi355q_result* r, const char* const* names, struct ArrowSchema* out_schema,
                                 struct ArrowArray* out_array, void* stream) {
  const mi355q_qmd& q = r->qmd;
  const int nt = q.n_targets;
  int64_t n_rows = 0;
  std::vector<std::vector<int64_t>> cols((size_t)nt);
  std::vector<std::vector<uint8_t>> nulls((size_t)nt);  // one flag per row (converted to bitmaps below)
  if (q.desc_type == MI355Q_NON_GROUPED_AGGREGATE) {
    std::vector<int64_t> iv((size_t)nt);
    std::vector<double> dv((size_t)nt);
    std::vector<int8_t> nu((size_t)nt);
    if (int32_t e = mi355q_result_fetch_rows(r, 1, iv.data(), dv.data(), nu.data(), &n_rows)) return e;
    for (int t = 0; t < nt; ++t) {
      int64_t bits = iv[t];
      if (q.target_is_fp[t]) std::memcpy(&bits, &dv[t], 8);
      cols[t].assign((size_t)n_rows, bits);
      nulls[t].assign((size_t)n_rows, (uint8_t)(nu[t] != 0));
    }
  } else {
    n_rows = mi355q_result_row_count(r);
    if (n_rows < 0) return MI355Q_ERR_HIP;
    DeviceGuard g(r->device_id);
    if (!g.ok) return MI355Q_ERR_HIP;
    DevWord dev;
    const size_t col_bytes = ((size_t)std::max<int64_t>(n_rows, 1) * 8 + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc(&dev.p, col_bytes * (size_t)nt));
    std::vector<void*> ptrs((size_t)nt);
    for (int t = 0; t < nt; ++t) ptrs[t] = (char*)dev.p + col_bytes * (size_t)t;
    int64_t got = 0;
    if (int32_t e = mi355q_result_to_columns(r, ptrs.data(), nt, &got, stream)) return e;
    n_rows = got;
    for (int t = 0; t < nt; ++t) {
      cols[t].resize((size_t)n_rows);
      if (n_rows) HIP_TRY(hipMemcpy(cols[t].data(), ptrs[t], (size_t)n_rows * 8, hipMemcpyDeviceToHost));
      // NULL = the inline sentinel mi355q_result_to_columns wrote (target_null / NULL_DOUBLE)
      const int64_t null_bits = q.target_is_fp[t] ? kNullDoubleBits : q.target_null[t];
      const bool can_be_null = q.target_is_fp[t] ? (q.target_skip_null[t] || q.target_agg[t] == MI355Q_AVG ||
                                                    q.target_agg[t] == MI355Q_PROJECT_KEY)
                                                 : (q.target_skip_null[t] || q.target_agg[t] == MI355Q_PROJECT_KEY);
      nulls[t].assign((size_t)n_rows, 0);
      if (can_be_null)
        for (int64_t i = 0; i < n_rows; ++i) nulls[t][(size_t)i] = cols[t][(size_t)i] == null_bits;
    }
  }
  // ---- schema: struct<target_0: int64 | float64, ...>
  // (the parent is handed to the caller's struct FIRST, children are attached one by one: whatever exists when an
  // allocation fails is reachable from out_schema / out_array and released by the wrapper)
  ArrowSchemaOwner* so = new ArrowSchemaOwner();
  *out_schema = ArrowSchema{};
  out_schema->format = "+s";
  out_schema->name = "";
  out_schema->n_children = 0;
  out_schema->release = release_batch_schema;
  out_schema->private_data = so;
  so->child_storage.resize((size_t)nt);
  so->child_ptrs.resize((size_t)nt);
  for (int t = 0; t < nt; ++t) {
    so->child_storage[t] = ArrowSchema{};
    so->child_ptrs[t] = &so->child_storage[t];
  }
  out_schema->children = so->child_ptrs.data();
  out_schema->n_children = nt;
  for (int t = 0; t < nt; ++t) {
    std::string* nm = new std::string(names && names[t] ? std::string(names[t]) : "target_" + std::to_string(t));
    ArrowSchema& c = so->child_storage[t];
    c.format = q.target_is_fp[t] ? "g" : "l";
    c.name = nm->c_str();
    c.flags = 2;  // ARROW_FLAG_NULLABLE
    c.private_data = nm;
    c.release = release_child_schema;
  }
  // ---- array
  ArrowBatchOwner* bo = new ArrowBatchOwner();
  *out_array = ArrowArray{};
  out_array->release = release_batch_array;
  out_array->private_data = bo;
  bo->child_storage.resize((size_t)nt);
  bo->child_ptrs.resize((size_t)nt);
  for (int t = 0; t < nt; ++t) {
    bo->child_storage[t] = ArrowArray{};
    bo->child_ptrs[t] = &bo->child_storage[t];
  }
  for (int t = 0; t < nt; ++t) {
    std::unique_ptr<ArrowColumnOwner> co(new ArrowColumnOwner());
    int64_t null_count = 0;
    for (uint8_t f : nulls[t]) null_count += f;
    if (null_count) {
      co->validity.assign((size_t)((n_rows + 7) / 8), 0);
      for (int64_t i = 0; i < n_rows; ++i)
        if (!nulls[t][(size_t)i]) co->validity[(size_t)(i >> 3)] |= (uint8_t)(1u << (i & 7));
    }
    co->values = std::move(cols[t]);
    co->buffers[0] = null_count ? (const void*)co->validity.data() : nullptr;
    co->buffers[1] = co->values.data();
    ArrowArray& a = bo->child_storage[t];
    a.length = n_rows;
    a.null_count = null_count;
    a.n_buffers = 2;
    a.buffers = co->buffers;
    a.release = release_child_array;
    a.private_data = co.release();
  }
  bo->buffers[0] = nullptr;
  out_array->length = n_rows;
  out_array->n_buffers = 1;
  out_array->buffers = bo->buffers;
  out_array->n_children = nt;
  out_array->children = bo->child_ptrs.data();
  return MI355Q_OK;
}

int32_t mi355q_result_topk(const mi355q_result* r, int32_t target_idx, int32_t descending,
                           int32_t nulls_first, int64_t k, void* out_rows_dev, int64_t* n_rows,
                           void* stream) {
  if (!r || !out_rows_dev || !n_rows || target_idx < 0 || target_idx >= r->qmd.n_targets || k < 1)
    return MI355Q_ERR_INVALID_PLAN;
  if (r->qmd.desc_type == MI355Q_NON_GROUPED_AGGREGATE) return MI355Q_ERR_UNSUPPORTED;
  if (k > topk_max_k()) return MI355Q_ERR_UNSUPPORTED;
  // ordering by a floating-point KEY projection (read from the key column) is not built
  if (r->qmd.target_agg[target_idx] == MI355Q_PROJECT_KEY && r->qmd.target_is_fp[target_idx])
    return MI355Q_ERR_UNSUPPORTED;
  if (r->qmd.output_columnar) {  // the rows come out row-wise (row_size bytes each)
    RowTwin t;
    if (int32_t e = make_row_twin(r, (hipStream_t)stream, &t)) return e;
    return mi355q_result_topk(t.tw, target_idx, descending, nulls_first, k, out_rows_dev, n_rows, stream);
  }
  if (k > r->qmd.entry_count) k = r->qmd.entry_count;
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  DevWord scratch;
  HIP_TRY(hipMalloc(&scratch.p, (size_t)topk_scratch_bytes(r->qmd.entry_count) + 64));
  int64_t* d_n = (int64_t*)((char*)scratch.p + topk_scratch_bytes(r->qmd.entry_count));
  const mi355q_qmd& q = r->qmd;
  HIP_TRY(launch_topk(r->dplan, q.idx_target_as_key, target_idx, q.target_null[target_idx],
                      q.target_is_fp[target_idx] != 0, descending != 0, nulls_first != 0, r->buf, k,
                      scratch.p, (int64_t*)out_rows_dev, d_n, s));
  HIP_TRY(hipMemcpyAsync(n_rows, d_n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return MI355Q_OK;
}

int32_t mi355q_result_sort(const mi355q_result* r, const mi355q_order_entry* order, int32_t n_order,
                           int64_t limit, int64_t offset, void* out_rows_dev, int64_t* n_rows, void* stream) {
  if (!r || !order || !out_rows_dev || !n_rows || n_order < 1 || n_order > MI355Q_MAX_TARGETS || limit < 0 ||
      offset < 0)
    return MI355Q_ERR_INVALID_PLAN;
  if (r->qmd.desc_type == MI355Q_NON_GROUPED_AGGREGATE) return MI355Q_ERR_UNSUPPORTED;
  SortOrderEntry oe[MI355Q_MAX_TARGETS];
  for (int i = 0; i < n_order; ++i) {
    const int t = order[i].target_idx;
    if (t < 0 || t >= r->qmd.n_targets) return MI355Q_ERR_INVALID_PLAN;
    // ordering by a floating-point KEY projection (read from the key column) is not built
    if (r->qmd.target_agg[t] == MI355Q_PROJECT_KEY && r->qmd.target_is_fp[t]) return MI355Q_ERR_UNSUPPORTED;
    oe[i] = SortOrderEntry{t, order[i].descending != 0, order[i].nulls_first != 0, r->qmd.target_is_fp[t] != 0,
                           r->qmd.target_null[t]};
  }
  if (r->qmd.output_columnar) {  // the rows come out row-wise (row_size bytes each)
    RowTwin t;
    if (int32_t e = make_row_twin(r, (hipStream_t)stream, &t)) return e;
    return mi355q_result_sort(t.tw, order, n_order, limit, offset, out_rows_dev, n_rows, stream);
  }
  if (r->qmd.entry_count >= ((int64_t)1 << 32)) return MI355Q_ERR_UNSUPPORTED;
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  DevWord scratch;
  const int64_t sb = sort_scratch_bytes(r->qmd.entry_count);
  HIP_TRY(hipMalloc(&scratch.p, (size_t)sb + 64));
  int64_t* d_n = (int64_t*)((char*)scratch.p + sb);
  HIP_TRY(launch_sort(r->dplan, r->qmd.idx_target_as_key, oe, n_order, r->buf, offset, limit, scratch.p,
                      (int64_t*)out_rows_dev, d_n, s));
  HIP_TRY(hipMemcpyAsync(n_rows, d_n, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return MI355Q_OK;
}

// Host-side iteration over the copied-back buffer, like ResultSet::getNextRow
// (ResultSetIteration.cpp:125-230, getTargetValueFromBufferRowwise) for 8-byte slots.
int32_t mi355q_result_fetch_rows(const mi355q_result* r, int64_t max_rows, int64_t* ival,
                                 double* dval, int8_t* is_null, int64_t* n_rows) {
  if (!r || !ival || !dval || !is_null || !n_rows) return MI355Q_ERR_INVALID_PLAN;
  if (r->qmd.desc_type == MI355Q_PROJECTION) return projection_fetch_rows(r, max_rows, ival, dval, is_null, n_rows);
  if (r->qmd.output_columnar) {
    RowTwin t;
    if (int32_t e = make_row_twin(r, nullptr, &t)) return e;
    return mi355q_result_fetch_rows(t.tw, max_rows, ival, dval, is_null, n_rows);
  }
  const mi355q_qmd& q = r->qmd;
  std::vector<int64_t> host;
  try {
    host.resize((size_t)(r->bytes / 8));
  } catch (...) {
    return MI355Q_ERR_OUT_OF_CPU_MEM;
  }
  if (int32_t e = mi355q_result_copy_to_host(r, host.data(), r->bytes)) return e;
  const int rq = q.row_size / 8, kq = q.key_bytes / 8, nt = q.n_targets;
  int64_t n = 0;
  for (int64_t e = 0; e < q.entry_count && n < max_rows; ++e) {
    const int64_t* row = host.data() + e * rq;
    bool empty = false;
    if (q.desc_type != MI355Q_NON_GROUPED_AGGREGATE) {
      if (q.keyless && q.slot_width == 4) {
        empty = ((const int32_t*)row)[q.idx_target_as_key] == (int32_t)q.init_vals[q.idx_target_as_key];
      } else if (q.keyless) {
        empty = row[q.idx_target_as_key] == q.init_vals[q.idx_target_as_key];
      } else if (q.key_width == 4) {
        empty = *(const int32_t*)row == kEmptyKey32;
      } else {
        empty = row[0] == kEmptyKey64;
      }
    }
    if (empty) continue;
    for (int t = 0; t < nt; ++t) {
      const size_t o = (size_t)n * nt + t;
      ival[o] = 0;
      dval[o] = 0.0;
      is_null[o] = 0;
      const int s = q.target_slot[t];
      const int agg = q.target_agg[t];
      if (agg == MI355Q_PROJECT_KEY && s < 0) {
        const int ki = q.target_key_idx[t];
        ival[o] = q.key_width == 4 ? (int64_t)((const int32_t*)row)[ki] : row[ki];
        is_null[o] = ival[o] == q.target_null[t];
        if (q.target_is_fp[t]) {  // floating-point key: the quad holds double bits (FLOAT widened)
          dval[o] = bits_dbl(ival[o]);
          ival[o] = 0;
        }
        continue;
      }
      // compact layouts hold 32-bit slots (read_int_from_buff with the slot's width)
      const int64_t v = q.slot_width == 4 ? (int64_t)((const int32_t*)(row + kq))[s] : row[kq + s];
      if (agg == MI355Q_AVG) {
        const int64_t cnt = row[kq + s + 1];
        if (cnt == 0) {  // pair_to_double: count 0 -> NULL_DOUBLE
          dval[o] = kNullDouble;
          is_null[o] = 1;
        } else {
          const double sum = q.target_arg_is_f32[t] ? (double)bits_flt((int32_t)v)
                                                    : q.target_arg_is_fp[t] ? bits_dbl(v) : (double)v;
          dval[o] = sum / (double)cnt;
        }
      } else if (q.target_arg_is_f32[t]) {  // float result in the low 4 bytes of the slot
        dval[o] = (double)bits_flt((int32_t)v);
        is_null[o] = q.target_skip_null[t] && (int32_t)v == (int32_t)q.target_null[t];
      } else if (agg == MI355Q_COUNT || agg == MI355Q_COUNT_IF) {
        ival[o] = v;
      } else if (q.target_is_fp[t]) {
        dval[o] = bits_dbl(v);
        is_null[o] = q.target_skip_null[t] && v == q.target_null[t];
      } else {
        ival[o] = v;
        const bool nullable = q.target_skip_null[t] || agg == MI355Q_PROJECT_KEY;
        is_null[o] = nullable && v == q.target_null[t];
      }
    }
    ++n;
  }
  *n_rows = n;
  return MI355Q_OK;
}

// ------------------------------------------------------------------------------- shards
int32_t mi355q_shard_partition(const mi355q_result* r, int32_t n_parts, void* out_rows,
                               int64_t* part_counts_dev, void* stream) {
  if (!r || !out_rows || !part_counts_dev || n_parts < 1 || n_parts > 256)
    return MI355Q_ERR_INVALID_PLAN;
  if (r->qmd.desc_type != MI355Q_GROUP_BY_BASELINE_HASH) return MI355Q_ERR_UNSUPPORTED;
  if (r->qmd.output_columnar) {  // the exchanged rows are row-wise whatever the table's layout
    RowTwin t;
    if (int32_t e = make_row_twin(r, (hipStream_t)stream, &t)) return e;
    return mi355q_shard_partition(t.tw, n_parts, out_rows, part_counts_dev, stream);
  }
  DeviceGuard g(r->device_id);
  DevWord cur;
  HIP_TRY(hipMalloc(&cur.p, sizeof(int64_t) * 256));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(launch_shard_partition(r->dplan, r->qmd.idx_target_as_key, r->buf, n_parts,
                                 (int64_t*)out_rows, part_counts_dev, (int64_t*)cur.p, s));
  HIP_TRY(hipStreamSynchronize(s));
  return MI355Q_OK;
}

int32_t mi355q_shard_merge_rows(mi355q_result* r, const void* rows, int64_t n_rows, void* stream) {
  if (!r || (!rows && n_rows > 0) || n_rows < 0) return MI355Q_ERR_INVALID_PLAN;
  if (n_rows == 0) return MI355Q_OK;
  if (r->qmd.desc_type != MI355Q_GROUP_BY_BASELINE_HASH) return MI355Q_ERR_UNSUPPORTED;
  if (r->qmd.output_columnar) {
    RowTwin t;
    if (int32_t e = make_row_twin(r, (hipStream_t)stream, &t)) return e;
    if (int32_t e = mi355q_shard_merge_rows(t.tw, rows, n_rows, stream)) return e;
    return store_row_twin(t, r, (hipStream_t)stream);
  }
  return run_reduce(r, (const int64_t*)rows, n_rows, stream);
}

static bool slice_exchange_shape(const mi355q_qmd& q) {
  return q.desc_type == MI355Q_GROUP_BY_BASELINE_HASH && !q.output_columnar && q.group_col_count == 1 &&
         q.key_width == 8 && q.slot_width == 8 && q.entry_count < ((int64_t)1 << 32);
}

int32_t mi355q_shard_pads(const mi355q_result* r, int32_t world, int32_t pad_rows, void* out_pads_dev,
                          int32_t* ok_dev, void* stream) {
  if (!r || !out_pads_dev || !ok_dev || world < 1 || world > 1024 || pad_rows < 1 || pad_rows > r->qmd.entry_count)
    return MI355Q_ERR_INVALID_PLAN;
  if (!slice_exchange_shape(r->qmd)) return MI355Q_ERR_UNSUPPORTED;
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  HIP_TRY(launch_shard_pads(r->dplan, r->buf, world, pad_rows, (int64_t*)out_pads_dev, ok_dev, (hipStream_t)stream));
  return MI355Q_OK;
}

int32_t mi355q_shard_merge_range(mi355q_result* r, const void* rows, int64_t n_rows, int64_t home_lo,
                                 int64_t home_hi, void* stream) {
  if (!r || (!rows && n_rows > 0) || n_rows < 0 || home_lo < 0 || home_hi < home_lo || home_hi > r->qmd.entry_count)
    return MI355Q_ERR_INVALID_PLAN;
  if (!slice_exchange_shape(r->qmd)) return MI355Q_ERR_UNSUPPORTED;
  if (n_rows == 0) return MI355Q_OK;
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, sizeof(int32_t)));
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(err.p, 0, sizeof(int32_t), s));
  HIP_TRY(launch_reduce_range(r->dplan, r->qmd.idx_target_as_key, r->buf, (const int64_t*)rows, n_rows, home_lo, home_hi,
                              (int32_t*)err.p, s));
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return h_err;
}

int32_t mi355q_shard_merge_slices(mi355q_result* r, const void* const* slices, const void* const* pads, int32_t n_src,
                                  int32_t pad_rows, int64_t home_lo, int64_t home_hi, void* stream) {
  if (!r || !slices || n_src < 1 || pad_rows < 0 || home_lo < 0 || home_hi <= home_lo || home_hi > r->qmd.entry_count)
    return MI355Q_ERR_INVALID_PLAN;
  if (!slice_exchange_shape(r->qmd) || n_src > 16) return MI355Q_ERR_UNSUPPORTED;
  for (int i = 0; i < n_src; ++i)
    if (!slices[i] || (pads && pad_rows > 0 && !pads[i])) return MI355Q_ERR_INVALID_PLAN;
  DeviceCtx& ctx = ctx_of(r->device_id);
  std::lock_guard<std::recursive_mutex> lk(ctx.mu);
  DeviceGuard g(r->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  const int n_cus = cu_count_of(r->device_id);
  const int64_t need = slice_merge_scratch_bytes(r->dplan, home_lo, home_hi, n_cus);
  if (need == 0) return MI355Q_ERR_UNSUPPORTED;  // the caller folds with mi355q_shard_merge_range
  if (need + 64 > ctx.scratch_bytes) {
    if (ctx.scratch) (void)hipFree(ctx.scratch);
    ctx.scratch = nullptr;
    ctx.scratch_bytes = 0;
    HIP_TRY(hipMalloc(&ctx.scratch, (size_t)need + 64));
    ctx.scratch_bytes = need + 64;
  }
  hipStream_t s = (hipStream_t)stream;
  // the error word: the last 64 bytes of the workspace
  int32_t* d_err = (int32_t*)((char*)ctx.scratch + ((need + 7) & ~(int64_t)7));
  HIP_TRY(hipMemsetAsync(d_err, 0, 2 * sizeof(int32_t), s));
  HIP_TRY(launch_slice_merge(r->dplan, r->buf, (const int64_t* const*)slices, pads && pad_rows > 0 ? (const int64_t* const*)pads : nullptr,
                             n_src, pad_rows, home_lo, home_hi, d_err, ctx.scratch, need, n_cus, s));
  int32_t h_err[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  // word 0: the table ran out of group slots; word 1: the stray list overflowed.  Either way rows
  // [home_lo, home_hi) of r are incomplete: re-initialise r and fold with mi355q_shard_merge_range.
  return (h_err[0] || h_err[1]) ? MI355Q_ERR_OUT_OF_SLOTS : MI355Q_OK;
}

}  // extern "C"
