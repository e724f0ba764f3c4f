// api_join.cpp — join hash tables of the C-ABI (include/mi355q.h): mi355q_join_build (perfect / keyed, OneToOne ->
// OneToMany, the four buffer layouts of the reference's docs hash_joins.rst), payload arrays, info.  Split out of api.cpp
// in round 5.
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

extern "C" {

// ------------------------------------------------------------------------------- joins
namespace {

// One attempt at one layout.  `one_to_many` selects hash types 2/3 instead of 0/1.
int32_t join_build_layout(const mi355q_join_spec* spec, bool perfect, bool one_to_many,
                          const JoinKeyCols& kc, hipStream_t s, mi355q_join_table* jt, int32_t* d_err) {
  const mi355q_range& r = spec->key_range;
  const int64_t n = spec->num_rows;
  if (jt->buf) (void)hipFree(jt->buf);
  if (jt->bitmap) (void)hipFree(jt->bitmap);
  jt->buf = jt->bitmap = nullptr;
  HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int32_t), s));
  jt->n_keys = kc.n;
  jt->width = perfect ? 8 : kc.width;
  auto alloc = [&](int64_t bytes) -> int32_t {
    jt->bytes = bytes;
    hipError_t e = hipMalloc(&jt->buf, (size_t)(bytes > 0 ? bytes : 4));
    if (e != hipSuccess) {
      last_hip_error = e;
      return MI355Q_ERR_OUT_OF_GPU_MEM;
    }
    return MI355Q_OK;
  };
  if (perfect) {
    jt->min_key = r.min;
    jt->max_key = r.max;
    jt->entry_count = r.max - r.min + 1;
  } else {
    jt->min_key = jt->max_key = 0;
    jt->entry_count = spec->keyed_entry_count > 0 ? spec->keyed_entry_count
                                                  : 2 * std::max<int64_t>(n, 1);  // BaselineJoinHashTable.cpp:484
    if (jt->entry_count > (int64_t)UINT32_MAX) return MI355Q_ERR_UNSUPPORTED;
  }
  const int64_t entries = jt->entry_count;
  if (!one_to_many) {
    if (perfect) {
      jt->hash_type = 0;
      if (int32_t e = alloc(entries * (int64_t)sizeof(int32_t))) return e;
      HIP_TRY(hipMemsetAsync(jt->buf, 0xFF, (size_t)jt->bytes, s));  // init_hash_join_buff: -1
      HIP_TRY(launch_join_fill_perfect(kc.col[0], kc.type[0], kc.nullable[0], n, r.min, r.max,
                                       (int32_t*)jt->buf, d_err, s));
      const size_t bm_bytes = (size_t)((entries + 31) / 32) * 4;
      hipError_t be = hipMalloc(&jt->bitmap, bm_bytes);
      if (be != hipSuccess) {
        last_hip_error = be;
        return MI355Q_ERR_OUT_OF_GPU_MEM;
      }
      HIP_TRY(launch_join_presence_bitmap((const int32_t*)jt->buf, entries, (uint32_t*)jt->bitmap, s));
    } else {
      jt->hash_type = 1;
      const int stride = kc.n + 1;
      if (int32_t e = alloc(entries * stride * kc.width)) return e;
      HIP_TRY(launch_join_init_keyed(jt->buf, entries, kc.n, stride, kc.width, s));
      HIP_TRY(launch_join_fill_keyed(kc, n, jt->buf, entries, stride, true, d_err, s));
    }
    return MI355Q_OK;
  }
  // one-to-many: [keys |] offsets | counts | payloads
  jt->hash_type = perfect ? 2 : 3;
  const int64_t key_bytes = perfect ? 0 : entries * kc.n * kc.width;
  if (int32_t e = alloc(key_bytes + (2 * entries + std::max<int64_t>(n, 1)) * (int64_t)sizeof(int32_t))) return e;
  if (!perfect) {
    HIP_TRY(launch_join_init_keyed(jt->buf, entries, kc.n, kc.n, kc.width, s));
    HIP_TRY(launch_join_fill_keyed(kc, n, jt->buf, entries, kc.n, false, d_err, s));
  }
  int32_t* offsets = (int32_t*)((int8_t*)jt->buf + key_bytes);
  DevWord tiles;
  HIP_TRY(hipMalloc(&tiles.p, sizeof(int64_t) * (size_t)(entries / 2048 + 2)));
  HIP_TRY(launch_join_one_to_many(kc, n, jt->hash_type, jt->buf, entries, jt->min_key, jt->max_key, offsets,
                                  offsets + entries, offsets + 2 * entries, (int64_t*)tiles.p, d_err, s));
  HIP_TRY(hipStreamSynchronize(s));  // tiles is freed on return
  return MI355Q_OK;
}

}  // namespace

int32_t mi355q_join_build(const mi355q_join_spec* spec, void* stream, mi355q_join_table** out) {
  if (!spec || !out || spec->num_rows < 0) return MI355Q_ERR_INVALID_PLAN;
  if (spec->num_rows > (int64_t)INT32_MAX) return MI355Q_ERR_UNSUPPORTED;  // int32 row ids
  const int n_keys = spec->n_keys > 1 ? spec->n_keys : 1;
  if (n_keys > MI355Q_MAX_GROUP_COLS) return MI355Q_ERR_INVALID_PLAN;
  JoinKeyCols kc{};
  kc.n = n_keys;
  kc.width = 4;
  for (int i = 0; i < n_keys; ++i) {
    kc.col[i] = (const int8_t*)(i == 0 ? spec->key_buffer : spec->more_key_buffers[i - 1]);
    kc.type[i] = i == 0 ? spec->key_type : spec->more_key_types[i - 1];
    kc.nullable[i] = i == 0 ? spec->key_nullable : spec->more_key_nullables[i - 1];
    if (kc.type[i] < MI355Q_INT8 || kc.type[i] > MI355Q_INT64) return MI355Q_ERR_UNSUPPORTED;
    if (spec->num_rows > 0 && !kc.col[i]) return MI355Q_ERR_INVALID_PLAN;
    // BaselineJoinHashTable::getKeyComponentWidth: 8 iff an inner key column is wider than 4 bytes
    if (type_width(kc.type[i]) > 4) kc.width = 8;
  }
  *out = nullptr;
  DeviceGuard g(spec->device_id);
  if (!g.ok) return MI355Q_ERR_HIP;
  hipStream_t s = (hipStream_t)stream;
  auto* jt = new (std::nothrow) mi355q_join_table();
  if (!jt) return MI355Q_ERR_OUT_OF_CPU_MEM;
  struct JG {
    mi355q_join_table* j;
    ~JG() { mi355q_join_free(j); }
  } jg{jt};
  jt->device_id = spec->device_id;
  jt->key_type = spec->key_type;
  const mi355q_range& r = spec->key_range;
  // PerfectJoinHashTable::getInstance (PerfectJoinHashTable.cpp:168-246): perfect when there is
  // ONE key column whose range is known and max-min+1 entries fit; else keyed
  // (HashJoin.cpp:340-372).
  int64_t max_entries = spec->max_perfect_entries > 0 ? spec->max_perfect_entries : (int64_t)INT32_MAX;
  const bool perfect = n_keys == 1 && !spec->prefer_baseline && r.valid && r.max >= r.min &&
                       ((__int128)r.max - (__int128)r.min) < (__int128)max_entries;
  DevWord err;
  HIP_TRY(hipMalloc(&err.p, sizeof(int32_t)));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  struct EG {
    hipEvent_t a, b;
    ~EG() {
      (void)hipEventDestroy(a);
      (void)hipEventDestroy(b);
    }
  } eg{e0, e1};
  HIP_TRY(hipEventRecord(e0, s));
  int32_t h_err = 0;
  // the reference tries OneToOne first and rebuilds as OneToMany when the fill reports a
  // duplicate key (PerfectJoinHashTable::reify / BaselineJoinHashTable::reify)
  for (int attempt = spec->one_to_many == 2 ? 1 : 0; attempt < 2; ++attempt) {
    if (int32_t e = join_build_layout(spec, perfect, attempt == 1, kc, s, jt, (int32_t*)err.p)) return e;
    HIP_TRY(hipMemcpyAsync(&h_err, err.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (h_err != MI355Q_ERR_JOIN_NOT_ONE_TO_ONE || spec->one_to_many == 0) break;
  }
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipStreamSynchronize(s));
  (void)hipEventElapsedTime(&jt->build_ms, e0, e1);
  if (h_err) return h_err;
  jt->dense = jt->hash_type == 0 && !spec->key_nullable && spec->num_rows == jt->entry_count;
  jg.j = nullptr;
  *out = jt;
  return MI355Q_OK;
}

int32_t mi355q_join_invalidate_payload(mi355q_join_table* t) {
  if (!t) return MI355Q_ERR_INVALID_PLAN;
  std::lock_guard<std::mutex> pl(t->pay_mu);
  // the buffers are kept (the next build reuses them); only their validity goes
  t->pay16_built = false;
  t->pay_col_built = false;
  t->pay16_col = nullptr;
  t->pay_col = nullptr;
  t->pay_refused = false;
  return MI355Q_OK;
}

int32_t mi355q_join_payload_info(const mi355q_join_table* t, int64_t* bytes, float* build_ms, int64_t* inner_version) {
  if (!t) return MI355Q_ERR_INVALID_PLAN;
  mi355q_join_table* jt = const_cast<mi355q_join_table*>(t);
  std::lock_guard<std::mutex> pl(jt->pay_mu);
  int64_t b = 0;
  const int64_t n = t->entry_count;
  if (t->pay_cnt) b += n * 4;
  if (t->pay_wsum) b += n * 8;
  if (t->pay_wnn) b += n * 4;
  if (t->pay16) b += n * 16;
  if (t->pay8) b += n * (t->pay_kkeys ? 16 : 8);
  if (t->pay_kkeys) b += n * 8;
  if (bytes) *bytes = b;
  if (build_ms) *build_ms = t->pay_build_ms;
  if (inner_version) *inner_version = t->pay16_built ? t->pay16_version : t->pay_version;
  return MI355Q_OK;
}

int32_t mi355q_join_key_shape(const mi355q_join_table* t, int32_t* key_components, int32_t* component_width) {
  if (!t) return MI355Q_ERR_INVALID_PLAN;
  if (key_components) *key_components = t->n_keys;
  if (component_width) *component_width = t->width;
  return MI355Q_OK;
}

void mi355q_join_free(mi355q_join_table* t) {
  if (!t) return;
  if (t->buf || t->bitmap || t->pay_cnt || t->pay16) {
    DeviceGuard g(t->device_id);
    if (t->buf) (void)hipFree(t->buf);
    if (t->bitmap) (void)hipFree(t->bitmap);
    if (t->pay_cnt) (void)hipFree(t->pay_cnt);
    if (t->pay_wsum) (void)hipFree(t->pay_wsum);
    if (t->pay_wnn) (void)hipFree(t->pay_wnn);
    if (t->pay16) (void)hipFree(t->pay16);
    if (t->pay8) (void)hipFree(t->pay8);
    if (t->pay_kkeys) (void)hipFree(t->pay_kkeys);
  }
  delete t;
}

int32_t mi355q_join_info(const mi355q_join_table* t, int32_t* hash_type, int64_t* entry_count,
                         int64_t* min_key, int64_t* max_key, void** device_ptr, int64_t* bytes,
                         float* build_ms) {
  if (!t) return MI355Q_ERR_INVALID_PLAN;
  if (hash_type) *hash_type = t->hash_type;
  if (entry_count) *entry_count = t->entry_count;
  if (min_key) *min_key = t->min_key;
  if (max_key) *max_key = t->max_key;
  if (device_ptr) *device_ptr = t->buf;
  if (bytes) *bytes = t->bytes;
  if (build_ms) *build_ms = t->build_ms;
  return MI355Q_OK;
}

}  // extern "C"
