// kernels_generic.hip — the generic members of the kernel family: any plan the C-ABI accepts
// runs here (row interpreter with atomic slot updates); the fast families in
// kernels_fast.hip take over at plan time for the shapes they cover.
//
// gfx950 only: 256-thread workgroups (4 waves, one per SIMD), grid-stride over each
// fragment like the reference's multi-fragment kernel loop
// (multifrag_query_hoisted_literals, RuntimeFunctions.cpp:2434-2471).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include "kernels.h"
#include "rowfunc.h"
#include "expr.h"

namespace mq {

namespace {

constexpr int kBlock = 256;

// the step's error words (+ the spill counter's copy) into pinned host memory, behind the step's kernels on their stream
__global__ __launch_bounds__(64) void k_words_to_host(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void k_init_buffer(int64_t* __restrict__ buf,
                                                         int64_t total_quads, RowInit init) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total_quads; i += stride) {
    buf[i] = init.quad[i % init.row_quad];
  }
}

// Generic row kernel.  Grouped plans update the output table with device atomics; the
// non-grouped plan keeps per-thread partial rows in LDS, folds them per block and issues one
// atomic merge per target per block (the reference folds per-thread partials on the host,
// Executor::reduceResults Execute.cpp:1444).
__global__ __launch_bounds__(kBlock) void k_generic(DevPlan p, const int8_t* const* __restrict__ cols,
                                                     const int64_t* __restrict__ num_rows,
                                                     int n_frags, int64_t* __restrict__ out,
                                                     int32_t* __restrict__ d_err) {
  constexpr int kLocStride = MI355Q_MAX_SLOTS + 1;
  __shared__ int64_t s_loc[kBlock * kLocStride];
  const bool ng = p.desc_type == MI355Q_NON_GROUPED_AGGREGATE;
  int64_t* my_loc = s_loc + threadIdx.x * kLocStride;
  if (ng) {
    for (int i = 0; i < p.slot_count; ++i) my_loc[i] = p.init_vals[i];
  }
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  int32_t err = 0;
  for (int f = 0; f < n_frags && !err; ++f) {
    const int8_t* const* fc = cols + (size_t)f * p.n_cols;
    const int64_t n = num_rows[f];
    for (int64_t pos = gtid; pos < n; pos += gsize) {
      const int32_t e = ng ? process_row<false>(p, fc, pos, out, my_loc)
                           : process_row<true>(p, fc, pos, out, nullptr);
      if (e) {
        err = e;
        break;
      }
    }
  }
  if (err) atomicCAS(d_err, 0, err);
  if (ng) {
    __syncthreads();
    if (threadIdx.x < p.n_targets) {
      const DevTarget& t = p.targets[threadIdx.x];
      int64_t acc[2] = {p.init_vals[t.slot], t.agg == MI355Q_AVG ? p.init_vals[t.slot + 1] : 0};
      // fold the block's 256 partial rows for this target, serially, with plain ops
      DevTarget lt = t;
      lt.slot = 0;
      int64_t lin[2] = {p.init_vals[t.slot], 0};
      for (int th = 0; th < kBlock; ++th) {
        const int64_t* src = s_loc + th * kLocStride + t.slot;
        reduce_target<false>(lt, lin, acc, src);
      }
      // one atomic merge per target per block
      DevTarget gt = t;
      int64_t that[MI355Q_MAX_SLOTS];
      that[0] = acc[0];
      that[1] = acc[1];
      gt.slot = 0;
      reduce_target<true>(gt, lin, out + t.slot, that);
    }
  }
}

// Generic row kernel for perfect-hash tables that fit LDS: every workgroup aggregates into a
// private LDS copy of the table (same row function, same atomics — they act on LDS instead of
// on one contended set of global rows), then folds its copy into the output with the
// ResultSetStorage::reduce rule per target (index aligned).  With 1 K groups the plain generic
// kernel serialises a billion rows on 1 K global addresses; this one touches them once per
// workgroup.
__global__ __launch_bounds__(kBlock) void k_generic_lds(DevPlan p, int idx_target_as_key, RowInit init,
                                                         const int8_t* const* __restrict__ cols,
                                                         const int64_t* __restrict__ num_rows, int n_frags,
                                                         int64_t* __restrict__ out,
                                                         int32_t* __restrict__ d_err) {
  extern __shared__ __attribute__((aligned(16))) int64_t s_tab[];
  const int64_t quads = p.entry_count * p.row_quad;
  for (int64_t i = threadIdx.x; i < quads; i += kBlock) s_tab[i] = init.quad[i % init.row_quad];
  __syncthreads();
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  int32_t err = 0;
  for (int f = 0; f < n_frags && !err; ++f) {
    const int8_t* const* fc = cols + (size_t)f * p.n_cols;
    const int64_t n = num_rows[f];
    for (int64_t pos = gtid; pos < n; pos += gsize) {
      const int32_t e = process_row<true>(p, fc, pos, s_tab, nullptr);
      if (e) {
        err = e;
        break;
      }
    }
  }
  if (err) atomicCAS(d_err, 0, err);
  __syncthreads();
  for (int64_t e = threadIdx.x; e < p.entry_count; e += kBlock) {
    const int64_t* src = s_tab + e * p.row_quad;
    if (is_empty_row(p, src, idx_target_as_key)) continue;
    int64_t* row = out + e * p.row_quad;
    for (int k = p.key_quad - 1; k >= 0; --k) MQ_STORE64(row + k, src[k]);
    for (int i = 0; i < p.n_targets; ++i) {
      reduce_target<true>(p.targets[i], p.init_vals, row + p.key_quad, src + p.key_quad);
    }
  }
}

// this (op)= that, entry-wise.  `that` holds that_entries rows of the same layout; baseline
// rows are re-hashed into `this` (get_group_value_reduction, ResultSetReduction.cpp:783-826),
// perfect/non-grouped rows are index aligned.
__global__ __launch_bounds__(kBlock) void k_reduce(DevPlan p, int idx_target_as_key,
                                                    int64_t* __restrict__ this_buf,
                                                    const int64_t* __restrict__ that_rows,
                                                    int64_t that_entries,
                                                    int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < that_entries; e += stride) {
    const int32_t err = reduce_entry<true>(p, idx_target_as_key, this_buf, that_rows + e * p.row_quad, e);
    if (err) atomicCAS(d_err, 0, err);
  }
}

__global__ __launch_bounds__(kBlock) void k_count_nonempty(DevPlan p, int idx_target_as_key,
                                                            const int64_t* __restrict__ buf,
                                                            unsigned long long* __restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  unsigned long long local = 0;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < p.entry_count; e += stride) {
    local += !is_empty_row(p, buf + e * p.row_quad, idx_target_as_key);
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(cnt, local);
}

MQ_D uint32_t shard_of(const DevPlan& p, const int64_t* row, int n_parts) {
  if (p.n_group > 1) {  // the whole key's hash, upper bits
    const int n_words = p.n_group * (p.key_width / 4);
    return (uint32_t)(((uint64_t)murmur3_words((const uint32_t*)row, n_words) * (uint64_t)n_parts) >> 32);
  }
  if (p.key_width == 4) return murmur3_u32((uint32_t) * (const int32_t*)row) % (uint32_t)n_parts;
  // upper hash bits, so a shard's keys still spread over the whole local table
  return (uint32_t)(((uint64_t)murmur3_u64((uint64_t)row[0]) * (uint64_t)n_parts) >> 32);
}

__global__ __launch_bounds__(kBlock) void k_shard_count(DevPlan p, int idx_target_as_key,
                                                         const int64_t* __restrict__ buf,
                                                         int n_parts,
                                                         int64_t* __restrict__ part_counts) {
  __shared__ unsigned long long s_cnt[256];
  for (int i = threadIdx.x; i < n_parts; i += kBlock) s_cnt[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < p.entry_count; e += stride) {
    const int64_t* row = buf + e * p.row_quad;
    if (is_empty_row(p, row, idx_target_as_key)) continue;
    atomicAdd(&s_cnt[shard_of(p, row, n_parts)], 1ull);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_parts; i += kBlock) {
    if (s_cnt[i]) atomicAdd((unsigned long long*)&part_counts[i], s_cnt[i]);
  }
}

__global__ void k_shard_offsets(const int64_t* __restrict__ part_counts, int n_parts,
                                int64_t* __restrict__ cursors) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int64_t acc = 0;
    for (int i = 0; i < n_parts; ++i) {
      cursors[i] = acc;
      acc += part_counts[i];
    }
  }
}

// Rows are appended run by run with ONE global atomic per (workgroup chunk, part): the chunk's
// rows are counted per part in LDS, the workgroup reserves a contiguous range of every run,
// and each row then takes its position from an LDS cursor.  (A global atomic per row on
// n_parts <= 8 words would serialise: ~12 ns per same-address atomic.)
constexpr int kShardChunk = 16;  // entries per lane per chunk
__global__ __launch_bounds__(kBlock) void k_shard_scatter(DevPlan p, int idx_target_as_key,
                                                           const int64_t* __restrict__ buf,
                                                           int n_parts,
                                                           int64_t* __restrict__ out_rows,
                                                           int64_t* __restrict__ cursors) {
  __shared__ unsigned int s_cnt[256];
  __shared__ unsigned long long s_base[256];
  const int64_t chunk = (int64_t)kBlock * kShardChunk;
  for (int64_t c0 = (int64_t)blockIdx.x * chunk; c0 < p.entry_count; c0 += (int64_t)gridDim.x * chunk) {
    for (int i = threadIdx.x; i < n_parts; i += kBlock) s_cnt[i] = 0;
    __syncthreads();
    uint32_t part[kShardChunk];
#pragma unroll
    for (int k = 0; k < kShardChunk; ++k) {
      const int64_t e = c0 + (int64_t)k * kBlock + threadIdx.x;
      part[k] = 0xffffffffu;
      if (e < p.entry_count) {
        const int64_t* row = buf + e * p.row_quad;
        if (!is_empty_row(p, row, idx_target_as_key)) {
          part[k] = shard_of(p, row, n_parts);
          atomicAdd(&s_cnt[part[k]], 1u);
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_parts; i += kBlock) {
      const unsigned int c = s_cnt[i];
      s_base[i] = c ? atomicAdd((unsigned long long*)&cursors[i], (unsigned long long)c) : 0ull;
      s_cnt[i] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kShardChunk; ++k) {
      if (part[k] == 0xffffffffu) continue;
      const int64_t e = c0 + (int64_t)k * kBlock + threadIdx.x;
      const int64_t* row = buf + e * p.row_quad;
      const int64_t dst = (int64_t)(s_base[part[k]] + atomicAdd(&s_cnt[part[k]], 1u));
      int64_t* o = out_rows + dst * p.row_quad;
      for (int j = 0; j < p.row_quad; ++j) o[j] = row[j];
    }
    __syncthreads();
  }
}

// ---- join hash table builds -------------------------------------------------------------
// OneToOne perfect: slot[key - min] = row id under CAS(-1 -> id)
// (fill_one_to_one_hashtable JoinHashImpl.h:43-52; fill_hash_join_buff HashJoinRuntime.cpp:203).
__global__ __launch_bounds__(kBlock) void k_join_fill_perfect(const int8_t* __restrict__ keys,
                                                               int type, int nullable, int64_t n,
                                                               int64_t min_key, int64_t max_key,
                                                               int32_t* __restrict__ buf,
                                                               int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t null_t = int_null_of(type);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    if (*(volatile int32_t*)d_err) break;  // the attempt has already failed (e.g. a duplicate key)
    const int64_t k = decode_int(keys, type, i);
    if (nullable && k == null_t) continue;
    if (k < min_key || k > max_key) {
      atomicCAS(d_err, 0, MI355Q_ERR_INVALID_PLAN);
      continue;
    }
    if (atomicCAS((int*)&buf[k - min_key], -1, (int)i) != -1) {
      // one report is enough: ten million same-address atomics would take 100 ms
      if (*(volatile int32_t*)d_err == 0) atomicCAS(d_err, 0, MI355Q_ERR_JOIN_NOT_ONE_TO_ONE);
    }
  }
}

// ---- composite / one-to-many join tables -------------------------------------------------
MQ_D bool load_join_key(const JoinKeyCols& kc, int64_t i, int64_t* keys) {
  bool ok = true;
  for (int k = 0; k < kc.n; ++k) {
    keys[k] = decode_int(kc.col[k], kc.type[k], i);
    // rows with a NULL key component are not inserted (GenericKeyHandler skips them,
    // fill_hash_join_buff_impl :203-216)
    ok = ok && !(kc.nullable[k] && keys[k] == int_null_of(kc.type[k]));
  }
  return ok;
}

// Insert-or-find of a key of n_keys components of type T in a keyed table (slot stride in
// components).  One component: the reference's CAS EMPTY -> key (write_baseline_hash_slot,
// HashJoinRuntime.cpp:505-538).  Several: the first component is the write lock (EMPTY ->
// EMPTY - 1 -> value) so a reader never sees half a key; see baseline_find_or_insert_multi.
template <typename T, typename U>
MQ_D int64_t keyed_insert(T* tab, uint32_t entries, int n_keys, int stride, const int64_t* keys,
                          T empty, int32_t* bad) {
  uint32_t words[2 * MI355Q_MAX_GROUP_COLS];
  const int n_words = pack_join_key(keys, n_keys, (int)sizeof(T), words);
  const T locked = empty - 1;
  if ((T)keys[0] == empty || (n_keys > 1 && (T)keys[0] == locked)) {
    *bad = 1;
    return -1;
  }
  const uint32_t h = murmur1_words(words, n_words) % entries;
  uint32_t hp = h;
  int64_t found = -1;
  bool done = false;
  int spins = 0;
  while (!done) {
    T* e = tab + (size_t)hp * stride;
    bool advance = false;
    const T old = (T)atomicCAS((U*)e, (U)empty, (U)(n_keys > 1 ? locked : (T)keys[0]));
    if (old == empty) {
      if (n_keys > 1) {
        for (int i = 1; i < n_keys; ++i) __hip_atomic_store(e + i, (T)keys[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        MQ_PUBLISH_ORDER();
        __hip_atomic_store(e, (T)keys[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      found = hp;
      done = true;
    } else if (n_keys > 1 && old == locked) {
      if (++spins > kMaxLockSpins) {
        *bad = 1;
        done = true;
      }
    } else if (old == (T)keys[0]) {
      bool same = true;
      if (n_keys > 1) {
        for (int i = 1; i < n_keys; ++i)
          same = same && __hip_atomic_load(e + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (T)keys[i];
      }
      if (same) {
        found = hp;
        done = true;
      } else {
        advance = true;
      }
    } else {
      advance = true;
    }
    if (advance) {
      hp = hp + 1 == entries ? 0 : hp + 1;
      if (hp == h) done = true;
    }
  }
  return found;
}

// keys EMPTY, payload component (one-to-one) -1 (init_baseline_hash_join_buff,
// HashJoinRuntime.cpp:346-373)
__global__ __launch_bounds__(kBlock) void k_join_init_keyed(void* __restrict__ tab, int64_t entries,
                                                             int n_keys, int stride, int width) {
  const int64_t total = entries * stride;
  const int64_t step = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += step) {
    const bool is_key = (int)(i % stride) < n_keys;
    if (width == 4) ((int32_t*)tab)[i] = is_key ? kEmptyKey32 : -1;
    else ((int64_t*)tab)[i] = is_key ? kEmptyKey64 : -1;
  }
}

__global__ __launch_bounds__(kBlock) void k_join_fill_keyed(JoinKeyCols kc, int64_t n, void* __restrict__ tab,
                                                             int64_t entries, int stride, int with_payload,
                                                             int32_t* __restrict__ d_err) {
  const int64_t step = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    if (*(volatile int32_t*)d_err) break;  // the attempt has already failed
    int64_t keys[MI355Q_MAX_GROUP_COLS];
    if (!load_join_key(kc, i, keys)) continue;
    int32_t bad = 0;
    int64_t slot;
    if (kc.width == 4) {
      slot = keyed_insert<int32_t, unsigned int>((int32_t*)tab, (uint32_t)entries, kc.n, stride, keys, kEmptyKey32, &bad);
    } else {
      slot = keyed_insert<int64_t, unsigned long long>((int64_t*)tab, (uint32_t)entries, kc.n, stride, keys,
                                                       kEmptyKey64, &bad);
    }
    if (slot < 0) {
      atomicCAS(d_err, 0, bad ? MI355Q_ERR_INVALID_PLAN : MI355Q_ERR_JOIN_TABLE_FULL);
      continue;
    }
    if (with_payload) {
      bool dup;
      if (kc.width == 4) {
        dup = atomicCAS((int*)tab + slot * stride + kc.n, -1, (int)i) != -1;
      } else {
        dup = atomicCAS((unsigned long long*)tab + slot * stride + kc.n, (unsigned long long)-1ll,
                        (unsigned long long)i) != (unsigned long long)-1ll;
      }
      if (dup && *(volatile int32_t*)d_err == 0) atomicCAS(d_err, 0, MI355Q_ERR_JOIN_NOT_ONE_TO_ONE);
    }
  }
}

// slot of an inner row's key in a built one-to-many table: perfect = key - min, keyed = the
// key's entry (read-only probe), -1 = not there
MQ_D int64_t one_to_many_slot(const JoinKeyCols& kc, int hash_type, const void* tab, int64_t entries,
                              int64_t min_key, int64_t max_key, const int64_t* keys) {
  if (hash_type == 2) return (keys[0] >= min_key && keys[0] <= max_key) ? keys[0] - min_key : -1;
  return keyed_slot_of(tab, (uint32_t)entries, kc.n, kc.width, kc.n, keys);
}

// count_matches (HashJoinRuntime.cpp:652-700)
__global__ __launch_bounds__(kBlock) void k_join_count(JoinKeyCols kc, int64_t n, int hash_type,
                                                        const void* __restrict__ tab, int64_t entries,
                                                        int64_t min_key, int64_t max_key,
                                                        int32_t* __restrict__ counts,
                                                        int32_t* __restrict__ d_err) {
  const int64_t step = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    int64_t keys[MI355Q_MAX_GROUP_COLS];
    if (!load_join_key(kc, i, keys)) continue;
    const int64_t slot = one_to_many_slot(kc, hash_type, tab, entries, min_key, max_key, keys);
    if (slot < 0) {
      atomicCAS(d_err, 0, MI355Q_ERR_INVALID_PLAN);  // key outside the declared range
      continue;
    }
    atomicAdd(&counts[slot], 1);
  }
}

// fill_row_ids (HashJoinRuntime.cpp:895-945): payloads[offsets[slot] + running count] = row id
__global__ __launch_bounds__(kBlock) void k_join_fill_ids(JoinKeyCols kc, int64_t n, int hash_type,
                                                           const void* __restrict__ tab, int64_t entries,
                                                           int64_t min_key, int64_t max_key,
                                                           const int32_t* __restrict__ offsets,
                                                           int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ payloads) {
  const int64_t step = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
    int64_t keys[MI355Q_MAX_GROUP_COLS];
    if (!load_join_key(kc, i, keys)) continue;
    const int64_t slot = one_to_many_slot(kc, hash_type, tab, entries, min_key, max_key, keys);
    if (slot < 0) continue;
    payloads[offsets[slot] + atomicAdd(&counts[slot], 1)] = (int32_t)i;
  }
}

// Exclusive scan of the per-slot counts into offsets, -1 where the count is zero (the
// reference: inclusive_scan of the counts shifted by one, then pos[i] = scan[i] only where
// count[i] != 0, HashJoinRuntime.cpp:1525-1548).  Three passes: tile sums, scan of the tile
// sums by one workgroup, tile-local scan + write.
constexpr int kScanTile = kBlock * 8;
__global__ __launch_bounds__(kBlock) void k_scan_tile_sums(const int32_t* __restrict__ counts, int64_t n,
                                                            int64_t* __restrict__ tile_sums) {
  __shared__ int64_t s_part[kBlock / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  int64_t local = 0;
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + (int64_t)k * kBlock + threadIdx.x;
    if (i < n) local += counts[i];
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t t = 0;
    for (int w = 0; w < kBlock / 64; ++w) t += s_part[w];
    tile_sums[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(kBlock) void k_scan_tile_offsets(int64_t* __restrict__ tile_sums, int64_t n_tiles) {
  // one workgroup: each thread owns a contiguous chunk of the tile sums
  __shared__ int64_t s_chunk[kBlock];
  const int64_t per = (n_tiles + kBlock - 1) / kBlock;
  const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
  int64_t sum = 0;
  for (int64_t i = lo; i < hi; ++i) sum += tile_sums[i];
  s_chunk[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t acc = 0;
    for (int t = 0; t < kBlock; ++t) {
      const int64_t v = s_chunk[t];
      s_chunk[t] = acc;
      acc += v;
    }
  }
  __syncthreads();
  int64_t acc = s_chunk[threadIdx.x];
  for (int64_t i = lo; i < hi; ++i) {
    const int64_t v = tile_sums[i];
    tile_sums[i] = acc;
    acc += v;
  }
}
__global__ __launch_bounds__(kBlock) void k_scan_write_offsets(const int32_t* __restrict__ counts, int64_t n,
                                                                const int64_t* __restrict__ tile_offsets,
                                                                int32_t* __restrict__ offsets) {
  // thread t owns 8 consecutive slots of the tile
  __shared__ int64_t s_thread[kBlock];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * 8;
  int32_t c[8];
  int64_t sum = 0;
  for (int k = 0; k < 8; ++k) {
    c[k] = base + k < n ? counts[base + k] : 0;
    sum += c[k];
  }
  s_thread[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t acc = tile_offsets[blockIdx.x];
    for (int t = 0; t < kBlock; ++t) {
      const int64_t v = s_thread[t];
      s_thread[t] = acc;
      acc += v;
    }
  }
  __syncthreads();
  int64_t acc = s_thread[threadIdx.x];
  for (int k = 0; k < 8; ++k) {
    if (base + k < n) offsets[base + k] = c[k] ? (int32_t)acc : -1;
    acc += c[k];
  }
}

// ---- ColumnarResults: dense per-target columns of the non-empty entries ------------------------
__global__ __launch_bounds__(kBlock) void k_mark_live(DevPlan p, int idx_target_as_key, const int64_t* __restrict__ buf,
                                                       int32_t* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < p.entry_count; e += stride) {
    flags[e] = is_empty_row(p, buf + e * p.row_quad, idx_target_as_key) ? 0 : 1;
  }
}

__global__ __launch_bounds__(kBlock) void k_columns_write(DevPlan p, ColumnarSpec cs, const int64_t* __restrict__ buf,
                                                           const int32_t* __restrict__ offsets,
                                                           int64_t* const* __restrict__ cols) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < p.entry_count; e += stride) {
    const int32_t dst = offsets[e];
    if (dst < 0) continue;
    const int64_t* row = buf + e * p.row_quad;
    for (int t = 0; t < p.n_targets; ++t) {
      const DevTarget& tg = p.targets[t];
      int64_t v;
      if (tg.agg == MI355Q_PROJECT_KEY && tg.slot < 0) {
        v = row_key_component(row, p.key_width, tg.key_idx);
      } else {
        const int64_t* s = row + p.key_quad;
        const int64_t raw = p.slot_width == 4 ? (int64_t)((const int32_t*)s)[tg.slot] : s[tg.slot];
        if (tg.agg == MI355Q_AVG) {  // pair_to_double
          const int64_t cnt = s[tg.slot + 1];
          const double sum = tg.arg_f32 ? (double)bits_flt((int32_t)raw) : tg.arg_fp ? bits_dbl(raw) : (double)raw;
          v = cnt == 0 ? kNullDoubleBits : dbl_bits(sum / (double)cnt);
        } else if (tg.arg_f32 && tg.agg != MI355Q_COUNT) {  // float result, widened
          const bool is_null = tg.skip_null && (int32_t)raw == (int32_t)cs.null_pat[t];
          v = is_null ? kNullDoubleBits : dbl_bits((double)bits_flt((int32_t)raw));
        } else {
          v = raw;  // integers and doubles already carry their inline NULL sentinel
        }
      }
      cols[t][dst] = v;
    }
  }
}

// ---- 8-byte -> 4-byte slots (compact COUNT(*)-only layouts) --------------------------------
__global__ __launch_bounds__(kBlock) void k_narrow_slots(const int64_t* __restrict__ wide, int wide_quad,
                                                          int key_quad, int slot_count, int narrow_quad,
                                                          int64_t entries, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < entries; e += stride) {
    narrow_row(wide + e * wide_quad, key_quad, slot_count, narrow_quad, out + e * narrow_quad);
  }
}

// ---- row-wise <-> columnar result buffers (output_columnar_) ---------------------------------
// one lane per entry: the column side is coalesced (consecutive entries), the row side walks
// whole rows (every fetched line is used by the loop over the row's quads)
__global__ __launch_bounds__(kBlock) void k_rows_to_columns(ColLayout L, const int64_t* __restrict__ rows,
                                                             int8_t* __restrict__ cols) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < L.entry_count; e += stride)
    entry_to_columns(L, rows + e * L.row_quad, cols, e);
}
__global__ __launch_bounds__(kBlock) void k_columns_to_rows(ColLayout L, const int8_t* __restrict__ cols,
                                                             int64_t* __restrict__ rows) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < L.entry_count; e += stride)
    entry_from_columns(L, cols, e, rows + e * L.row_quad);
}
// initColumnarGroups (QueryMemoryInitializer.cpp:713-780): every column filled with its init value
__global__ __launch_bounds__(kBlock) void k_init_columns(ColLayout L, RowInit init, int8_t* __restrict__ cols) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < L.entry_count; e += stride)
    entry_to_columns(L, init.quad, cols, e);
}

// ---- packed multi-column keys --------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack_keys(PackSpec ps, const int8_t* const* __restrict__ cols,
                                                       const int64_t* __restrict__ num_rows, int n_frags,
                                                       int n_cols, int64_t* const* __restrict__ packed,
                                                       int32_t* __restrict__ d_err) {
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  bool bad = false;
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * n_cols;
    int64_t* dst = packed[f];
    const int64_t n = num_rows[f];
    for (int64_t pos = gtid; pos < n; pos += gsize) {
      if (ps.raw_f32) {  // one FLOAT key: widened, as the table stores it (NULL = FLT_MIN widens like any value)
        __builtin_nontemporal_store(dbl_bits((double)*(const float*)(fc[ps.cols[0]] + pos * 4)), dst + pos);
        continue;
      }
      uint64_t code = 0;
      for (int g = 0; g < ps.n; ++g) {
        int64_t k = decode_int(fc[ps.cols[g]], ps.types[g], pos);
        uint64_t c;
        if (ps.mode >= 1) {  // perfect hash: the entry index
          if (ps.translate[g] && k == int_null_of(ps.types[g])) k = ps.null_key[g];
          c = (uint64_t)k - (uint64_t)ps.min[g];
          if (ps.bucket[g]) c /= (uint64_t)ps.bucket[g];  // get_group_value_fast's (key - min) / bucket
          bad = bad || k < ps.min[g] || c >= ps.card[g];
          code += c * (uint64_t)ps.mul[g];
          continue;
        }
        if (ps.nullable[g] && k == int_null_of(ps.types[g])) {
          c = ps.card[g] - 1;
        } else {
          c = (uint64_t)k - (uint64_t)ps.min[g];
          // a value outside the declared range cannot be packed: the caller re-runs the step
          // with the row kernel (which hashes any key)
          bad = bad || k < ps.min[g] || c >= ps.card[g] - (ps.nullable[g] ? 1 : 0);
        }
        code |= c << ps.shift[g];
      }
      __builtin_nontemporal_store((int64_t)code, dst + pos);
    }
  }
  if (bad) atomicCAS(d_err, 0, MI355Q_ERR_UNSUPPORTED);
}

__global__ __launch_bounds__(kBlock) void k_unpack_emit(PackSpec ps, DevPlan p, const int64_t* __restrict__ tmp,
                                                         int64_t tmp_entries, int64_t* __restrict__ out,
                                                         int32_t* __restrict__ d_err) {
  const int tmp_quad = 1 + p.slot_count;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < tmp_entries; e += stride) {
    const int64_t* src = tmp + e * tmp_quad;
    if (src[0] == kEmptyKey64) continue;
    const uint64_t code = (uint64_t)src[0];
    int64_t keys[MI355Q_MAX_GROUP_COLS];
    if (ps.raw_f32) keys[0] = (int64_t)code;
    for (int g = 0; g < ps.n && !ps.raw_f32; ++g) {
      const uint64_t c = (code >> ps.shift[g]) & ps.mask[g];
      keys[g] = (ps.nullable[g] && c == ps.card[g] - 1) ? int_null_of(ps.types[g]) : (int64_t)(c + (uint64_t)ps.min[g]);
    }
    // the packed keys are pairwise distinct and `out` is fresh: claim a row, never search
    int64_t* slots = baseline_insert_distinct_multi(out, (uint32_t)p.entry_count, p.row_quad, p.key_width, ps.n,
                                                    keys);
    if (!slots) {
      atomicCAS(d_err, 0, MI355Q_ERR_OUT_OF_SLOTS);
      continue;
    }
    // every packed key is distinct, so this lane is the only writer of the row's slots
    for (int j = 0; j < p.slot_count; ++j) slots[j] = src[1 + j];
  }
}

// perfect-hash layouts: the packed key IS the entry index, so the row is addressed directly; key
// quads receive the translated keys, projected-key slots the original values (NULL restored).
// The temp table is either a baseline table keyed by the index (mode 1) or itself index-aligned
// (mode 2: small tables aggregated by the LDS perfect-hash kernel).
__global__ __launch_bounds__(kBlock) void k_unpack_perfect(PackSpec ps, DevPlan p, const int64_t* __restrict__ tmp,
                                                            int64_t tmp_entries, int tmp_quad, int tmp_key_quad,
                                                            int64_t* __restrict__ out, int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < tmp_entries; e += stride) {
    const int64_t* src = tmp + e * tmp_quad;
    int64_t idx;
    if (ps.mode == 2) {
      const bool empty = tmp_key_quad ? src[0] == kEmptyKey64 : src[ps.tmp_idx_target] == ps.tmp_init;
      if (empty) continue;
      idx = e;
    } else {
      if (src[0] == kEmptyKey64) continue;
      idx = src[0];
    }
    if (idx < 0 || idx >= p.entry_count) {
      atomicCAS(d_err, 0, MI355Q_ERR_OUT_OF_SLOTS);
      continue;
    }
    int64_t tk[MI355Q_MAX_GROUP_COLS], orig[MI355Q_MAX_GROUP_COLS];
    for (int g = 0; g < ps.n; ++g) {
      const int64_t d = (idx / ps.mul[g]) % (int64_t)ps.card[g];
      tk[g] = d * (ps.bucket[g] ? ps.bucket[g] : 1) + ps.min[g];
      orig[g] = (ps.translate[g] && tk[g] == ps.null_key[g]) ? int_null_of(ps.types[g]) : tk[g];
    }
    int64_t* row = out + idx * p.row_quad;
    for (int g = 0; g < p.key_quad; ++g) row[g] = tk[g];
    int64_t* slots = row + p.key_quad;
    for (int j = 0; j < p.slot_count; ++j) {
      const int sj = ps.slot_src[j];
      slots[j] = sj >= 0 ? src[tmp_key_quad + sj] : orig[-(sj + 1)];
    }
  }
}

// ---- projected expressions (scan / filter / PROJECT) --------------------------------------
// Every expression of the plan is evaluated once per row into a dense temporary column of its result type;
// the step then runs on the lowered plan (plan.cpp lower_exprs), in which those columns are ordinary inputs, so
// every kernel family applies to group keys / aggregate arguments / quals that are expressions.  `cols` is the
// EXTENDED fragment table of the pass: [frag][n_cols physical + n expressions], the expression columns being
// the outputs.  The reference's row function evaluates target and group-by expressions only for rows that
// passed the quals (and found a match under an INNER join), so an overflow only counts there; an expression
// used by a qual is evaluated for every row (Executor::compileBody: filters first, then the body).
constexpr int kProgQuads = MI355Q_MAX_EXPRS * MI355Q_MAX_EXPR_NODES * (int)sizeof(XNode) / 8;  // the programs in LDS, 8-byte words
constexpr int kProjJ = 4;  // rows of a lane evaluated together: a node is decoded once per 4 x 64 rows (expr.h eval_expr_rows)
__global__ __launch_bounds__(kBlock) void k_project(const DevExprSet* __restrict__ xsp, DevPlan p, uint32_t qual_expr_mask,
                                                     const int8_t* const* __restrict__ cols,
                                                     const int64_t* __restrict__ num_rows, int n_frags, int stack_below,
                                                     int32_t* __restrict__ d_err) {
  extern __shared__ __attribute__((aligned(16))) int64_t s_proj_stack[];  // the programs, then stack_below x kProjJ x kBlock values
  const DevExprSet& xs = *xsp;  // (device memory: 8 programs of 24 nodes do not fit the 4 KB of kernel arguments)
  const int nc = xs.n_cols + xs.n;
  constexpr int kTile = kProjJ * kBlock;
  // LDS: the programs as XNodes (handler + literal), then the stack
  XNode* const s_prog = (XNode*)s_proj_stack;
  for (int w = threadIdx.x; w < xs.n * MI355Q_MAX_EXPR_NODES; w += kBlock) {
    const DevExprNode& n = xs.e[w / MI355Q_MAX_EXPR_NODES].nodes[w % MI355Q_MAX_EXPR_NODES];
    XNode x;
    x.h = n.flags >> kExHandlerShift;
    x.pad_ = 0;
    x.lit = n.ilit;
    s_prog[w] = x;
  }
  __syncthreads();
  ExLdsStack stk;
  stk.st = s_proj_stack + kProgQuads;
  stk.tid = threadIdx.x;
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * nc;
    const int64_t n = num_rows[f];
#pragma unroll 1
    for (int64_t base = (int64_t)blockIdx.x * kTile; base < n; base += (int64_t)gridDim.x * kTile) {
      int64_t pos[kProjJ];
      bool live[kProjJ];
#pragma unroll
      for (int j = 0; j < kProjJ; ++j) {
        const int64_t r = base + j * kBlock + threadIdx.x;
        live[j] = r < n;
        pos[j] = live[j] ? r : n - 1;  // (a lane past the end evaluates the last row again and keeps nothing of it)
      }
      // every plain physical column the programs read, all rows of the tile, in one batch of independent loads
      int64_t raw[kExPre][kProjJ];
#pragma unroll
      for (int c = 0; c < kExPre; ++c) {
        if (c < xs.n_pre) {
          const int8_t* col = fc[xs.pre_col[c]];
          if (xs.pre_code[c] == MI355Q_INT32) {
#pragma unroll
            for (int j = 0; j < kProjJ; ++j) raw[c][j] = (int64_t)*(const int32_t*)(col + pos[j] * 4);
          } else {  // INT64 / DOUBLE: the 8 bytes as they are
#pragma unroll
            for (int j = 0; j < kProjJ; ++j) raw[c][j] = *(const int64_t*)(col + pos[j] * 8);
          }
        } else {
#pragma unroll
          for (int j = 0; j < kProjJ; ++j) raw[c][j] = 0;
        }
      }
      uint32_t err_mask[kProjJ] = {};
      int32_t first_err[kProjJ] = {}, first_qual_err[kProjJ] = {};  // the code of the first expression that failed (7 / 1)
      bool any_err = false;
#pragma unroll 1
      for (int k = 0; k < xs.n; ++k) {
        int64_t v[kProjJ];
        int32_t err[kProjJ];
        if ((xs.noerr_mask >> k) & 1) eval_expr_rows<kProjJ, kBlock, false>(xs.e[k], s_prog + k * MI355Q_MAX_EXPR_NODES, fc, pos, raw, stk, v, err);
        else eval_expr_rows<kProjJ, kBlock, true>(xs.e[k], s_prog + k * MI355Q_MAX_EXPR_NODES, fc, pos, raw, stk, v, err);
        int8_t* dst = const_cast<int8_t*>(fc[xs.n_cols + k]);
#pragma unroll
        for (int j = 0; j < kProjJ; ++j) {
          if (!live[j]) continue;
          store_expr_value(dst, xs.e[k], pos[j], v[j]);
          if (err[j]) {
            any_err = true;
            err_mask[j] |= 1u << k;
            if (!first_err[j]) first_err[j] = err[j];
            if (!first_qual_err[j] && ((qual_expr_mask >> k) & 1u)) first_qual_err[j] = err[j];
          }
        }
      }
      if (!any_err) continue;
      for (int j = 0; j < kProjJ; ++j) {  // rare: does the row count?
        if (!err_mask[j]) continue;
        bool counts = (err_mask[j] & qual_expr_mask) != 0;
        if (!counts) {
          counts = quals_pass(p, fc, pos[j]);
          if (counts && p.join_col >= 0 && p.join_kind != MI355Q_JOIN_LEFT) {
            int64_t jk[MI355Q_MAX_GROUP_COLS];
            bool null_key = false;
            for (int i = 0; i < p.join_n_keys; ++i) {
              jk[i] = decode_int(fc[p.join_cols[i]], p.join_types[i], pos[j]);
              null_key = null_key || (p.join_nullables[i] && jk[i] == int_null_of(p.join_types[i]));
            }
            counts = !null_key && join_lookup(p, jk).count > 0;
          }
        }
        if (counts) atomicCAS(d_err, 0, first_qual_err[j] ? first_qual_err[j] : first_err[j]);
      }
    }
  }
}

// ---- a grouped step over fact JOIN dim (one-to-one table) as a step WITHOUT a join (api.cpp execute_join_gather): one pass
// probes the join table per outer row and lays down, as dense temporary outer columns, the inner columns the aggregates
// read (the column's NULL where the row has no match: LEFT joins) and a 0 / 1 INT32 "matched" column (INNER joins filter on it) —
// the reference's join loop body reads exactly these values through the matched row id (IRCodegen.cpp buildJoinLoops,
// ColumnIR.cpp codegenOuterJoinNullPlaceholder).
struct JoinGather {
  int32_t n_inner, flag_col, nc2, pad_;
  int32_t inner_col[MI355Q_MAX_COLS], dst_col[MI355Q_MAX_COLS], width[MI355Q_MAX_COLS];
  int64_t null_pat[MI355Q_MAX_COLS];
};
__global__ __launch_bounds__(kBlock) void k_join_gather(DevPlan p, JoinGather jg, const int8_t* const* __restrict__ cols,
                                                         const int64_t* __restrict__ num_rows, int n_frags) {
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * jg.nc2;
    const int64_t n = num_rows[f];
    for (int64_t pos = gtid; pos < n; pos += gsize) {
      int64_t jk[MI355Q_MAX_GROUP_COLS];
      bool null_key = false;
      for (int i = 0; i < p.join_n_keys; ++i) {
        jk[i] = decode_int(fc[p.join_cols[i]], p.join_types[i], pos);
        null_key = null_key || (p.join_nullables[i] && jk[i] == int_null_of(p.join_types[i]));
      }
      int64_t inner_pos = -1;
      if (!null_key) {  // a NULL key matches nothing (hash_join_idx_nullable)
        const JoinMatch jm = join_lookup(p, jk);
        if (jm.count > 0) inner_pos = jm.single;
      }
      if (jg.flag_col >= 0) *(int32_t*)const_cast<int8_t*>(fc[jg.flag_col] + pos * 4) = inner_pos >= 0 ? 1 : 0;
      for (int j = 0; j < jg.n_inner; ++j) {
        const int8_t* src = p.inner_cols[jg.inner_col[j]];
        int8_t* dst = const_cast<int8_t*>(fc[jg.dst_col[j]]);
        switch (jg.width[j]) {
          case 1: *(int8_t*)(dst + pos) = inner_pos >= 0 ? *(const int8_t*)(src + inner_pos) : (int8_t)jg.null_pat[j]; break;
          case 2: *(int16_t*)(dst + pos * 2) = inner_pos >= 0 ? *(const int16_t*)(src + inner_pos * 2) : (int16_t)jg.null_pat[j]; break;
          case 4: *(int32_t*)(dst + pos * 4) = inner_pos >= 0 ? *(const int32_t*)(src + inner_pos * 4) : (int32_t)jg.null_pat[j]; break;
          default: *(int64_t*)(dst + pos * 8) = inner_pos >= 0 ? *(const int64_t*)(src + inner_pos * 8) : jg.null_pat[j];
        }
      }
    }
  }
}

// ---- projected expressions of ONE operation over ONE plain column, four rows per lane -------------------------
// k_project interprets a node list per row, one 4- or 8-byte load and store per lane, and its by-value expression
// set is indexed with run-time values (so it lives in scratch): 1.4 TB/s on `CAST(x AS DOUBLE)` — 8.7 ms of the
// 10.6 ms of BaselineHash/BH001 at 1 B rows (profiles/r04_refbench_lds_shapes_1b_call2.jsonl).  The reference
// benchmark's expressions are all of two shapes, and so are most GROUP BY / aggregate arguments in practice:
//   CAST(col AS DOUBLE | FLOAT)         (cast_<int>_to_<fp>_nullable, RuntimeFunctions.cpp:262-330)
//   col + | - | * literal               (ArithmeticIR.cpp:39-75, overflow check :861-909)
// over a plain INT or BIGINT column.  These run here: 16-byte loads and stores, roles compiled in.  An overflow only
// raises d_err[2]; the caller then runs k_project, which knows whether the offending row counts (quals, inner join).
struct SimpleProj {
  int32_t src_col, dst_col;  // positions in the per-fragment column table
  int32_t nullable;          // the operand may hold the type's inline NULL
  int32_t pad_;
  int64_t lit;
};
enum : int { SP_TO_F64 = 0, SP_TO_F32 = 1, SP_ADD = 2, SP_SUB = 3, SP_MUL = 4 };

template <typename ST, int KIND>
MQ_D auto simple_proj_one(ST v, bool nullable, int64_t lit, bool& ovf) {
  constexpr ST kNull = std::is_same<ST, int32_t>::value ? (ST)INT32_MIN : (ST)INT64_MIN;
  const bool is_null = nullable && v == kNull;
  if constexpr (KIND == SP_TO_F64) {
    return is_null ? kNullDoubleBits : dbl_bits((double)v);
  } else if constexpr (KIND == SP_TO_F32) {
    return is_null ? kNullFloatBits : flt_bits((float)v);
  } else {
    if (is_null) return kNull;
    long long w;
    bool o;
    if constexpr (KIND == SP_ADD) o = __builtin_add_overflow((long long)v, (long long)lit, &w);
    else if constexpr (KIND == SP_SUB) o = __builtin_sub_overflow((long long)v, (long long)lit, &w);
    else o = __builtin_mul_overflow((long long)v, (long long)lit, &w);
    if constexpr (std::is_same<ST, int32_t>::value) o = w > (long long)INT32_MAX || w < (long long)INT32_MIN;
    ovf = ovf || o;
    return (ST)w;
  }
}

template <typename ST, int KIND>
__global__ __launch_bounds__(kBlock) void k_project_simple(SimpleProj sp, const int8_t* const* __restrict__ cols,
                                                            const int64_t* __restrict__ num_rows, int n_frags, int nc,
                                                            int32_t* __restrict__ d_err) {
  using RT = decltype(simple_proj_one<ST, KIND>(ST{}, false, 0, *(bool*)nullptr));
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  const bool nullable = sp.nullable != 0;
  bool ovf = false;
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * nc;
    const ST* __restrict__ src = (const ST*)fc[sp.src_col];
    RT* __restrict__ dst = (RT*)const_cast<int8_t*>(fc[sp.dst_col]);
    const int64_t n = num_rows[f];
    const int64_t nq = n >> 2;
    struct alignas(16) SQ { ST v[4]; };
    struct alignas(16) DQ { RT v[4]; };
    for (int64_t q = gtid; q < nq; q += 2 * gsize) {
      const int64_t q2 = q + gsize;
      const SQ a = ((const SQ*)src)[q];
      SQ b = a;
      if (q2 < nq) b = ((const SQ*)src)[q2];
      DQ ra, rb;
#pragma unroll
      for (int i = 0; i < 4; ++i) ra.v[i] = simple_proj_one<ST, KIND>(a.v[i], nullable, sp.lit, ovf);
      ((DQ*)dst)[q] = ra;
      if (q2 < nq) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rb.v[i] = simple_proj_one<ST, KIND>(b.v[i], nullable, sp.lit, ovf);
        ((DQ*)dst)[q2] = rb;
      }
    }
    const int64_t tail = (nq << 2) + gtid;
    if (tail < n) dst[tail] = simple_proj_one<ST, KIND>(src[tail], nullable, sp.lit, ovf);
  }
  if (ovf) atomicExch(d_err + 2, 1);
}

// ---- several value columns through the single-value families --------------------------------
// A grouped step whose aggregates read two or three different columns (the reference's MultiStep benchmark:
// max(x100), max(x10), max(x10 + 1), sum(x100), sum(x10 + 1) per group) is run once per VALUE column through the
// single-value kernel families — same keys, same quals, so every run finds the same groups — and the runs' tables
// are zipped into the final layout: one lane per entry of a run's table locates the group's row in the final
// table (same entry index for perfect-hash layouts; the reference's insert-or-find for baseline ones) and copies
// the run's slots to where the final layout keeps them.
// ---- GROUP BY CAST(<integer column> AS DOUBLE | FLOAT): the step ran on the integer column (a perfect-hash layout, `sub`);
// every live entry is re-keyed with the cast value — the bit pattern of the double the key widens to (castToTypeIn(group_key,
// 64), IRCodegen.cpp:1505-1507; NULL -> the NULL of the cast's type, cast_<int>_to_<fp>_nullable) — and merged into the
// baseline-hash table of the stated plan with the reduce rule (two integers that cast to one FLOAT become one group).
struct CastKeyArgs {
  int32_t idx_key_s;      // keyless integer-keyed table: the target whose slot tells an empty entry
  int32_t cast_to_float;  // the cast's type: FLOAT (else DOUBLE)
  int32_t translate;      // the integer column is nullable: its NULL sits at `null_key` (max + 1)
  int32_t reserved;
  int64_t key_min, null_key;
};
__global__ __launch_bounds__(kBlock) void k_cast_key_emit(DevPlan pf, DevPlan ps, CastKeyArgs ck,
                                                           const int64_t* __restrict__ sub, int64_t* __restrict__ fin,
                                                           int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < ps.entry_count; e += stride) {
    const int64_t* row_s = sub + e * ps.row_quad;
    if (is_empty_row(ps, row_s, ck.idx_key_s)) continue;
    const int64_t* slots_s = row_s + ps.key_quad;
    // the integer key of the entry: index -> translated key (single column: mul = 1); the translated NULL key is the NULL
    const int64_t tk = e + ck.key_min;
    const bool is_null = ck.translate && tk == ck.null_key;
    int64_t key;
    if (ck.cast_to_float) key = dbl_bits((double)(is_null ? kNullFloat : (float)tk));
    else key = is_null ? kNullDoubleBits : dbl_bits((double)tk);
    int64_t* slots_f = baseline_find_or_insert(fin, (uint32_t)pf.entry_count, pf.row_quad, pf.key_width, key);
    if (!slots_f) {
      atomicCAS(d_err, 0, -1);  // out of group slots: the caller grows the table and retries
      continue;
    }
    for (int i = 0; i < pf.n_targets; ++i) {
      const DevTarget& tf = pf.targets[i];
      const DevTarget& ts = ps.targets[i];
      if (tf.slot < 0) continue;
      int64_t win[2];
      if (tf.agg == MI355Q_PROJECT_KEY) {
        win[0] = key;
        win[1] = 0;
      } else {
        if (ts.slot < 0) continue;
        win[0] = slots_s[ts.slot];
        win[1] = tf.agg == MI355Q_AVG ? slots_s[ts.slot + 1] : 0;
      }
      DevTarget lt = tf;
      lt.slot = 0;
      // (a DOUBLE cast is injective on these keys and the table was initialised for this emission: the row belongs to this lane
      // alone — plain read-modify-write, round 6; 10 M entries x 7 slots of global atomics were 3.1 ms; two integers may share a FLOAT)
      if (ck.cast_to_float) reduce_target<true>(lt, pf.init_vals + tf.slot, slots_f + tf.slot, win);
      else reduce_target<false>(lt, pf.init_vals + tf.slot, slots_f + tf.slot, win);
    }
  }
}

// ---- a baseline step whose integer key columns all have ranges ran as a PERFECT hash over the product of the ranges
// (api.cpp execute_perfect_twin): every live entry's keys are rebuilt from its index — entry = sum_i (key_i - min_i) *
// prod_{j<i} card_j, a translated NULL key back to the column's NULL — and the entry is merged into the baseline table
// of the stated plan with the reduce rule.
struct TwinArgs {
  int32_t n_keys, idx_key_s;
  int32_t translate[MI355Q_MAX_GROUP_COLS], key_type[MI355Q_MAX_GROUP_COLS];
  int64_t key_min[MI355Q_MAX_GROUP_COLS], key_card[MI355Q_MAX_GROUP_COLS], null_key[MI355Q_MAX_GROUP_COLS];
};
__global__ __launch_bounds__(kBlock) void k_perfect_twin_emit(DevPlan pf, DevPlan ps, TwinArgs ta,
                                                               const int64_t* __restrict__ sub, int64_t* __restrict__ fin,
                                                               int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < ps.entry_count; e += stride) {
    const int64_t* row_s = sub + e * ps.row_quad;
    if (is_empty_row(ps, row_s, ta.idx_key_s)) continue;
    const int64_t* slots_s = row_s + ps.key_quad;
    int64_t keys[MI355Q_MAX_GROUP_COLS] = {0, 0, 0, 0};
    int64_t rem = e;
    for (int g = 0; g < ta.n_keys; ++g) {
      const int64_t tk = ta.key_min[g] + rem % ta.key_card[g];
      rem /= ta.key_card[g];
      keys[g] = ta.translate[g] && tk == ta.null_key[g] ? int_null_of(ta.key_type[g]) : tk;
    }
    int64_t* slots_f;
    if (ta.n_keys == 1) {
      slots_f = baseline_find_or_insert(fin, (uint32_t)pf.entry_count, pf.row_quad, pf.key_width, keys[0]);
    } else {
      bool bad = false;
      slots_f = baseline_find_or_insert_multi(fin, (uint32_t)pf.entry_count, pf.row_quad, pf.key_width, ta.n_keys, keys, &bad);
      if (bad) {
        atomicCAS(d_err, 0, MI355Q_ERR_INVALID_PLAN);
        continue;
      }
    }
    if (!slots_f) {
      atomicCAS(d_err, 0, -1);  // out of group slots: the caller grows the table and retries
      continue;
    }
    for (int i = 0; i < pf.n_targets; ++i) {
      const DevTarget& tf = pf.targets[i];
      const DevTarget& ts = ps.targets[i];
      if (tf.slot < 0 || ts.slot < 0 || tf.agg == MI355Q_PROJECT_KEY) continue;  // (baseline: projections read the key columns)
      int64_t win[2];
      win[0] = slots_s[ts.slot];
      win[1] = tf.agg == MI355Q_AVG ? slots_s[ts.slot + 1] : 0;
      DevTarget lt = tf;
      lt.slot = 0;
      // (distinct entries have distinct keys and the table was initialised for this emission: the row is this lane's alone)
      reduce_target<false>(lt, pf.init_vals + tf.slot, slots_f + tf.slot, win);
    }
  }
}

// ---- baseline steps whose integer keys lie on a LATTICE (key = min + stride x i: the reference benchmark's BIGINT columns
// x10k_s10k ... hold multiples of 10 000, no range the planner could index) — api.cpp execute_affine_twin.  k_key_gcd finds
// the stride of a key column on the first fragment; k_affine_keys writes i as a dense INT32 column and VERIFIES every row of
// every fragment (a key off the lattice or outside the range raises the flag: the route is given up, nothing is guessed);
// the step runs grouped by those INT32 columns on a perfect-hash twin; k_affine_twin_emit re-keys its entries.
MQ_D uint64_t gcd_u64(uint64_t a, uint64_t b) {
  while (b) {
    const uint64_t t = a % b;
    a = b;
    b = t;
  }
  return a;
}
// (two stages without barriers or shuffles: one partial per lane, then 256 lanes fold 256 partials each; the host folds the rest)
__global__ __launch_bounds__(kBlock) void k_key_gcd(const int8_t* __restrict__ col, int width, int64_t n, int64_t kmin, int nullable,
                                                     unsigned long long* __restrict__ per_lane) {
  uint64_t g = 0;
  const int64_t null_v = width == 4 ? (int64_t)INT32_MIN : INT64_MIN;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int64_t v = width == 4 ? (int64_t)((const int32_t*)col)[i] : ((const int64_t*)col)[i];
    if (nullable && v == null_v) continue;
    const uint64_t d = (uint64_t)v - (uint64_t)kmin;
    if (d == 0) continue;
    g = g == 0 ? d : (d % g == 0 ? g : gcd_u64(g, d));
  }
  per_lane[(size_t)blockIdx.x * kBlock + threadIdx.x] = g;
}
__global__ __launch_bounds__(kBlock) void k_key_gcd_fold(const unsigned long long* __restrict__ per_lane, int n_per_lane,
                                                          unsigned long long* __restrict__ out) {
  uint64_t g = 0;
  for (int i = threadIdx.x; i < n_per_lane; i += kBlock) {
    const uint64_t o = per_lane[i];
    g = g == 0 ? o : (o == 0 ? g : gcd_u64(g, o));
  }
  out[threadIdx.x] = g;
}

struct AffineKeys {
  int32_t n, nc2, pad_[2];
  int32_t src_col[MI355Q_MAX_GROUP_COLS], dst_col[MI355Q_MAX_GROUP_COLS], width[MI355Q_MAX_GROUP_COLS], nullable[MI355Q_MAX_GROUP_COLS];
  int64_t kmin[MI355Q_MAX_GROUP_COLS], stride[MI355Q_MAX_GROUP_COLS], card[MI355Q_MAX_GROUP_COLS];  // card: lattice points
};
__global__ __launch_bounds__(kBlock) void k_affine_keys(AffineKeys ak, const int8_t* const* __restrict__ cols,
                                                         const int64_t* __restrict__ num_rows, int n_frags, int32_t* __restrict__ d_flag) {
  const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kBlock;
  bool off_lattice = false;
  for (int f = 0; f < n_frags; ++f) {
    const int8_t* const* fc = cols + (size_t)f * ak.nc2;
    const int64_t n = num_rows[f];
    for (int64_t pos = gtid; pos < n; pos += gsize) {
      for (int k = 0; k < ak.n; ++k) {
        const int8_t* src = fc[ak.src_col[k]];
        const int64_t v = ak.width[k] == 4 ? (int64_t)((const int32_t*)src)[pos] : ((const int64_t*)src)[pos];
        const int64_t null_v = ak.width[k] == 4 ? (int64_t)INT32_MIN : INT64_MIN;
        int32_t out;
        if (ak.nullable[k] && v == null_v) {
          out = INT32_MIN;
        } else {
          const uint64_t d = (uint64_t)v - (uint64_t)ak.kmin[k];
          const uint64_t q = d / (uint64_t)ak.stride[k];
          if (v < ak.kmin[k] || q * (uint64_t)ak.stride[k] != d || q >= (uint64_t)ak.card[k]) off_lattice = true;
          out = (int32_t)(uint32_t)q;
        }
        ((int32_t*)const_cast<int8_t*>(fc[ak.dst_col[k]]))[pos] = out;
      }
    }
  }
  if (off_lattice) atomicExch(d_flag, 1);  // (rare: the route is given up)
}

struct AffineTwinArgs {
  int32_t n_keys, idx_key_s;
  int32_t translate[MI355Q_MAX_GROUP_COLS], key_type[MI355Q_MAX_GROUP_COLS];
  // twin side: the key column's minimum, its cardinality (NULL slot included) and the translated NULL key; stated side: the
  // key of twin value t is base + (t - twin_min) x stride
  int64_t twin_min[MI355Q_MAX_GROUP_COLS], twin_card[MI355Q_MAX_GROUP_COLS], twin_null[MI355Q_MAX_GROUP_COLS];
  int64_t base[MI355Q_MAX_GROUP_COLS], stride[MI355Q_MAX_GROUP_COLS];
};
__global__ __launch_bounds__(kBlock) void k_affine_twin_emit(DevPlan pf, DevPlan ps, AffineTwinArgs ta,
                                                              const int64_t* __restrict__ sub, int64_t* __restrict__ fin,
                                                              int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < ps.entry_count; e += stride) {
    const int64_t* row_s = sub + e * ps.row_quad;
    if (is_empty_row(ps, row_s, ta.idx_key_s)) continue;
    const int64_t* slots_s = row_s + ps.key_quad;
    int64_t keys[MI355Q_MAX_GROUP_COLS] = {0, 0, 0, 0};
    int64_t rem = e;
    for (int g = 0; g < ta.n_keys; ++g) {
      const int64_t tk = ta.twin_min[g] + rem % ta.twin_card[g];
      rem /= ta.twin_card[g];
      keys[g] = ta.translate[g] && tk == ta.twin_null[g] ? int_null_of(ta.key_type[g])
                                                          : ta.base[g] + (tk - ta.twin_min[g]) * ta.stride[g];
    }
    int64_t* slots_f;
    if (ta.n_keys == 1) {
      slots_f = baseline_find_or_insert(fin, (uint32_t)pf.entry_count, pf.row_quad, pf.key_width, keys[0]);
    } else {
      bool bad = false;
      slots_f = baseline_find_or_insert_multi(fin, (uint32_t)pf.entry_count, pf.row_quad, pf.key_width, ta.n_keys, keys, &bad);
      if (bad) {
        atomicCAS(d_err, 0, MI355Q_ERR_INVALID_PLAN);
        continue;
      }
    }
    if (!slots_f) {
      atomicCAS(d_err, 0, -1);  // out of group slots: the caller grows the table and retries
      continue;
    }
    for (int i = 0; i < pf.n_targets; ++i) {
      const DevTarget& tf = pf.targets[i];
      const DevTarget& ts = ps.targets[i];
      if (tf.slot < 0 || ts.slot < 0 || tf.agg == MI355Q_PROJECT_KEY) continue;  // (baseline: projections read the key columns)
      int64_t win[2];
      win[0] = slots_s[ts.slot];
      win[1] = tf.agg == MI355Q_AVG ? slots_s[ts.slot + 1] : 0;
      DevTarget lt = tf;
      lt.slot = 0;
      // (distinct entries have distinct keys and the table was initialised for this emission: the row is this lane's alone)
      reduce_target<false>(lt, pf.init_vals + tf.slot, slots_f + tf.slot, win);
    }
  }
}

struct ZipMap {
  int32_t n;                          // slot copies
  int32_t src[MI355Q_MAX_SLOTS], dst[MI355Q_MAX_SLOTS];  // slot index in the run's row -> slot index in the final row
  // aggregates of `column + literal` computed as aggregates of the column (api.cpp execute_shifted_args): where the
  // group has non-NULL rows (their count sits in the run's slot cnt_src) the copy adds the literal (kind 1: MIN / MAX)
  // or literal x count (kind 2: SUM, the sum half of AVG); kind 0 = plain copy
  int32_t kind[MI355Q_MAX_SLOTS], cnt_src[MI355Q_MAX_SLOTS];
  int64_t lit[MI355Q_MAX_SLOTS];
  // baseline -> baseline with the same keys, entry count and key width into an EMPTY table: an entry keeps its position
  // (any probe-consistent placement is a valid table, and the run's is one) — a sequential copy instead of one
  // find-or-insert per group
  int32_t positional;
};
__global__ __launch_bounds__(kBlock) void k_zip_targets(DevPlan pf, DevPlan ps, int idx_key_s, const int64_t* __restrict__ sub,
                                                         int64_t* __restrict__ fin, ZipMap zm, int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < ps.entry_count; e += stride) {
    const int64_t* row_s = sub + e * ps.row_quad;
    if (is_empty_row(ps, row_s, idx_key_s)) continue;
    const int64_t* slots_s = row_s + ps.key_quad;
    int64_t* slots_f;
    if (pf.desc_type == MI355Q_GROUP_BY_PERFECT_HASH) {
      int64_t* row_f = fin + e * pf.row_quad;   // same key columns and ranges: the same entry index
      if (!pf.keyless) {
        if (MQ_LOAD64(row_f) == kEmptyKey64) {
          // the translated keys of the entry: from the run's key quads when it kept them, else from the index
          int64_t rem = e;
          for (int g = pf.n_group - 1; g >= 0; --g) {
            int64_t tk;
            if (!ps.keyless) {
              tk = row_s[g];
            } else {
              const int64_t d = rem / pf.group_mul[g];
              rem -= d * pf.group_mul[g];
              tk = d * (pf.group_bucket[g] ? pf.group_bucket[g] : 1) + pf.group_min[g];
            }
            row_f[g] = tk;  // (entry e's row is this lane's alone: plain stores, no agent-scope write-through)
          }
        }
        slots_f = row_f + pf.n_group;
      } else {
        slots_f = row_f;
      }
    } else if (zm.positional) {
      int64_t* row_f = fin + e * pf.row_quad;
      for (int k = 0; k < pf.key_quad; ++k) row_f[k] = row_s[k];
      slots_f = row_f + pf.key_quad;
    } else {
      int64_t keys[MI355Q_MAX_GROUP_COLS];
      for (int g = 0; g < pf.n_group; ++g) keys[g] = row_key_component(row_s, ps.key_width, g);
      if (pf.n_group == 1) {
        slots_f = baseline_find_or_insert(fin, (uint32_t)pf.entry_count, pf.row_quad, pf.key_width, keys[0]);
      } else {
        bool bad = false;
        slots_f = baseline_find_or_insert_multi(fin, (uint32_t)pf.entry_count, pf.row_quad, pf.key_width, pf.n_group, keys, &bad);
        if (bad) {
          atomicCAS(d_err, 0, MI355Q_ERR_INVALID_PLAN);
          continue;
        }
      }
      if (!slots_f) {
        atomicCAS(d_err, 0, -1);
        continue;
      }
    }
    for (int j = 0; j < zm.n; ++j) {
      int64_t v = slots_s[zm.src[j]];
      if (zm.kind[j]) {
        const int64_t c = slots_s[zm.cnt_src[j]];
        if (c > 0) v = (int64_t)((uint64_t)v + (uint64_t)zm.lit[j] * (zm.kind[j] == 2 ? (uint64_t)c : 1ull));
      }
      slots_f[zm.dst[j]] = v;  // (the row belongs to this lane — its own entry, or the slot it just claimed)
    }
  }
}

// ---- synthetic columns ------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_generate(void* __restrict__ dst, int64_t n_rows,
                                                      int64_t row_offset, int kind, uint64_t seed,
                                                      int64_t a, int64_t b, int64_t c, double a_f,
                                                      int null_every) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_rows; i += stride) {
    const uint64_t row = (uint64_t)(row_offset + i);
    const uint64_t u = splitmix64(seed ^ (row * 0x9E3779B97F4A7C15ull));
    const bool is_null = null_every > 0 && (u >> 40) % (uint64_t)null_every == 0;
    switch (kind) {
      case MI355Q_GEN_I32_UNIFORM31:
        ((int32_t*)dst)[i] = is_null ? INT32_MIN : (int32_t)(u >> 33);
        break;
      case MI355Q_GEN_I32_MOD:
        ((int32_t*)dst)[i] = is_null ? INT32_MIN : (int32_t)((int64_t)(u % (uint64_t)a) + b);
        break;
      case MI355Q_GEN_I64_MOD:
        ((int64_t*)dst)[i] = is_null ? INT64_MIN : (int64_t)(u % (uint64_t)a) + b;
        break;
      case MI355Q_GEN_I64_MOD_MUL:
        ((int64_t*)dst)[i] = is_null ? INT64_MIN : (int64_t)(u % (uint64_t)a) * b + c;
        break;
      default:
        ((double*)dst)[i] = is_null ? kNullDouble : (double)(u >> 11) * 0x1.0p-53 * a_f;
    }
  }
}

inline int grid_for(int64_t work_items, int max_blocks = 2048) {
  int64_t b = (work_items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

}  // namespace

hipError_t launch_words_to_host(const int32_t* d_src, int32_t* h_dst_dev, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_words_to_host, dim3(1), dim3(64), 0, s, d_src, h_dst_dev, n);
  return hipGetLastError();
}

static thread_local TuneKnobs g_knobs;
const TuneKnobs& tune_knobs() { return g_knobs; }
void set_tune_knobs(const TuneKnobs& k) { g_knobs = k; }
static thread_local const BoolFilter* t_step_bf = nullptr;
static thread_local const BoolFilter* t_step_bf_dev = nullptr;
const BoolFilter* step_bool_filter() { return t_step_bf; }
const BoolFilter* step_bool_filter_dev() { return t_step_bf_dev; }
void set_step_bool_filter(const BoolFilter* bf, const BoolFilter* dev) {
  t_step_bf = bf;
  t_step_bf_dev = dev;
}

hipError_t launch_init_buffer(int64_t* buf, int64_t entry_count, const RowInit& init,
                              hipStream_t s) {
  const int64_t quads = entry_count * init.row_quad;
  hipLaunchKernelGGL(k_init_buffer, dim3(grid_for(quads)), dim3(kBlock), 0, s, buf, quads, init);
  return hipGetLastError();
}

hipError_t launch_generic(const DevPlan& p, int idx_target_as_key, const RowInit& init,
                          const int8_t* const* d_cols, const int64_t* d_num_rows, int n_frags,
                          int64_t max_frag_rows, int64_t* out, int32_t* d_err, int n_cus,
                          hipStream_t s) {
  const int64_t tab_bytes = p.entry_count * p.row_quad * 8;
  if (p.desc_type == MI355Q_GROUP_BY_PERFECT_HASH && tab_bytes > 0 && tab_bytes <= 64 * 1024) {
    // two workgroups per CU keep their private tables in LDS
    const int grid = grid_for(max_frag_rows, n_cus * 2);
    (void)hipFuncSetAttribute((const void*)k_generic_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)tab_bytes);
    hipLaunchKernelGGL(k_generic_lds, dim3(grid), dim3(kBlock), (size_t)tab_bytes, s, p, idx_target_as_key,
                       init, d_cols, d_num_rows, n_frags, out, d_err);
    return hipGetLastError();
  }
  const int grid = grid_for(max_frag_rows, n_cus * 8);
  hipLaunchKernelGGL(k_generic, dim3(grid), dim3(kBlock), 0, s, p, d_cols, d_num_rows, n_frags,
                     out, d_err);
  return hipGetLastError();
}

hipError_t launch_reduce(const DevPlan& p, int idx_target_as_key, int64_t* this_buf,
                         const int64_t* that_rows, int64_t that_entries, int32_t* d_err,
                         hipStream_t s) {
  if (that_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_reduce, dim3(grid_for(that_entries)), dim3(kBlock), 0, s, p,
                     idx_target_as_key, this_buf, that_rows, that_entries, d_err);
  return hipGetLastError();
}

// ---- multi-device merge by home-slot slices (baseline tables with one int64 key).
// Rank r of `world` owns the keys whose HOME slot (MurmurHash3(key) % entry_count, GroupByRuntime.cpp:20-48)
// lies in [bound(r), bound(r + 1)), bound(r) = r * entry_count / world.  Because linear probing keeps a
// key at or shortly after its home slot, the rows rank r needs from any other rank are that rank's
// table rows [bound(r), bound(r + 1)) — a contiguous slice, sent as it is — plus the few rows of the
// probe cluster that runs past the slice's end (the `pad`: the next pad_rows rows, wrapping at the
// table's end).  No partition pass, no counts to exchange: every split size is known up front.
//   k_shard_pads    copies the pad after each slice and checks that it contains an empty slot (then no
//                   cluster that starts inside the slice can reach beyond it)
//   k_reduce_range  folds received rows into this rank's table, keeping only keys whose home slot is in
//                   the rank's range (a slice also holds strays of the PREVIOUS range's clusters, a pad
//                   holds rows of the NEXT range: both belong to somebody else)
__global__ __launch_bounds__(kBlock) void k_shard_pads(DevPlan p, const int64_t* __restrict__ buf, int world,
                                                        int pad_rows, int64_t* __restrict__ out_pads,
                                                        int32_t* __restrict__ ok) {
  const int r = blockIdx.x;  // one workgroup per boundary
  __shared__ int s_found;
  if (threadIdx.x == 0) s_found = 0;
  __syncthreads();
  const int64_t hi = (int64_t)(((__int128)(r + 1) * p.entry_count) / world);
  int found = 0;
  for (int i = threadIdx.x; i < pad_rows; i += kBlock) {
    int64_t e = hi + i;
    if (e >= p.entry_count) e -= p.entry_count;
    if (e >= p.entry_count) e %= p.entry_count;
    const int64_t* row = buf + e * p.row_quad;
    int64_t* dst = out_pads + ((int64_t)r * pad_rows + i) * p.row_quad;
    for (int j = 0; j < p.row_quad; ++j) dst[j] = row[j];
    found |= row[0] == kEmptyKey64;
  }
  if (found) atomicOr(&s_found, 1);
  __syncthreads();
  if (threadIdx.x == 0) ok[r] = s_found;
}

__global__ __launch_bounds__(kBlock) void k_reduce_range(DevPlan p, int idx_target_as_key, int64_t* __restrict__ this_buf,
                                                          const int64_t* __restrict__ that_rows, int64_t that_entries,
                                                          uint32_t home_lo, uint32_t home_hi, int32_t* __restrict__ d_err) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < that_entries; e += stride) {
    const int64_t* row = that_rows + e * p.row_quad;
    const int64_t key = row[0];
    if (key == kEmptyKey64) continue;
    const uint32_t home = murmur3_u64((uint64_t)key) % (uint32_t)p.entry_count;
    if (home < home_lo || home >= home_hi) continue;
    const int32_t err = reduce_entry<true>(p, idx_target_as_key, this_buf, row, e);
    if (err) atomicCAS(d_err, 0, err);
  }
}

hipError_t launch_shard_pads(const DevPlan& p, const int64_t* buf, int world, int pad_rows, int64_t* out_pads,
                             int32_t* d_ok, hipStream_t s) {
  if (world < 1 || pad_rows < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_shard_pads, dim3(world), dim3(kBlock), 0, s, p, buf, world, pad_rows, out_pads, d_ok);
  return hipGetLastError();
}

hipError_t launch_reduce_range(const DevPlan& p, int idx_target_as_key, int64_t* this_buf, const int64_t* that_rows,
                               int64_t that_entries, int64_t home_lo, int64_t home_hi, int32_t* d_err, hipStream_t s) {
  if (that_entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_reduce_range, dim3(grid_for(that_entries)), dim3(kBlock), 0, s, p, idx_target_as_key, this_buf,
                     that_rows, that_entries, (uint32_t)home_lo, (uint32_t)home_hi, d_err);
  return hipGetLastError();
}

hipError_t launch_count_nonempty(const DevPlan& p, int idx_target_as_key, const int64_t* buf,
                                 unsigned long long* d_count, hipStream_t s) {
  hipError_t e = hipMemsetAsync(d_count, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_count_nonempty, dim3(grid_for(p.entry_count)), dim3(kBlock), 0, s, p,
                     idx_target_as_key, buf, d_count);
  return hipGetLastError();
}

hipError_t launch_shard_partition(const DevPlan& p, int idx_target_as_key, const int64_t* buf,
                                  int n_parts, int64_t* out_rows, int64_t* d_part_counts,
                                  int64_t* d_cursors, hipStream_t s) {
  if (n_parts < 1 || n_parts > 256) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(d_part_counts, 0, sizeof(int64_t) * n_parts, s);
  if (e != hipSuccess) return e;
  const int grid = grid_for(p.entry_count);
  hipLaunchKernelGGL(k_shard_count, dim3(grid), dim3(kBlock), 0, s, p, idx_target_as_key, buf,
                     n_parts, d_part_counts);
  hipLaunchKernelGGL(k_shard_offsets, dim3(1), dim3(64), 0, s, d_part_counts, n_parts, d_cursors);
  hipLaunchKernelGGL(k_shard_scatter, dim3(grid), dim3(kBlock), 0, s, p, idx_target_as_key, buf,
                     n_parts, out_rows, d_cursors);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_join_presence_bitmap(const int32_t* __restrict__ table,
                                                                  int64_t entries,
                                                                  uint32_t* __restrict__ bitmap) {
  // one lane per slot, one 64-bit ballot per wave = two bitmap words
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t padded = (entries + 63) & ~(int64_t)63;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < padded; i += stride) {
    const bool live = i < entries && table[i] >= 0;
    const unsigned long long m = __ballot(live);
    if ((threadIdx.x & 63) == 0) {
      bitmap[i >> 5] = (uint32_t)m;
      if (i + 32 < entries) bitmap[(i >> 5) + 1] = (uint32_t)(m >> 32);
    }
  }
}

hipError_t launch_join_presence_bitmap(const int32_t* table, int64_t entries, uint32_t* bitmap,
                                       hipStream_t s) {
  if (entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_join_presence_bitmap, dim3(grid_for(entries)), dim3(kBlock), 0, s, table, entries,
                     bitmap);
  return hipGetLastError();
}

hipError_t launch_join_fill_perfect(const int8_t* keys, int type, int nullable, int64_t n,
                                    int64_t min_key, int64_t max_key, int32_t* buf,
                                    int32_t* d_err, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_join_fill_perfect, dim3(grid_for(n)), dim3(kBlock), 0, s, keys, type,
                     nullable, n, min_key, max_key, buf, d_err);
  return hipGetLastError();
}

hipError_t launch_join_init_keyed(void* tab, int64_t entries, int n_keys, int stride, int width,
                                  hipStream_t s) {
  hipLaunchKernelGGL(k_join_init_keyed, dim3(grid_for(entries * stride)), dim3(kBlock), 0, s, tab, entries,
                     n_keys, stride, width);
  return hipGetLastError();
}

hipError_t launch_join_fill_keyed(const JoinKeyCols& kc, int64_t n, void* tab, int64_t entries, int stride,
                                  bool with_payload, int32_t* d_err, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_join_fill_keyed, dim3(grid_for(n)), dim3(kBlock), 0, s, kc, n, tab, entries, stride,
                     with_payload ? 1 : 0, d_err);
  return hipGetLastError();
}

// offsets / counts / payloads of a one-to-many table whose key section (keyed tables) is
// already filled; tile_scratch holds (entries / 2048 + 1) int64
hipError_t launch_join_one_to_many(const JoinKeyCols& kc, int64_t n, int hash_type, const void* tab,
                                   int64_t entries, int64_t min_key, int64_t max_key, int32_t* offsets,
                                   int32_t* counts, int32_t* payloads, int64_t* tile_scratch,
                                   int32_t* d_err, hipStream_t s) {
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)entries, s);
  if (e != hipSuccess) return e;
  if (n > 0) {
    hipLaunchKernelGGL(k_join_count, dim3(grid_for(n)), dim3(kBlock), 0, s, kc, n, hash_type, tab, entries,
                       min_key, max_key, counts, d_err);
  }
  const int64_t n_tiles = (entries + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, counts, entries, tile_scratch);
  hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(kBlock), 0, s, tile_scratch, n_tiles);
  hipLaunchKernelGGL(k_scan_write_offsets, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, counts, entries,
                     tile_scratch, offsets);
  e = hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)entries, s);
  if (e != hipSuccess) return e;
  if (n > 0) {
    hipLaunchKernelGGL(k_join_fill_ids, dim3(grid_for(n)), dim3(kBlock), 0, s, kc, n, hash_type, tab, entries,
                       min_key, max_key, offsets, counts, payloads);
  }
  return hipGetLastError();
}

hipError_t launch_to_columns(const DevPlan& p, int idx_target_as_key, const ColumnarSpec& cs, const int64_t* buf,
                             int32_t* flags, int32_t* offsets, int64_t* tile_scratch, int64_t* const* cols,
                             hipStream_t s) {
  const int64_t n = p.entry_count;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_mark_live, dim3(grid_for(n)), dim3(kBlock), 0, s, p, idx_target_as_key, buf, flags);
  const int64_t n_tiles = (n + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, flags, n, tile_scratch);
  hipLaunchKernelGGL(k_scan_tile_offsets, dim3(1), dim3(kBlock), 0, s, tile_scratch, n_tiles);
  hipLaunchKernelGGL(k_scan_write_offsets, dim3((unsigned)n_tiles), dim3(kBlock), 0, s, flags, n, tile_scratch,
                     offsets);
  hipLaunchKernelGGL(k_columns_write, dim3(grid_for(n)), dim3(kBlock), 0, s, p, cs, buf, offsets, cols);
  return hipGetLastError();
}

hipError_t launch_narrow_slots(const int64_t* wide, int wide_quad, int key_quad, int slot_count,
                               int narrow_quad, int64_t entries, int64_t* out, hipStream_t s) {
  if (entries <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_narrow_slots, dim3(grid_for(entries)), dim3(kBlock), 0, s, wide, wide_quad, key_quad,
                     slot_count, narrow_quad, entries, out);
  return hipGetLastError();
}

hipError_t launch_rows_to_columns(const ColLayout& L, const int64_t* rows, void* cols, hipStream_t s) {
  if (L.entry_count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_rows_to_columns, dim3(grid_for(L.entry_count)), dim3(kBlock), 0, s, L, rows, (int8_t*)cols);
  return hipGetLastError();
}
hipError_t launch_columns_to_rows(const ColLayout& L, const void* cols, int64_t* rows, hipStream_t s) {
  if (L.entry_count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_columns_to_rows, dim3(grid_for(L.entry_count)), dim3(kBlock), 0, s, L, (const int8_t*)cols,
                     rows);
  return hipGetLastError();
}
hipError_t launch_init_columns(const ColLayout& L, const RowInit& init, void* cols, hipStream_t s) {
  if (L.entry_count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_init_columns, dim3(grid_for(L.entry_count)), dim3(kBlock), 0, s, L, init, (int8_t*)cols);
  return hipGetLastError();
}

hipError_t launch_pack_keys(const PackSpec& ps, const int8_t* const* d_cols, const int64_t* d_num_rows,
                            int n_frags, int n_cols, int64_t max_frag_rows, int64_t* const* packed_cols,
                            int32_t* d_err, int n_cus, hipStream_t s) {
  if (n_frags <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_keys, dim3(grid_for(max_frag_rows, n_cus * 8)), dim3(kBlock), 0, s, ps, d_cols,
                     d_num_rows, n_frags, n_cols, packed_cols, d_err);
  return hipGetLastError();
}

hipError_t launch_zip_targets(const DevPlan& pf, const DevPlan& ps, int idx_key_s, const int64_t* sub, int64_t* fin,
                              const int32_t* src, const int32_t* dst, int n, int32_t* d_err, hipStream_t s,
                              const int32_t* kind, const int32_t* cnt_src, const int64_t* lit, bool into_empty_table) {
  ZipMap zm{};
  zm.n = n;
  zm.positional = into_empty_table && pf.desc_type == MI355Q_GROUP_BY_BASELINE_HASH && ps.desc_type == pf.desc_type &&
                  pf.entry_count == ps.entry_count && pf.key_width == ps.key_width && pf.n_group == ps.n_group &&
                  pf.key_quad == ps.key_quad;
  for (int i = 0; i < n; ++i) {
    zm.src[i] = src[i];
    zm.dst[i] = dst[i];
    zm.kind[i] = kind ? kind[i] : 0;
    zm.cnt_src[i] = kind ? cnt_src[i] : 0;
    zm.lit[i] = kind ? lit[i] : 0;
  }
  if (ps.entry_count <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_zip_targets, dim3(grid_for(ps.entry_count)), dim3(kBlock), 0, s, pf, ps, idx_key_s, sub, fin, zm, d_err);
  return hipGetLastError();
}

namespace {
// CAST(plain INT / BIGINT column AS DOUBLE | FLOAT), or that column + - * a literal of its own type
bool simple_proj_shape(const DevExpr& e, int n_phys, SimpleProj* sp, int* src_type, int* kind) {
  if (e.n_nodes < 2 || e.n_nodes > 3) return false;
  const DevExprNode& c = e.nodes[0];
  if (c.op != MI355Q_EX_COL || (c.ilit != MI355Q_INT32 && c.ilit != MI355Q_INT64) || c.type != (int32_t)c.ilit) return false;
  if (c.arg >= n_phys) return false;  // (the value of an earlier expression: only k_project evaluates a row's expressions in order)
  *src_type = c.type;
  sp->src_col = c.arg;
  sp->nullable = (c.flags & EXF_NULLABLE) ? 1 : 0;
  sp->lit = 0;
  if (e.n_nodes == 2) {
    const DevExprNode& k = e.nodes[1];
    if (k.op != MI355Q_EX_CAST || k.arg != c.type) return false;
    if (k.type == MI355Q_DOUBLE) *kind = SP_TO_F64;
    else if (k.type == MI355Q_FLOAT) *kind = SP_TO_F32;
    else return false;
    return true;
  }
  const DevExprNode& l = e.nodes[1];
  const DevExprNode& o = e.nodes[2];
  if (l.op != MI355Q_EX_LIT || l.arg != 0 /* the NULL literal */ || l.type != c.type || o.type != c.type) return false;
  if (o.op == MI355Q_EX_ADD) *kind = SP_ADD;
  else if (o.op == MI355Q_EX_SUB) *kind = SP_SUB;
  else if (o.op == MI355Q_EX_MUL) *kind = SP_MUL;
  else return false;
  sp->lit = l.ilit;
  return true;
}
template <typename ST>
void launch_simple_kind(int kind, dim3 grid, hipStream_t s, const SimpleProj& sp, const int8_t* const* d_cols,
                        const int64_t* d_num_rows, int n_frags, int nc, int32_t* d_err) {
  switch (kind) {
    case SP_TO_F64: hipLaunchKernelGGL((k_project_simple<ST, SP_TO_F64>), grid, dim3(kBlock), 0, s, sp, d_cols, d_num_rows, n_frags, nc, d_err); break;
    case SP_TO_F32: hipLaunchKernelGGL((k_project_simple<ST, SP_TO_F32>), grid, dim3(kBlock), 0, s, sp, d_cols, d_num_rows, n_frags, nc, d_err); break;
    case SP_ADD: hipLaunchKernelGGL((k_project_simple<ST, SP_ADD>), grid, dim3(kBlock), 0, s, sp, d_cols, d_num_rows, n_frags, nc, d_err); break;
    case SP_SUB: hipLaunchKernelGGL((k_project_simple<ST, SP_SUB>), grid, dim3(kBlock), 0, s, sp, d_cols, d_num_rows, n_frags, nc, d_err); break;
    default: hipLaunchKernelGGL((k_project_simple<ST, SP_MUL>), grid, dim3(kBlock), 0, s, sp, d_cols, d_num_rows, n_frags, nc, d_err);
  }
}
}  // namespace

bool project_simple_shapes(const DevExprSet& xs) {
  for (int k = 0; k < xs.n; ++k) {
    SimpleProj sp{};
    int st = 0, kind = 0;
    if (!simple_proj_shape(xs.e[k], xs.n_cols, &sp, &st, &kind)) return false;
  }
  return xs.n > 0;
}

// `simple`: every expression is of the one-operation shape AND the caller found every source chunk 16-byte aligned —
// one k_project_simple launch per expression; an overflow raises d_err[2] and the caller comes back with simple = false
hipError_t launch_perfect_twin_emit(const DevPlan& pf, const DevPlan& ps, int idx_key_s, int n_keys, const int32_t* translate,
                                    const int32_t* key_type, const int64_t* key_min, const int64_t* key_card,
                                    const int64_t* null_key, const int64_t* sub, int64_t* fin, int32_t* d_err, hipStream_t s) {
  if (ps.entry_count <= 0) return hipSuccess;
  TwinArgs ta{};
  ta.n_keys = n_keys;
  ta.idx_key_s = idx_key_s;
  for (int g = 0; g < n_keys; ++g) {
    ta.translate[g] = translate[g];
    ta.key_type[g] = key_type[g];
    ta.key_min[g] = key_min[g];
    ta.key_card[g] = key_card[g];
    ta.null_key[g] = null_key[g];
  }
  hipLaunchKernelGGL(k_perfect_twin_emit, dim3(grid_for(ps.entry_count)), dim3(kBlock), 0, s, pf, ps, ta, sub, fin, d_err);
  return hipGetLastError();
}

hipError_t launch_key_gcd(const void* col, int width, int64_t n, int64_t kmin, int nullable, unsigned long long* scratch,
                          unsigned long long* out256, hipStream_t s) {
  constexpr int kBlocks = 64;  // scratch: kBlocks * kBlock partials
  hipLaunchKernelGGL(k_key_gcd, dim3(kBlocks), dim3(kBlock), 0, s, (const int8_t*)col, width, n, kmin, nullable, scratch);
  hipLaunchKernelGGL(k_key_gcd_fold, dim3(1), dim3(kBlock), 0, s, scratch, kBlocks * kBlock, out256);
  return hipGetLastError();
}

hipError_t launch_affine_keys(int n, const int32_t* src_col, const int32_t* dst_col, const int32_t* width, const int32_t* nullable,
                              const int64_t* kmin, const int64_t* stride, const int64_t* card, int nc2, const int8_t* const* d_cols,
                              const int64_t* d_num_rows, int n_frags, int64_t max_frag_rows, int32_t* d_flag, int n_cus, hipStream_t s) {
  if (n_frags <= 0 || n <= 0) return hipSuccess;
  AffineKeys ak{};
  ak.n = n;
  ak.nc2 = nc2;
  for (int k = 0; k < n; ++k) {
    ak.src_col[k] = src_col[k];
    ak.dst_col[k] = dst_col[k];
    ak.width[k] = width[k];
    ak.nullable[k] = nullable[k];
    ak.kmin[k] = kmin[k];
    ak.stride[k] = stride[k];
    ak.card[k] = card[k];
  }
  hipLaunchKernelGGL(k_affine_keys, dim3(grid_for(max_frag_rows, n_cus * 8)), dim3(kBlock), 0, s, ak, d_cols, d_num_rows, n_frags, d_flag);
  return hipGetLastError();
}

hipError_t launch_affine_twin_emit(const DevPlan& pf, const DevPlan& ps, int idx_key_s, int n_keys, const int32_t* translate,
                                   const int32_t* key_type, const int64_t* twin_min, const int64_t* twin_card, const int64_t* twin_null,
                                   const int64_t* base, const int64_t* stride, const int64_t* sub, int64_t* fin, int32_t* d_err,
                                   hipStream_t s) {
  if (ps.entry_count <= 0) return hipSuccess;
  AffineTwinArgs ta{};
  ta.n_keys = n_keys;
  ta.idx_key_s = idx_key_s;
  for (int g = 0; g < n_keys; ++g) {
    ta.translate[g] = translate[g];
    ta.key_type[g] = key_type[g];
    ta.twin_min[g] = twin_min[g];
    ta.twin_card[g] = twin_card[g];
    ta.twin_null[g] = twin_null[g];
    ta.base[g] = base[g];
    ta.stride[g] = stride[g];
  }
  hipLaunchKernelGGL(k_affine_twin_emit, dim3(grid_for(ps.entry_count)), dim3(kBlock), 0, s, pf, ps, ta, sub, fin, d_err);
  return hipGetLastError();
}

hipError_t launch_cast_key_emit(const DevPlan& pf, const DevPlan& ps, int idx_key_s, int cast_to_float, int translate,
                                int64_t key_min, int64_t null_key, const int64_t* sub, int64_t* fin, int32_t* d_err,
                                hipStream_t s) {
  if (ps.entry_count <= 0) return hipSuccess;
  const CastKeyArgs ck{idx_key_s, cast_to_float, translate, 0, key_min, null_key};
  hipLaunchKernelGGL(k_cast_key_emit, dim3(grid_for(ps.entry_count)), dim3(kBlock), 0, s, pf, ps, ck, sub, fin, d_err);
  return hipGetLastError();
}

hipError_t launch_project(const DevExprSet& xs, DevExprSet* d_xs_area, const DevPlan& p, uint32_t qual_expr_mask,
                          const int8_t* const* d_cols, const int64_t* d_num_rows, int n_frags, int64_t max_frag_rows, int32_t* d_err,
                          int n_cus, hipStream_t s, bool simple) {
  if (n_frags <= 0 || xs.n <= 0) return hipSuccess;
  if (simple) {
    const dim3 grid(grid_for((max_frag_rows + 7) / 8, n_cus * 8));
    for (int k = 0; k < xs.n; ++k) {
      SimpleProj sp{};
      int st = 0, kind = 0;
      if (!simple_proj_shape(xs.e[k], xs.n_cols, &sp, &st, &kind)) return hipErrorInvalidValue;
      sp.dst_col = xs.n_cols + k;
      if (st == MI355Q_INT32) launch_simple_kind<int32_t>(kind, grid, s, sp, d_cols, d_num_rows, n_frags, xs.n_cols + xs.n, d_err);
      else launch_simple_kind<int64_t>(kind, grid, s, sp, d_cols, d_num_rows, n_frags, xs.n_cols + xs.n, d_err);
    }
    return hipGetLastError();
  }
  // the device copy of the programs: handlers, literal patterns, the columns loaded per tile (expr.h xh_label_programs)
  static thread_local DevExprSet up;  // (the upload's source outlives the call: the caller synchronises the stream)
  up = xs;
  const int deepest = xh_label_programs(&up, true);
  hipError_t e = hipMemcpyAsync(d_xs_area, &up, sizeof(up), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  // LDS: the evaluation stack BELOW its top (the top stays in registers), sized for the deepest of the programs
  const int below = deepest - 1;
  const size_t lds = (size_t)kProgQuads * 8 + (size_t)below * kProjJ * kBlock * 8;
  const int64_t tiles = (max_frag_rows + kProjJ * kBlock - 1) / (kProjJ * kBlock);
  hipLaunchKernelGGL(k_project, dim3(grid_for(tiles * kBlock, n_cus * 8)), dim3(kBlock), lds, s, (const DevExprSet*)d_xs_area, p,
                     qual_expr_mask, d_cols, d_num_rows, n_frags, below, d_err);
  return hipGetLastError();
}

hipError_t launch_join_gather(const DevPlan& p, int n_inner, const int32_t* inner_col, const int32_t* dst_col, const int32_t* width,
                              const int64_t* null_pat, int flag_col, int nc2, const int8_t* const* d_cols, const int64_t* d_num_rows,
                              int n_frags, int64_t max_frag_rows, int n_cus, hipStream_t s) {
  if (n_frags <= 0) return hipSuccess;
  JoinGather jg{};
  jg.n_inner = n_inner;
  jg.flag_col = flag_col;
  jg.nc2 = nc2;
  for (int j = 0; j < n_inner; ++j) {
    jg.inner_col[j] = inner_col[j];
    jg.dst_col[j] = dst_col[j];
    jg.width[j] = width[j];
    jg.null_pat[j] = null_pat[j];
  }
  hipLaunchKernelGGL(k_join_gather, dim3(grid_for(max_frag_rows, n_cus * 8)), dim3(kBlock), 0, s, p, jg, d_cols, d_num_rows, n_frags);
  return hipGetLastError();
}

hipError_t launch_unpack_emit(const PackSpec& ps, const DevPlan& p, const int64_t* tmp, int64_t tmp_entries,
                              int tmp_quad, int tmp_key_quad, int64_t* out, int32_t* d_err, hipStream_t s) {
  if (tmp_entries <= 0) return hipSuccess;
  if (ps.mode >= 1) {
    hipLaunchKernelGGL(k_unpack_perfect, dim3(grid_for(tmp_entries)), dim3(kBlock), 0, s, ps, p, tmp, tmp_entries,
                       tmp_quad, tmp_key_quad, out, d_err);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_unpack_emit, dim3(grid_for(tmp_entries)), dim3(kBlock), 0, s, ps, p, tmp, tmp_entries, out,
                     d_err);
  return hipGetLastError();
}

hipError_t launch_generate(void* dst, int64_t n_rows, int64_t row_offset, int kind, uint64_t seed,
                           int64_t a, int64_t b, int64_t c, double a_f, int null_every,
                           hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_generate, dim3(grid_for(n_rows, 8192)), dim3(kBlock), 0, s, dst, n_rows,
                     row_offset, kind, seed, a, b, c, a_f, null_every);
  return hipGetLastError();
}

}  // namespace mq
