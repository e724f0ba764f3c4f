// kernels_part.hip — partition-then-aggregate GROUP BY for high-cardinality int64 keys
// (SURVEY cfg3 / cfg3-filtered, the headline workload).
//
// Why: with 10 M groups the output table (20 M x 32 B = 640 MB) is larger than every on-chip
// memory, so updating it row by row (the reference's GPU path: one CAS + atomics per row,
// cuda_mapd_rt.cu:180-246,436-545) turns every row into random HBM read-modify-writes.
// Instead the scan hash-partitions the surviving rows into P key ranges sized so that one
// range's groups fit a per-CU LDS hash table, and a second kernel aggregates each range
// entirely in LDS and emits each group ONCE into the HeavyDB-layout table:
//
//   phase 1  k_part_scatter    stream the columns (16 B/lane loads), filter, hash, append the
//                              16-byte record {key, value bits} to this block's private run
//                              of partition p.  Runs are block-private, so the cursors are LDS
//                              atomics and no global atomic is issued per row.  STAGED = true
//                              write-combines four records per partition in LDS and emits
//                              whole 64-byte lines.
//   phase 2  k_part_aggregate  one workgroup per partition: LDS open-addressing table
//                              {key, partial slots}, ds atomics per record, then one
//                              insert-or-find + atomics per GROUP into the output table.
//
// Robustness: a run that overflows its capacity, or an LDS table that fills up, sends the row
// straight to the output table with the same CAS + atomics the direct kernel uses, so skewed
// inputs degrade in speed, never in correctness.  Rows are processed in chunks of fragments
// so the scratch stays bounded.
#include "fast_common.h"

namespace mq {

using namespace fast;

namespace {

constexpr int kPartBlock = 1024;

struct alignas(16) Rec {
  int64_t key;
  int64_t val;
};

struct PartGeom {
  int32_t P, lgP, B;
  uint32_t cap;      // records per (block, partition) run
  uint32_t E;        // LDS table entries in phase 2
  int32_t ns_int;    // distinct partial slots kept per group in LDS
};

struct PartSlots {
  int32_t int_op[8];                 // op of each internal slot
  int32_t out_map[MI355Q_MAX_SLOTS]; // output slot -> internal slot
  int64_t int_init[8];               // identity of each internal slot
};

struct TableArgs {
  int64_t* out;
  uint32_t entry_count;
  int32_t row_quad;
  SlotProg sp;
};

MQ_D void spill_direct(const TableArgs& t, int64_t key, int64_t val_bits, int32_t* d_err,
                       unsigned long long* spills) {
  int64_t* slots = baseline_find_or_insert(t.out, t.entry_count, t.row_quad, 8, key);
  if (!slots) {
    atomicCAS(d_err, 0, -1);
    return;
  }
  apply_slots_global(t.sp, slots, bits_dbl(val_bits), val_bits);
  atomicAdd(spills, 1ull);
}

template <typename VT>
MQ_D int64_t val_bits_of(VT v);
template <>
MQ_D int64_t val_bits_of<int64_t>(int64_t v) { return v; }
template <>
MQ_D int64_t val_bits_of<double>(double v) { return dbl_bits(v); }
template <>
MQ_D int64_t val_bits_of<none_t>(none_t) { return 0; }

// ------------------------------------------------------------------------- phase 1
template <typename FT, typename VT, bool STAGED>
__global__ __launch_bounds__(kPartBlock) void k_part_scatter(
    const int8_t* const* __restrict__ cols, const int64_t* __restrict__ num_rows, int n_frags,
    int n_cols, RangeFilter flt, int kcol, int vcol, PartGeom g, Rec* __restrict__ scratch,
    uint32_t* __restrict__ cnt, TableArgs tab, int32_t* __restrict__ d_err,
    unsigned long long* __restrict__ spills) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* cursor = (uint32_t*)smem_raw;  // [P] records appended so far
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < g.P; i += kPartBlock) cursor[i] = 0;
  __syncthreads();
  const int sh = 32 - g.lgP;
  if (!STAGED) {
    scan_fragments<FT, int64_t, VT>(cols, num_rows, n_frags, n_cols, flt.col, kcol, vcol,
                                    [&](FT fv, int64_t key, VT val) {
      if (!filter_pass<FT>(flt, fv)) return;
      const uint32_t h = murmur3_u64((uint64_t)key);
      const uint32_t pid = g.lgP ? (h >> sh) : 0u;
      const uint32_t slot = atomicAdd(&cursor[pid], 1u);
      const int64_t vb = val_bits_of<VT>(val);
      if (slot < g.cap) {
        Rec r{key, vb};
        scratch[((size_t)pid * g.B + b) * g.cap + slot] = r;
      } else {
        spill_direct(tab, key, vb, d_err, spills);
      }
    });
  } else {
    // LDS write-combining: 4-record (64-byte) staging line per partition.  A record with
    // stream position `slot` may enter the line only while generation slot>>2 is the one
    // being filled; the lane that lands on position 3 queues the line for a cooperative
    // 64-byte flush after the barrier.
    Rec* stage = (Rec*)(smem_raw + (((size_t)g.P * 4 + 15) & ~(size_t)15));  // [P][4]
    uint32_t* flushed = (uint32_t*)(stage + (size_t)g.P * 4);                // [P] gens flushed
    uint32_t* full_list = flushed + g.P;                                     // [P]: <= 1 line per stream per round
    __shared__ uint32_t s_nfull;
    __shared__ uint32_t s_pending;
    for (int i = threadIdx.x; i < g.P; i += kPartBlock) flushed[i] = 0;
    if (threadIdx.x == 0) {
      s_nfull = 0;
      s_pending = 0;
    }
    __syncthreads();
    const int64_t gtid = (int64_t)blockIdx.x * kPartBlock + threadIdx.x;
    const int64_t gsize = (int64_t)gridDim.x * kPartBlock;
    for (int f = 0; f < n_frags; ++f) {
      const int8_t* const* fc = cols + (size_t)f * n_cols;
      const int8_t* fb = is_none<FT>::value ? nullptr : fc[flt.col];
      const int8_t* kb = fc[kcol];
      const int8_t* vb = is_none<VT>::value ? nullptr : fc[vcol];
      const int64_t n = num_rows[f];
      const int64_t nq = (n + 3) >> 2;
      // every lane walks the same number of tiles so the barriers stay uniform
      const int64_t tiles = (nq + gsize - 1) / gsize;
      for (int64_t t = 0; t < tiles; ++t) {
        const int64_t q = gtid + t * gsize;
        int64_t keys[4], vals[4];
        uint32_t pid[4], slot[4];
        uint32_t pend = 0;
        if (q < nq) {
          const int64_t r0 = q << 2;
          if (r0 + 4 <= n) {
            Quad<FT> f0;
            Quad<int64_t> k0;
            Quad<VT> v0;
            load_quad<FT>(fb, q, f0);
            load_quad<int64_t>(kb, q, k0);
            load_quad<VT>(vb, q, v0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (filter_pass<FT>(flt, quad_get(f0, i))) pend |= 1u << i;
              keys[i] = k0.v[i];
              vals[i] = val_bits_of<VT>(quad_get(v0, i));
            }
          } else {
            for (int i = 0; i < 4 && r0 + i < n; ++i) {
              if (filter_pass<FT>(flt, load_one<FT>(fb, r0 + i))) pend |= 1u << i;
              keys[i] = load_one<int64_t>(kb, r0 + i);
              vals[i] = val_bits_of<VT>(load_one<VT>(vb, r0 + i));
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (pend & (1u << i)) {
            const uint32_t h = murmur3_u64((uint64_t)keys[i]);
            pid[i] = g.lgP ? (h >> sh) : 0u;
            slot[i] = atomicAdd(&cursor[pid[i]], 1u);
            if (slot[i] >= g.cap) {  // run full: straight to the output table
              spill_direct(tab, keys[i], vals[i], d_err, spills);
              pend &= ~(1u << i);
            }
          }
        }
        for (;;) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if ((pend & (1u << i)) && (slot[i] >> 2) == *(volatile uint32_t*)&flushed[pid[i]]) {
              Rec r{keys[i], vals[i]};
              stage[(size_t)pid[i] * 4 + (slot[i] & 3)] = r;
              pend &= ~(1u << i);
              if ((slot[i] & 3) == 3) full_list[atomicAdd(&s_nfull, 1u)] = pid[i];
            }
          }
          if (pend) atomicOr(&s_pending, 1u);
          __syncthreads();
          const uint32_t nfull = s_nfull;
          const uint32_t more = s_pending;
          for (uint32_t j = threadIdx.x; j < nfull * 4; j += kPartBlock) {
            const uint32_t p = full_list[j >> 2];
            const uint32_t gen = flushed[p];
            scratch[((size_t)p * g.B + b) * g.cap + (size_t)gen * 4 + (j & 3)] = stage[(size_t)p * 4 + (j & 3)];
          }
          __syncthreads();
          for (uint32_t j = threadIdx.x; j < nfull; j += kPartBlock) flushed[full_list[j]] += 1;
          if (threadIdx.x == 0) {
            s_nfull = 0;
            s_pending = 0;
          }
          __syncthreads();
          if (!more) break;
        }
      }
    }
    // drain the partially filled lines
    for (int p = threadIdx.x; p < g.P; p += kPartBlock) {
      const uint32_t total = cursor[p] < g.cap ? cursor[p] : g.cap;
      const uint32_t done = flushed[p] * 4;
      for (uint32_t s = done; s < total; ++s) {
        scratch[((size_t)p * g.B + b) * g.cap + s] = stage[(size_t)p * 4 + (s & 3)];
      }
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < g.P; p += kPartBlock) {
    const uint32_t c = cursor[p];
    cnt[(size_t)p * g.B + b] = c < g.cap ? c : g.cap;
  }
}

// ------------------------------------------------------------------------- phase 2
MQ_D void lds_apply(int op, int64_t* s, int64_t vb) {
  switch (op) {
    case SO_COUNT: atomicAdd((unsigned long long*)s, 1ull); break;
    case SO_SUM_I: atomicAdd((unsigned long long*)s, (unsigned long long)vb); break;
    case SO_SUM_F: atomicAdd((double*)s, bits_dbl(vb)); break;
    case SO_MIN_I: atomicMin((long long*)s, (long long)vb); break;
    case SO_MAX_I: atomicMax((long long*)s, (long long)vb); break;
    case SO_MIN_F: a_minmax_f64<true, false, false>(s, bits_dbl(vb), 0.0); break;
    case SO_MAX_F: a_minmax_f64<true, true, false>(s, bits_dbl(vb), 0.0); break;
    default: break;
  }
}

MQ_D void global_merge(int op, int64_t* gslot, int64_t partial) {
  switch (op) {
    case SO_COUNT:
    case SO_SUM_I: atomicAdd((unsigned long long*)gslot, (unsigned long long)partial); break;
    case SO_SUM_F: atomicAdd((double*)gslot, bits_dbl(partial)); break;
    case SO_MIN_I: atomicMin((long long*)gslot, (long long)partial); break;
    case SO_MAX_I: atomicMax((long long*)gslot, (long long)partial); break;
    case SO_MIN_F: a_minmax_f64<true, false, false>(gslot, bits_dbl(partial), 0.0); break;
    case SO_MAX_F: a_minmax_f64<true, true, false>(gslot, bits_dbl(partial), 0.0); break;
    default: break;
  }
}

__global__ __launch_bounds__(kPartBlock) void k_part_aggregate(PartGeom g, const Rec* __restrict__ scratch,
                                                                const uint32_t* __restrict__ cnt,
                                                                PartSlots ps, TableArgs tab,
                                                                int32_t* __restrict__ d_err,
                                                                unsigned long long* __restrict__ spills) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int64_t* lkeys = (int64_t*)smem_raw;   // [E]
  int64_t* lslots = lkeys + g.E;         // [ns_int][E]  (slot-major: conflict-free strides)
  uint32_t* lcnt = (uint32_t*)(lslots + (size_t)g.ns_int * g.E);  // [B]
  const int ns = g.ns_int;
  auto insert_rec = [&](const Rec& r) {
    const uint32_t h = murmur3_u64((uint64_t)r.key);
    // the partition used the top lgP bits; the LDS slot uses the bits below them
    const uint32_t low = g.lgP ? (h << g.lgP) : h;
    uint32_t e = (uint32_t)(((uint64_t)low * g.E) >> 32);
    for (uint32_t probes = 0; probes < g.E; ++probes) {
      const int64_t old = (int64_t)atomicCAS((unsigned long long*)&lkeys[e],
                                             (unsigned long long)kEmptyKey64,
                                             (unsigned long long)r.key);
      if (old == kEmptyKey64 || old == r.key) {
        for (int j = 0; j < ns; ++j) lds_apply(ps.int_op[j], &lslots[(size_t)j * g.E + e], r.val);
        return;
      }
      e = e + 1 == g.E ? 0 : e + 1;
    }
    spill_direct(tab, r.key, r.val, d_err, spills);  // LDS table full
  };
  for (int p = blockIdx.x; p < g.P; p += gridDim.x) {
    for (uint32_t e = threadIdx.x; e < g.E; e += kPartBlock) {
      lkeys[e] = kEmptyKey64;
      for (int j = 0; j < ns; ++j) lslots[(size_t)j * g.E + e] = ps.int_init[j];
    }
    __syncthreads();
    // run lengths of this partition -> LDS (one coalesced read), then one wave per run with
    // four 16-byte record loads in flight per lane
    for (int b = threadIdx.x; b < g.B; b += kPartBlock) lcnt[b] = cnt[(size_t)p * g.B + b];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < g.B; b += kPartBlock / 64) {
      const uint32_t n = lcnt[b];
      const Rec* run = scratch + ((size_t)p * g.B + b) * g.cap;
      uint32_t i = lane;
      for (; i + 192 < n; i += 256) {
        const Rec r0 = run[i], r1 = run[i + 64], r2 = run[i + 128], r3 = run[i + 192];
        insert_rec(r0);
        insert_rec(r1);
        insert_rec(r2);
        insert_rec(r3);
      }
      for (; i < n; i += 64) insert_rec(run[i]);
    }
    __syncthreads();
    // emit: one insert-or-find per group, partial slots merged with atomics (a key may
    // already be there from an earlier chunk or from a spilled row)
    for (uint32_t e = threadIdx.x; e < g.E; e += kPartBlock) {
      const int64_t key = lkeys[e];
      if (key == kEmptyKey64) continue;
      int64_t* slots = baseline_find_or_insert(tab.out, tab.entry_count, tab.row_quad, 8, key);
      if (!slots) {
        atomicCAS(d_err, 0, -1);
        continue;
      }
      for (int j = 0; j < tab.sp.n; ++j) {
        const int m = ps.out_map[j];
        if (m >= 0) global_merge(tab.sp.op[j], slots + j, lslots[(size_t)m * g.E + e]);
      }
    }
    __syncthreads();
  }
}

uint32_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return (uint32_t)p;
}

int64_t op_identity(int op) {
  switch (op) {
    case SO_MIN_I: return INT64_MAX;
    case SO_MAX_I: return INT64_MIN;
    case SO_MIN_F: return dbl_bits(1.7976931348623157e308);
    case SO_MAX_F: return dbl_bits(-1.7976931348623157e308);
    default: return 0;
  }
}

struct PartPlanHost {
  PartGeom g;
  PartSlots ps;
  int64_t chunk_rows;       // max rows per chunk
  int64_t scratch_bytes;    // records + counts
  size_t lds1, lds2;
  bool staged;
};

constexpr size_t kLdsBudget = 140 * 1024;

bool make_part_plan(const DevPlan& p, const FastShape& fs, const FragView& fv, int n_cus,
                    int64_t scratch_cap, bool staged, PartPlanHost* out) {
  PartPlanHost& h = *out;
  h.staged = staged;
  // internal slots: distinct ops only (COUNT(*) and AVG's count share one LDS counter)
  int n_int = 0;
  for (int j = 0; j < MI355Q_MAX_SLOTS; ++j) h.ps.out_map[j] = -1;
  for (int j = 0; j < fs.sp.n; ++j) {
    const int op = fs.sp.op[j];
    if (op == SO_KEY) continue;
    int m = -1;
    for (int k = 0; k < n_int; ++k)
      if (h.ps.int_op[k] == op) m = k;
    if (m < 0) {
      if (n_int >= 8) return false;
      m = n_int++;
      h.ps.int_op[m] = op;
      h.ps.int_init[m] = op_identity(op);
    }
    h.ps.out_map[j] = m;
  }
  if (n_int == 0) return false;
  h.g.ns_int = n_int;
  const size_t entry_bytes = 8 * (size_t)(1 + n_int);
  const uint32_t e_max = (uint32_t)(kLdsBudget / entry_bytes);
  // expected groups: the caller sizes the table at ~2 x NDV (50 % fill, docs results.rst)
  const uint64_t groups = (uint64_t)(p.entry_count / 2 > 0 ? p.entry_count / 2 : 1);
  uint32_t P = next_pow2((groups + (uint64_t)(0.55 * e_max) - 1) / (uint64_t)(0.55 * e_max));
  const uint32_t p_max = staged ? 2048u : 8192u;
  if (P > p_max) P = p_max;
  if (debug_part_p() > 0) P = next_pow2((uint64_t)debug_part_p()) > p_max ? p_max : next_pow2((uint64_t)debug_part_p());
  h.g.P = (int32_t)P;
  h.g.lgP = 0;
  while ((1u << h.g.lgP) < P) ++h.g.lgP;
  uint64_t e_want = (groups / P) * 2 + 64;  // ~50 % fill
  if (e_want > e_max) e_want = e_max;
  if (e_want < 256) e_want = 256;
  h.g.E = (uint32_t)e_want;
  h.g.B = n_cus;  // one 1024-lane workgroup per CU
  // chunking: worst case every row survives the filter; shrink the chunk until the runs
  // (1.2 x mean + 6 sigma + slack per run) fit the scratch cap, never below one fragment
  int64_t chunk_rows = fv.total_rows;
  for (;;) {
    const double per_run = (double)chunk_rows / ((double)P * h.g.B);
    uint64_t cap = (uint64_t)(per_run * 1.2 + 6.0 * __builtin_sqrt(per_run + 1.0)) + 36;
    cap = (cap + 3) & ~3ull;  // whole 64-byte lines
    if (cap > 0x7fffffffull) return false;
    h.g.cap = (uint32_t)cap;
    h.scratch_bytes = (int64_t)P * h.g.B * (int64_t)cap * (int64_t)sizeof(Rec) + (int64_t)P * h.g.B * 4 + 256;
    if (h.scratch_bytes <= scratch_cap || chunk_rows <= fv.max_frag_rows) break;
    chunk_rows = (int64_t)(chunk_rows * 0.9);
    if (chunk_rows < fv.max_frag_rows) chunk_rows = fv.max_frag_rows;
  }
  h.chunk_rows = chunk_rows;
  h.lds1 = (size_t)P * 4;
  if (staged) {
    h.lds1 = (((size_t)P * 4 + 15) & ~(size_t)15) + (size_t)P * 4 * sizeof(Rec) + (size_t)P * 4 +
             (size_t)P * 4;
    if (h.lds1 > 158 * 1024) return false;
  }
  h.lds2 = (size_t)h.g.E * entry_bytes + (size_t)h.g.B * 4;
  return true;
}

template <typename FT, typename VT>
hipError_t launch_scatter_t(bool staged, int grid, size_t lds, hipStream_t s, const FragView& fv,
                            int f0, int nf, const RangeFilter& flt, int kcol, int vcol,
                            const PartGeom& g, Rec* scratch, uint32_t* cnt, const TableArgs& tab,
                            int32_t* d_err, unsigned long long* spills) {
  const int8_t* const* cols = fv.d_cols + (size_t)f0 * fv.n_cols;
  const int64_t* rows = fv.d_num_rows + f0;
  // opt in to > 64 KB of dynamic LDS (gfx950: 160 KB per workgroup)
  if (staged)
    (void)hipFuncSetAttribute((const void*)k_part_scatter<FT, VT, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (staged) {
    hipLaunchKernelGGL((k_part_scatter<FT, VT, true>), dim3(grid), dim3(kPartBlock), lds, s, cols, rows,
                       nf, fv.n_cols, flt, kcol, vcol, g, scratch, cnt, tab, d_err, spills);
  } else {
    hipLaunchKernelGGL((k_part_scatter<FT, VT, false>), dim3(grid), dim3(kPartBlock), lds, s, cols, rows,
                       nf, fv.n_cols, flt, kcol, vcol, g, scratch, cnt, tab, d_err, spills);
  }
  return hipGetLastError();
}

template <typename FT>
hipError_t launch_scatter_v(const FastShape& fs, bool staged, int grid, size_t lds, hipStream_t s,
                            const FragView& fv, int f0, int nf, int kcol, const PartGeom& g,
                            Rec* scratch, uint32_t* cnt, const TableArgs& tab, int32_t* d_err,
                            unsigned long long* spills) {
  const int vcol = fs.vcol < 0 ? 0 : fs.vcol;
  if (fs.vcol < 0)
    return launch_scatter_t<FT, none_t>(staged, grid, lds, s, fv, f0, nf, fs.flt, kcol, vcol, g, scratch, cnt, tab, d_err, spills);
  if (fs.vtype == MI355Q_INT64)
    return launch_scatter_t<FT, int64_t>(staged, grid, lds, s, fv, f0, nf, fs.flt, kcol, vcol, g, scratch, cnt, tab, d_err, spills);
  return launch_scatter_t<FT, double>(staged, grid, lds, s, fv, f0, nf, fs.flt, kcol, vcol, g, scratch, cnt, tab, d_err, spills);
}

}  // namespace

int64_t part_scratch_bytes(const DevPlan& p, const FragView& fv, int n_cus, int64_t cap_bytes,
                           bool staged) {
  FastShape fs;
  if (!grouped_fast_shape(p, fv, &fs)) return 0;
  if (cap_bytes <= 0) cap_bytes = (int64_t)32 << 30;
  PartPlanHost h;
  if (!make_part_plan(p, fs, fv, n_cus, cap_bytes, staged, &h)) return 0;
  return h.scratch_bytes + 64;
}

hipError_t launch_baseline_partitioned(const DevPlan& p, const FragView& fv, int64_t* out,
                                       int32_t* d_err, void* scratch, int64_t scratch_bytes,
                                       int64_t cap_bytes, bool staged, int n_cus, hipStream_t s,
                                       LaunchStats* st) {
  FastShape fs;
  if (!grouped_fast_shape(p, fv, &fs)) return hipErrorInvalidValue;
  if (cap_bytes <= 0) cap_bytes = (int64_t)32 << 30;
  PartPlanHost h;
  // same inputs as part_scratch_bytes -> the same plan
  if (!make_part_plan(p, fs, fv, n_cus, cap_bytes, staged, &h)) return hipErrorInvalidValue;
  if (h.scratch_bytes + 64 > scratch_bytes) return hipErrorInvalidValue;
  hipEvent_t* ev_pool = st->ev_pool;
  const int n_ev = st->n_ev;
  Rec* recs = (Rec*)scratch;
  const size_t rec_bytes = (size_t)h.g.P * h.g.B * h.g.cap * sizeof(Rec);
  uint32_t* cnt = (uint32_t*)((char*)scratch + rec_bytes);
  unsigned long long* spills = (unsigned long long*)((char*)scratch + ((h.scratch_bytes + 7) & ~7ll));
  hipError_t e = hipMemsetAsync(spills, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return e;
  TableArgs tab{out, (uint32_t)p.entry_count, p.row_quad, fs.sp};
  st->kernel_name = staged ? "k_part_scatter_staged" : "k_part_scatter";
  st->variant = staged ? 3 : 2;
  st->n_launches = 0;
  (void)hipFuncSetAttribute((const void*)k_part_aggregate, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h.lds2);
  int f = 0;
  int ev_i = 0;
  while (f < fv.n_frags) {
    int64_t rows = 0;
    int f1 = f;
    while (f1 < fv.n_frags && (f1 == f || rows + fv.h_num_rows[f1] <= h.chunk_rows)) {
      rows += fv.h_num_rows[f1];
      ++f1;
    }
    if (ev_pool && ev_i + 1 < n_ev) (void)hipEventRecord(ev_pool[ev_i], s);
    if (fs.fil_type == 0)
      e = launch_scatter_v<none_t>(fs, staged, h.g.B, h.lds1, s, fv, f, f1 - f, p.group_col, h.g, recs, cnt, tab, d_err, spills);
    else if (fs.fil_type == MI355Q_INT32)
      e = launch_scatter_v<int32_t>(fs, staged, h.g.B, h.lds1, s, fv, f, f1 - f, p.group_col, h.g, recs, cnt, tab, d_err, spills);
    else
      e = launch_scatter_v<int64_t>(fs, staged, h.g.B, h.lds1, s, fv, f, f1 - f, p.group_col, h.g, recs, cnt, tab, d_err, spills);
    if (e != hipSuccess) return e;
    if (ev_pool && ev_i + 1 < n_ev) {
      (void)hipEventRecord(ev_pool[ev_i + 1], s);
      ev_i += 2;
    }
    st->n_launches += 1;
    const int grid2 = h.g.P < n_cus ? h.g.P : n_cus;
    hipLaunchKernelGGL(k_part_aggregate, dim3(grid2), dim3(kPartBlock), h.lds2, s, h.g, recs, cnt,
                       h.ps, tab, d_err, spills);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    f = f1;
  }
  st->spill_counter = spills;
  st->n_events_used = ev_i;
  return hipSuccess;
}

}  // namespace mq
