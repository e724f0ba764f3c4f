// kernels_idx.hip — GROUP BY on a PERFECT-HASH layout that is too large for LDS: partition by entry index, then aggregate.
//
// The reference benchmark's mid- and high-cardinality perfect-hash shapes (Benchmarks/synthetic_benchmark/queries:
// PerfectHashSingleCol/PHS005-007 — x100k .. x10m; PerfectHashMultiCol/PHM004-006; MultiStep/MSPHS003-012, MSPHM003-007:
// one to three INT key columns, one to three INT value columns, 100 K - 10 M groups) used to borrow the baseline-hash
// family: k_pack_keys wrote the entry index as an int64 column (4 B read + 8 B written + 8 B read per row), the step ran on
// that column with 16-byte {kid, value} records, once PER VALUE COLUMN, the runs were zipped (k_zip_targets) and the
// temporary table re-emitted (k_unpack_perfect) — 40 B/row moved for 8 B/row of columns, 17 - 90 ms per 1 B rows
// (profiles/r04_refbench_1b_call3.jsonl: 0.02 - 0.08 of the roofline).  A perfect-hash table needs none of that: the entry
// index IS the group, so
//   phase 1  k_idx_scatter<NK, NV, RS>: reads the key and value columns as they are, computes the entry index e (32-bit:
//            sum_i (key_i - min_i) * mul_i, NULL keys translated; get_group_value_fast / perfect_key_hash,
//            GroupByRuntime.cpp:208-223, GroupByAndAggregate.cpp:1546-1598), partitions by e / S1 (a partition is a
//            contiguous slice of the table) and writes narrow records through the producer / flusher pipeline of
//            k_part_scatter (12 producer waves, 4 flusher waves, 128-byte staging lines, no workgroup barrier):
//              RS = 2   4-byte records {e : u32}                           NO value column (COUNT(*) / key projections only)
//              RS = 1   8-byte records {e : u32, v : i32}                 one value column
//              RS = 0  16-byte records {e, v0, v1, v2}                    two or three value columns, ONE exchange
//            PACKED records (PK, round 6): where the plan carries the value columns' ExpressionRanges (DevTarget::arg_rng —
//            the benchmark's y10 / x100 have 10 / 100 values) a record is ONE word: the entry's index INSIDE its partition
//            (the run names the partition) above the values' codes (value - min, 0 = NULL where the column is nullable):
//              RS = 3   2-byte records   bits(S1) + value bits <= 16   (PHS005 / 006: 9 .. 12 + 4 bits; Sort/S00x: index only)
//              RS = 2   4-byte records   ... <= 32                      (PHS007: 14 + 4; MSPHS005: 14 + 7 + 4 instead of 16 bytes)
//            the range is a HINT: a value outside it leaves as a full record through the spill list, so the result is exact
//            whatever the data holds.
//   phase 2  k_idx_aggregate<NV, MM, RS>: one workgroup per (partition, sub-range); the LDS table is indexed by e - lo (no
//            keys, no probing): rows (u32) and per value column non-NULL count (u32), sum (i64), min / max (i32) — the
//            typed LDS layout of kernels_lds.hip; every live entry is then merged into the (initialised) output table with
//            the reduce rule, so a later chunk simply merges again.
//   spill    records that meet a full run (skew) go to a list and are applied to the output table one by one afterwards
//            (k_idx_spill); if the list overflows the step is handed back to the packed route.
// Traffic per row with one value column: 8 B read + 8 B written + 8 B (x sub-ranges) read back.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "fast_common.h"
#include "lds_args.h"

namespace mq {

using namespace fast;

namespace {

constexpr int kIdxBlock = 1024;
constexpr int kIdxStageUnits = 8192;     // 16-byte units staged per workgroup (128 KB of LDS)
constexpr int kIdxSegUnits = 8;          // units per 128-byte segment (one lane each when it is flushed)
constexpr int kIdxProdWaves = 12;
constexpr int kIdxFlushWaves = kIdxBlock / 64 - kIdxProdWaves;
constexpr int kIdxSegPerFlusher = (kIdxStageUnits / kIdxSegUnits) / (kIdxFlushWaves * 64);  // 4
constexpr int kIdxMaxSub = 8;            // sub-ranges per partition (each one reads the partition's records again)
constexpr uint32_t kIdxMaxSpins = 1u << 20;
constexpr uint32_t kIdxSpillBlock = 256;
constexpr unsigned long long kIdxSpillBusy = ~0ull;
constexpr uint32_t kIdxSpillMin = 1u << 20;
constexpr size_t kIdxLdsTable = 150 * 1024;
constexpr size_t kIdxLdsTablePk = 158 * 1024;  // (packed records: PHS007's 9 766 entries x 16 bytes fit ONE sub-range)

struct IdxGeom {
  int32_t P, lgL, B;      // partitions, log2(units per staged line), scatter workgroups
  uint32_t L;             // units per line = kIdxStageUnits / P
  uint32_t cap;           // UNITS per (workgroup, partition) run, a multiple of L
  uint32_t d;             // entries of the table
  uint32_t S1, s1_rcp;    // entries per partition, floor(2^32 / S1)
  uint32_t R, S2;         // sub-ranges per partition, entries per sub-range (= LDS entries of a unit)
  int32_t nk, nv, mm, rs;
  uint32_t spill_cap;
  int32_t pk;             // packed records: (entry - partition's first entry) << vb | value codes
  uint32_t vb;            // bits of the value codes together
  int32_t mmode;          // packed records, phase 2: 0 no min / max; 1 min, 2 max, 3 both (codes); 4 a presence mask (<= 32 values)
};

// what both phases need of the plan (static indices after unrolling: never a scratch copy)
struct IdxCols {
  int32_t key_col[kLdsKeys], key_translate[kLdsKeys];
  uint32_t key_min[kLdsKeys], key_card[kLdsKeys], key_mul[kLdsKeys], key_null[kLdsKeys];
  int64_t key_min64[kLdsKeys], key_null64[kLdsKeys];
  int32_t val_col[kLdsVals], val_nullable[kLdsVals];
  int32_t target_v[MI355Q_MAX_TARGETS];
  // packed records: code = value - val_min (+ 1 where the column is nullable: 0 = NULL), val_card values, at bit val_shift
  int32_t val_min[kLdsVals];
  uint32_t val_card[kLdsVals], val_shift[kLdsVals], val_mask[kLdsVals];
};

struct IdxSpill {
  uint32_t* count;   // device word
  v4i32* entries;    // [cap] {e, v0, v1, v2}
  int32_t* d_err;    // [0] reference error code, [1] the family gives up (spill overflow / broken protocol)
  uint32_t cap;
};

MQ_D uint32_t idx_peek(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
MQ_D void idx_poke(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
MQ_D int32_t idx_get(const v4i32& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
MQ_D uint32_t idx_part_of(const IdxGeom& g, uint32_t e) {
  const uint32_t q = __umulhi(e, g.s1_rcp);
  return q + (e - q * g.S1 >= g.S1 ? 1u : 0u);
}

// packed records (the plan admits them for tables of <= 2^24 entries and partitions of > 256): partition AND the index
// inside it from full-rate 24-bit multiplies (v_mul_hi_u32_u24 / v_mul_u32_u24) instead of two quarter-rate 32-bit ones
MQ_D void idx_part_split24(const IdxGeom& g, uint32_t e, uint32_t& p, uint32_t& el) {
  uint32_t q = (uint32_t)(((uint64_t)(e & 0xffffffu) * (uint64_t)(g.s1_rcp & 0xffffffu)) >> 32);
  uint32_t rem = e - (q & 0xffffffu) * (g.S1 & 0xffffffu);  // (q < 2^16, S1 < 2^24)
  if (rem >= g.S1) {
    q += 1u;
    rem -= g.S1;
  }
  p = q;
  el = rem;
}

// spill positions are handed out from workgroup-private blocks (one global atomic per 256 entries)
MQ_D uint32_t idx_spill_slot(const IdxSpill& sl, unsigned long long* blk) {
  for (;;) {
    const unsigned long long w = __hip_atomic_load(blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (w == kIdxSpillBusy) continue;
    const uint32_t next = (uint32_t)w, end = (uint32_t)(w >> 32);
    if (next < end) {
      if (atomicCAS(blk, w, ((unsigned long long)end << 32) | (next + 1)) == w) return next;
    } else if (atomicCAS(blk, w, kIdxSpillBusy) == w) {
      const uint32_t base = atomicAdd(sl.count, kIdxSpillBlock);
      __hip_atomic_store(blk, ((unsigned long long)(base + kIdxSpillBlock) << 32) | (base + 1), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
      return base;
    }
  }
}
MQ_D void idx_spill(const IdxSpill& sl, unsigned long long* blk, const v4i32& rec) {
  const uint32_t i = idx_spill_slot(sl, blk);
  if (i >= sl.cap) {
    atomicExch(sl.d_err + 1, 1);
    return;
  }
  sl.entries[i] = rec;
}

// Cooperative flush of 128-byte segments (as flush_segments of kernels_part.hip): `need` marks the lanes whose own segment
// (index seg_base + lane) must leave; 8 lanes write one segment, 8 segments per wave instruction.
MQ_D void idx_flush_segments(bool need, uint32_t dst_unit, const v4i32* __restrict__ stage, int seg_base,
                             v4i32* __restrict__ scratch) {
  const uint64_t mask = __ballot(need);
  if (!mask) return;
  const int lane = threadIdx.x & 63;
  const int n_need = __popcll(mask);
  const int rank_need = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  const int dest = need ? rank_need : n_need + (lane - rank_need);
  const int owner_of_rank = __builtin_amdgcn_ds_permute(dest << 2, lane);
  for (int it = 0; it * 8 < n_need; ++it) {
    const int j = it * 8 + (lane >> 3);
    const int owner = __shfl(owner_of_rank, j & 63, 64);
    const uint32_t o = (uint32_t)__shfl((int)dst_unit, owner, 64);
    if (j < n_need) {
      const v4i32 r = stage[(size_t)(seg_base + owner) * kIdxSegUnits + (lane & 7)];
      __builtin_nontemporal_store(r, scratch + (size_t)o + (lane & 7));
    }
  }
}

// ------------------------------------------------------------------------------------------------------- phase 1
template <int NK, int NV, int RS, bool PK>
__global__ __launch_bounds__(kIdxBlock) void k_idx_scatter(const int8_t* const* __restrict__ cols,
                                                            const int64_t* __restrict__ num_rows, int n_frags, int n_cols,
                                                            IdxGeom g, IdxCols c, v4i32* __restrict__ scratch,
                                                            uint32_t* __restrict__ cnt, IdxSpill sl) {
  static_assert(PK ? (RS == 3 || (RS == 2 && NV >= 1)) : ((RS == 0 && NV >= 2) || (RS == 1 && NV == 1) || (RS == 2 && NV == 0)),
                "record size follows the value columns (or the packed word)");
  constexpr int NVA = NV > 0 ? NV : 1;  // (array sizes: no zero-length arrays)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v4i32* stage = (v4i32*)smem_raw;                                          // [P][L] units
  uint32_t* cursor = (uint32_t*)(smem_raw + (size_t)kIdxStageUnits * 16);   // [P] record positions handed out
  uint32_t* written = cursor + g.P;                                         // [P] records staged
  uint32_t* flushed = written + g.P;                                        // [P] lines sent = index of the open line
  uint32_t* done = flushed + g.P;                                           // producer waves finished
  unsigned long long* sp_blk = (unsigned long long*)(((uintptr_t)(done + 1) + 7) & ~(uintptr_t)7);
  const int t = threadIdx.x, b = blockIdx.x, G = gridDim.x;
  const int wave = t >> 6, lane = t & 63;
  const int lgL = g.lgL;              // units per line
  const int lgLr = g.lgL + RS;        // RECORDS per line
  const uint32_t Lrm1 = (g.L << RS) - 1;
  const uint32_t cap_recs = g.cap << RS;
  for (int i = t; i < 3 * g.P + 1; i += kIdxBlock) cursor[i] = 0;
  if (t == 0) *sp_blk = 0;
  __syncthreads();

  if (wave < kIdxProdWaves) {
    // ---------------------------------------------------------------------------------------------- producers
    constexpr int64_t kSuperQuads = (int64_t)kIdxProdWaves * 64;  // quads per workgroup step
    uint32_t kmin[NK], kcard[NK], kmul[NK], knull[NK];
    bool ktr[NK], vnull[NVA];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      kmin[k] = c.key_min[k];
      kcard[k] = c.key_card[k];
      kmul[k] = c.key_mul[k];
      knull[k] = c.key_null[k];
      ktr[k] = c.key_translate[k] != 0;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) vnull[v] = c.val_nullable[v] != 0;
    (void)vnull;
    v4i32 c_rec[4];
    uint32_t c_pid[4], c_slot[4];
    uint32_t c_mask = 0;
    bool bad = false;
    int f = 0;
    int64_t base = 0;
    int64_t nt_f = n_frags > 0 ? ((num_rows[0] + 3) / 4 + kSuperQuads - 1) / kSuperQuads : 0;
    int64_t gt = b;
    auto seek = [&]() {
      while (f < n_frags && gt >= base + nt_f) {
        base += nt_f;
        ++f;
        nt_f = f < n_frags ? ((num_rows[f] + 3) / 4 + kSuperQuads - 1) / kSuperQuads : 0;
      }
    };
    auto quad_of = [&]() -> int64_t { return (gt - base) * kSuperQuads + wave * 64 + lane; };
    struct Tile {
      v4i32 k[NK], v[NVA];
      int valid;
    };
    auto load_tile = [&](Tile& tl) {
      const int8_t* const* fc = cols + (size_t)f * n_cols;
      const int64_t n = num_rows[f];
      const int64_t quad = quad_of();
      const int64_t r0 = quad << 2;
      tl.valid = 0;
      if (r0 >= n) return;
      if (r0 + 4 <= n) {
#pragma unroll
        for (int k = 0; k < NK; ++k) tl.k[k] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)fc[c.key_col[k]] + quad);
#pragma unroll
        for (int v = 0; v < NV; ++v) tl.v[v] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)fc[c.val_col[v]] + quad);
        tl.valid = 4;
      } else {
        tl.valid = (int)(n - r0);
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const int32_t* src = (const int32_t*)fc[c.key_col[k]] + r0;
          tl.k[k] = v4i32{src[0], tl.valid > 1 ? src[1] : 0, tl.valid > 2 ? src[2] : 0, 0};
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int32_t* src = (const int32_t*)fc[c.val_col[v]] + r0;
          tl.v[v] = v4i32{src[0], tl.valid > 1 ? src[1] : 0, tl.valid > 2 ? src[2] : 0, 0};
        }
      }
    };
    // stage one record if its line is open; true = the record has left this lane
    auto try_stage = [&](const v4i32& rec, uint32_t p, uint32_t s) -> bool {
      if (idx_peek(&flushed[p]) != (s >> lgLr)) return false;
      asm volatile("" ::: "memory");  // compiler order only: the LDS itself runs a wave in order
      if (RS == 3) {
        ((uint16_t*)stage)[((size_t)p << lgLr) + (s & Lrm1)] = (uint16_t)rec.x;
      } else if (RS == 2) {
        ((uint32_t*)stage)[((size_t)p << lgLr) + (s & Lrm1)] = (uint32_t)rec.x;
      } else if (RS == 1) {
        ((unsigned long long*)stage)[((size_t)p << lgLr) + (s & Lrm1)] =
            ((unsigned long long)(uint32_t)rec.y << 32) | (unsigned long long)(uint32_t)rec.x;
      } else {
        stage[((size_t)p << lgL) + (s & Lrm1)] = rec;
      }
      asm volatile("" ::: "memory");
      atomicAdd(&written[p], 1u);
      return true;
    };
    auto retry_pending = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((c_mask & (1u << i)) && try_stage(c_rec[i], c_pid[i], c_slot[i])) c_mask &= ~(1u << i);
      }
    };
    seek();
    Tile cur;
    cur.valid = 0;
    bool have = f < n_frags;
    if (have) load_tile(cur);
    while (have) {
      gt += G;
      seek();
      const bool have_next = f < n_frags;
      Tile nxt;
      nxt.valid = 0;
      if (have_next) load_tile(nxt);
      retry_pending();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bool park = false;
        uint32_t p = 0, s = 0;
        v4i32 rec{0, 0, 0, 0};
        if (i < cur.valid) {
          // entry index, 32-bit throughout (idx_part_eligible admits |key_min| < 2^30 and cardinalities < 2^30)
          uint32_t e = 0;
          bool in_range = true;
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            const int32_t kv = idx_get(cur.k[k], i);
            uint32_t ku = (uint32_t)kv;
            if (ktr[k] && kv == INT32_MIN) ku = knull[k];
            const uint32_t dd = ku - kmin[k];
            in_range = in_range && dd < kcard[k];
            e += k == 0 ? dd : dd * kmul[k];  // (the first component's multiplier is 1: mul_lo is a quarter-rate instruction)
          }
          if (!in_range || e >= g.d) {
            bad = true;  // a key outside its declared range (reported as out of slots, like the row kernel)
          } else {
            rec.x = (int32_t)e;
            if (NV > 0) rec.y = idx_get(cur.v[0], i);
            if (NV > 1) rec.z = idx_get(cur.v[NV > 1 ? 1 : 0], i);
            if (NV > 2) rec.w = idx_get(cur.v[NV > 2 ? 2 : 0], i);
            uint32_t el = 0;
            if (PK) idx_part_split24(g, e, p, el);
            else p = idx_part_of(g, e);
            if (PK) {
              // the packed word; a value outside its declared range (the range is a hint) leaves as a full record
              const v4i32 full = rec;
              uint32_t w = el << g.vb;
              bool fits = true;
#pragma unroll
              for (int v = 0; v < NV; ++v) {
                const int32_t val = idx_get(full, v + 1);
                uint32_t code = 0;
                if (!(vnull[v] && val == INT32_MIN)) {
                  const uint32_t dv = (uint32_t)val - (uint32_t)c.val_min[v];
                  fits = fits && dv < c.val_card[v];
                  code = dv + (vnull[v] ? 1u : 0u);
                }
                w |= code << c.val_shift[v];
              }
              rec = v4i32{(int32_t)w, 0, 0, 0};
              if (!fits) {
                idx_spill(sl, sp_blk, full);
              } else {
                s = atomicAdd(&cursor[p], 1u);
                if (s >= cap_recs) idx_spill(sl, sp_blk, full);
                else park = !try_stage(rec, p, s);
              }
            } else {
              s = atomicAdd(&cursor[p], 1u);
              if (s >= cap_recs) idx_spill(sl, sp_blk, rec);
              else park = !try_stage(rec, p, s);
            }
          }
        }
        // line not open yet: park the record in pending slot i; if an older record still waits there, the WHOLE wave
        // waits for the flushers and keeps retrying every pending record of every lane (never spin in divergent code)
        uint32_t spins = 0;
        while (__any(park && (c_mask & (1u << i)))) {
          retry_pending();
          if (park && try_stage(rec, p, s)) park = false;
          __builtin_amdgcn_s_sleep(2);
          if (++spins > kIdxMaxSpins) {  // cannot happen unless the protocol is broken: bail out
            atomicExch(sl.d_err + 1, 2);
            c_mask = 0;
            break;
          }
        }
        if (park) {
          c_rec[i] = rec;
          c_pid[i] = p;
          c_slot[i] = s;
          c_mask |= 1u << i;
        }
      }
      cur = nxt;
      have = have_next;
    }
    for (uint32_t spins = 0; __any(c_mask != 0); ++spins) {
      retry_pending();
      __builtin_amdgcn_s_sleep(2);
      if (spins > kIdxMaxSpins) {
        atomicExch(sl.d_err + 1, 3);
        break;
      }
    }
    if (bad) atomicCAS(sl.d_err, 0, MI355Q_ERR_OUT_OF_SLOTS);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) atomicAdd(done, 1u);
    return;
  }

  // ------------------------------------------------------------------------------------------------ flushers
  // Flusher lane `fid` owns segments fid + 256 k (k = 0..3); the segments of one line (<= 64) share a pass and a wave.
  const int fid = (wave - kIdxProdWaves) * 64 + lane;
  const int lgSpl = lgL - 3;  // segments per line
  uint32_t my_fl[kIdxSegPerFlusher];
  for (int k = 0; k < kIdxSegPerFlusher; ++k) my_fl[k] = 0;
  uint32_t idle = 0;
  for (;;) {
    const bool all_done = idx_peek(done) == (uint32_t)kIdxProdWaves;  // read BEFORE the scan
    bool any_work = false;
#pragma unroll
    for (int k = 0; k < kIdxSegPerFlusher; ++k) {
      const int seg = k * (kIdxFlushWaves * 64) + fid;
      const int p = seg >> lgSpl;
      const uint32_t sidx = (uint32_t)seg & ((1u << lgSpl) - 1);
      const uint32_t w = idx_peek(&written[p]);
      const bool need = (w >> lgLr) > my_fl[k] && (my_fl[k] << lgL) < g.cap;
      asm volatile("" ::: "memory");
      idx_flush_segments(need, ((uint32_t)p * g.B + b) * g.cap + (my_fl[k] << lgL) + sidx * kIdxSegUnits, stage, seg - lane,
                         scratch);
      asm volatile("" ::: "memory");
      if (need) {
        my_fl[k] += 1;
        if (sidx == 0) idx_poke(&flushed[p], my_fl[k]);  // after this wave's reads of the line
      }
      any_work |= need;
    }
    if (!__any(any_work)) {
      if (all_done) break;
      __builtin_amdgcn_s_sleep(4);
      if (++idle > kIdxMaxSpins) {  // producers stuck: give up rather than hang the device
        atomicExch(sl.d_err + 1, 4);
        break;
      }
    } else {
      idle = 0;
    }
  }
  // every producer is done and every complete line is out: drain the partial open lines
#pragma unroll
  for (int k = 0; k < kIdxSegPerFlusher; ++k) {
    const int seg = k * (kIdxFlushWaves * 64) + fid;
    const int p = seg >> lgSpl;
    const uint32_t sidx = (uint32_t)seg & ((1u << lgSpl) - 1);
    const uint32_t w = idx_peek(&written[p]);
    const uint32_t rem_units = ((w - (my_fl[k] << lgLr)) + ((1u << RS) - 1u)) >> RS;  // < L (+ the partly filled last unit)
    const bool need = sidx * kIdxSegUnits < rem_units && (my_fl[k] << lgL) + sidx * kIdxSegUnits < g.cap;
    idx_flush_segments(need, ((uint32_t)p * g.B + b) * g.cap + (my_fl[k] << lgL) + sidx * kIdxSegUnits, stage, seg - lane,
                       scratch);
    if (sidx == 0) {
      const uint32_t cc = idx_peek(&cursor[p]);
      cnt[(size_t)p * g.B + b] = cc < cap_recs ? cc : cap_recs;
    }
  }
  // the unused tail of this workgroup's last spill block must read as "no entry"
  {
    const unsigned long long w = *sp_blk;
    const uint32_t next = (uint32_t)w, end = (uint32_t)(w >> 32);
    for (uint32_t i = next + fid; i < end && i < sl.cap; i += kIdxFlushWaves * 64) sl.entries[i] = v4i32{-1, 0, 0, 0};
  }
}

// --------------------------------------------------------------------------------------------- emission (reduce rule)
// One group's partial — rows, and per value column non-NULL count / sum / min / max — merged into row `e` of the output
// table, target by target, with the reduce rule (reduce_target: what ResultSetStorage::reduce does with two buffers), as the
// flush of k_groupby_lds_typed does.  The table is initialised before the step, so a later chunk merges again.
// A: atomic merges (the spill records: several lanes may hold records of one entry); phase 2 emits every entry from exactly one
// lane of one workgroup per launch, so it merges with plain read-modify-writes (10 M entries x 7 slots of global atomics cost
// PHS007 ~2 ms per launch).
template <int NK, bool A>
MQ_D void idx_emit(const IdxCols& c, const DevPlan& p, int64_t* __restrict__ out, uint32_t e, uint32_t rows,
                   const uint32_t (&cn)[kLdsVals], const int64_t (&sm)[kLdsVals], const int32_t (&mn)[kLdsVals],
                   const int32_t (&mx)[kLdsVals]) {
  int64_t key0 = 0, key1 = 0, key2 = 0, tk0 = 0, tk1 = 0, tk2 = 0;
  uint32_t rem = e;
#pragma unroll
  for (int g = NK - 1; g >= 0; --g) {  // entry index -> key components (mul_g ascending with g)
    const uint32_t dd = rem / c.key_mul[g];
    rem -= dd * c.key_mul[g];
    const int64_t tk = (int64_t)dd + c.key_min64[g];
    const int64_t orig = (c.key_translate[g] && tk == c.key_null64[g]) ? (int64_t)INT32_MIN : tk;
    if (g == 0) { tk0 = tk; key0 = orig; } else if (g == 1) { tk1 = tk; key1 = orig; } else { tk2 = tk; key2 = orig; }
  }
  int64_t* row = out + (size_t)e * p.row_quad;
  int64_t* slots = row;
  if (!p.keyless) {
    if (MQ_LOAD64(row) == kEmptyKey64) {
      if (NK > 2) MQ_STORE64(row + 2, tk2);
      if (NK > 1) MQ_STORE64(row + 1, tk1);
      MQ_STORE64(row, tk0);
    }
    slots = row + NK;
  }
#pragma unroll
  for (int i = 0; i < MI355Q_MAX_TARGETS; ++i) {
    if (i >= p.n_targets) break;
    const DevTarget& tg = p.targets[i];
    if (tg.slot < 0) continue;
    int64_t v0 = p.init_vals[tg.slot], v1 = tg.agg == MI355Q_AVG ? p.init_vals[tg.slot + 1] : 0;
    if (tg.agg == MI355Q_PROJECT_KEY) {
      v0 = tg.key_idx == 0 ? key0 : tg.key_idx == 1 ? key1 : key2;
    } else {
      const int vc = c.target_v[i];
      if (vc < 0) {
        v0 = (int64_t)rows;
      } else {
        const uint32_t cnt_v = vc == 0 ? cn[0] : vc == 1 ? cn[1] : cn[2];
        const int64_t sum_v = vc == 0 ? sm[0] : vc == 1 ? sm[1] : sm[2];
        const int32_t min_v = vc == 0 ? mn[0] : vc == 1 ? mn[1] : mn[2];
        const int32_t max_v = vc == 0 ? mx[0] : vc == 1 ? mx[1] : mx[2];
        switch (tg.agg) {
          case MI355Q_COUNT: v0 = (int64_t)cnt_v; break;
          case MI355Q_AVG:
            v1 = (int64_t)cnt_v;
            [[fallthrough]];
          case MI355Q_SUM:
            if (cnt_v) v0 = sum_v;
            break;
          case MI355Q_MIN:
            if (cnt_v) v0 = (int64_t)min_v;
            break;
          default:
            if (cnt_v) v0 = (int64_t)max_v;
        }
      }
    }
    int64_t win2[2] = {v0, v1};
    DevTarget lt = tg;
    lt.slot = 0;
    reduce_target<A>(lt, p.init_vals + tg.slot, slots + tg.slot, win2);
  }
}

// ------------------------------------------------------------------------------------------------------- phase 2
template <int NK, int NV, bool MM, int RS>
__global__ __launch_bounds__(kIdxBlock) void k_idx_aggregate(IdxGeom g, IdxCols c, DevPlan p, const v4i32* __restrict__ scratch,
                                                              const uint32_t* __restrict__ cnt, int64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t E = g.S2;
  // sum[NV][E] i64 | rows[E] u32 | cnt[NV][E] u32 | min[NV][E] i32 | max[NV][E] i32 | lcnt[B] u32
  constexpr int NVA = NV > 0 ? NV : 1;
  unsigned long long* l_sum = (unsigned long long*)smem_raw;
  uint32_t* l_rows = (uint32_t*)(l_sum + (size_t)NV * E);
  uint32_t* l_cnt = l_rows + E;
  int32_t* l_min = (int32_t*)(l_cnt + (size_t)NV * E);
  int32_t* l_max = l_min + (MM ? (size_t)NV * E : 0);
  uint32_t* lcnt = (uint32_t*)(l_max + (MM ? (size_t)NV * E : 0));
  const int t = threadIdx.x, G = gridDim.x;
  const int R = (int)g.R;
  bool vnull[NVA];
#pragma unroll
  for (int v = 0; v < NV; ++v) vnull[v] = c.val_nullable[v] != 0;
  (void)vnull;
  for (int it = 0;; ++it) {
    const int u = blockIdx.x + G * it;
    const int pp = u / R, r = u % R;
    if (pp >= g.P) break;
    const uint64_t p_lo = (uint64_t)pp * g.S1;
    const uint64_t p_hi = p_lo + g.S1 < g.d ? p_lo + g.S1 : g.d;
    uint64_t a = p_lo + (uint64_t)r * g.S2, z = a + g.S2;
    if (z > p_hi) z = p_hi;
    if (a > z) a = z;
    const uint32_t lo = (uint32_t)a, n_slots = (uint32_t)(z - a);
    if (it > 0) __syncthreads();  // the previous unit's emission is done before the arrays are cleared
    for (int i = t; i < g.B; i += kIdxBlock) lcnt[i] = cnt[(size_t)pp * g.B + i];
    for (uint32_t e = t; e < n_slots; e += kIdxBlock) {
      l_rows[e] = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        l_sum[v * E + e] = 0;
        l_cnt[v * E + e] = 0;
        if (MM) {
          l_min[v * E + e] = INT32_MAX;
          l_max[v * E + e] = INT32_MIN;
        }
      }
    }
    __syncthreads();
    if (n_slots) {
      auto one = [&](uint32_t ge, int32_t v0, int32_t v1, int32_t v2) {
        const uint32_t e = ge - lo;
        if (e >= n_slots) return;  // the partition's other sub-range
        atomicAdd(l_rows + e, 1u);
        const int32_t vv[3] = {v0, v1, v2};
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (vnull[v] && vv[v] == INT32_MIN) continue;
          atomicAdd(l_cnt + v * E + e, 1u);
          atomicAdd(l_sum + v * E + e, (unsigned long long)(int64_t)vv[v]);
          if (MM) {
            atomicMin(l_min + v * E + e, vv[v]);
            atomicMax(l_max + v * E + e, vv[v]);
          }
        }
      };
      // a unit of the run = 16 bytes = one (RS = 0), two (RS = 1) or four (RS = 2) records; `n` counts records
      auto unit = [&](const v4i32& q, uint32_t rec0, uint32_t n) {
        if (RS == 2) {
          if (rec0 < n) one((uint32_t)q.x, 0, 0, 0);
          if (rec0 + 1 < n) one((uint32_t)q.y, 0, 0, 0);
          if (rec0 + 2 < n) one((uint32_t)q.z, 0, 0, 0);
          if (rec0 + 3 < n) one((uint32_t)q.w, 0, 0, 0);
        } else if (RS) {
          if (rec0 < n) one((uint32_t)q.x, q.y, 0, 0);
          if (rec0 + 1 < n) one((uint32_t)q.z, q.w, 0, 0);
        } else if (rec0 < n) {
          one((uint32_t)q.x, q.y, q.z, q.w);
        }
      };
      // one wave per run; four 16-byte loads per lane, the next four already in flight
      const int wave = t >> 6, lane = t & 63;
      for (int b = wave; b < g.B; b += kIdxBlock / 64) {
        const uint32_t n = lcnt[b];
        if (!n) continue;
        const v4i32* run = scratch + ((size_t)pp * g.B + b) * g.cap;
        const uint32_t n_units = (n + ((1u << RS) - 1u)) >> RS;
        const uint32_t last = n_units - 1;
        auto at = [&](uint32_t i) -> uint32_t { return i < last ? i : last; };  // clamped: always loadable
        v4i32 c0 = __builtin_nontemporal_load(run + at(lane)), c1 = __builtin_nontemporal_load(run + at(lane + 64)),
              c2 = __builtin_nontemporal_load(run + at(lane + 128)), c3 = __builtin_nontemporal_load(run + at(lane + 192));
        for (uint32_t base = 0; base < n_units; base += 256) {
          const uint32_t i = base + lane, nx = i + 256;
          const v4i32 n0 = __builtin_nontemporal_load(run + at(nx)), n1 = __builtin_nontemporal_load(run + at(nx + 64)),
                      n2 = __builtin_nontemporal_load(run + at(nx + 128)), n3 = __builtin_nontemporal_load(run + at(nx + 192));
          // (a clamped index repeats the last unit: its records are dropped through the record position)
          unit(c0, i < n_units ? (i << RS) : n, n);
          unit(c1, i + 64 < n_units ? ((i + 64) << RS) : n, n);
          unit(c2, i + 128 < n_units ? ((i + 128) << RS) : n, n);
          unit(c3, i + 192 < n_units ? ((i + 192) << RS) : n, n);
          c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
      }
    }
    __syncthreads();
    for (uint32_t e = t; e < n_slots; e += kIdxBlock) {
      const uint32_t rows = l_rows[e];
      if (!rows) continue;
      uint32_t cn[kLdsVals] = {0, 0, 0};
      int64_t sm[kLdsVals] = {0, 0, 0};
      int32_t mn[kLdsVals] = {0, 0, 0}, mx[kLdsVals] = {0, 0, 0};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        cn[v] = l_cnt[v * E + e];
        sm[v] = (int64_t)l_sum[v * E + e];
        if (MM) {
          mn[v] = l_min[v * E + e];
          mx[v] = l_max[v * E + e];
        }
      }
      idx_emit<NK, false>(c, p, out, lo + e, rows, cn, sm, mn, mx);
    }
  }
}

// ------------------------------------------------------------------------------------- phase 2, packed records
// The packed word names small codes, so a unit's LDS entry shrinks and a record costs fewer LDS atomics (the plain member:
// rows + per value column cnt, sum (64-bit), min, max = 5 for PHS005; this one: 2):
//   cs[NV][E]   u64  {non-NULL values : sum of their codes} in ONE 64-bit add (the plan bounds records x max code < 2^32)
//   first[E]    u32  NV = 0: the entry's rows; else the rows whose FIRST value is NULL (rows = cs[0].cnt + first: a NULL is
//                    the rare case, the common record does not touch it)
//   MMODE 4:    mask[NV][E] u32, bit = code seen (every column <= 32 values: min and max from one atomicOr)
//   MMODE 1-3:  min[NV][E] (bit 0) / max[NV][E] (bit 1) of the codes, u32
// Emission converts back: sum = cnt x min_value + sum of codes, min / max = min_value + code.
template <int NV, int MMODE>
__global__ __launch_bounds__(kIdxBlock) void k_idx_aggregate_pk(IdxGeom g, IdxCols c, DevPlan p, const v4i32* __restrict__ scratch,
                                                                 const uint32_t* __restrict__ cnt, int64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const uint32_t E = g.S2;
  constexpr int NVA = NV > 0 ? NV : 1;
  constexpr int NM = MMODE == 4 ? 1 : (MMODE & 1) + ((MMODE >> 1) & 1);  // 4-byte arrays per value column
  unsigned long long* l_cs = (unsigned long long*)smem_raw;
  uint32_t* l_first = (uint32_t*)(l_cs + (size_t)NV * E);
  uint32_t* l_m0 = l_first + E;                                              // mask, or min (or max alone)
  uint32_t* l_m1 = l_m0 + (MMODE == 3 ? (size_t)NV * E : 0);                 // max beside min
  uint32_t* lcnt = l_m0 + (size_t)NM * NV * E;
  const int t = threadIdx.x, G = gridDim.x;
  const int R = (int)g.R;
  const uint32_t vb = g.vb;
  const bool half = g.rs == 3;
  bool vnull[NVA];
  uint32_t vsh[NVA], vmk[NVA];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    vnull[v] = c.val_nullable[v] != 0;
    vsh[v] = c.val_shift[v];
    vmk[v] = c.val_mask[v];
  }
  (void)vnull; (void)vsh; (void)vmk;
  for (int it = 0;; ++it) {
    const int u = blockIdx.x + G * it;
    const int pp = u / R, r = u % R;
    if (pp >= g.P) break;
    const uint64_t p_lo = (uint64_t)pp * g.S1;
    const uint64_t p_hi = p_lo + g.S1 < g.d ? p_lo + g.S1 : g.d;
    uint64_t a = p_lo + (uint64_t)r * g.S2, z = a + g.S2;
    if (z > p_hi) z = p_hi;
    if (a > z) a = z;
    const uint32_t lo = (uint32_t)a, n_slots = (uint32_t)(z - a);
    const uint32_t sub_lo = (uint32_t)(a - p_lo);
    if (it > 0) __syncthreads();  // the previous unit's emission is done before the arrays are cleared
    for (int i = t; i < g.B; i += kIdxBlock) lcnt[i] = cnt[(size_t)pp * g.B + i];
    for (uint32_t e = t; e < n_slots; e += kIdxBlock) {
      l_first[e] = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        l_cs[v * E + e] = 0;
        if (MMODE == 4) l_m0[v * E + e] = 0;
        if (MMODE == 1 || MMODE == 3) l_m0[v * E + e] = 0xffffffffu;
        if (MMODE == 2) l_m0[v * E + e] = 0;
        if (MMODE == 3) l_m1[v * E + e] = 0;
      }
    }
    __syncthreads();
    if (n_slots) {
      auto one = [&](uint32_t w) {
        const uint32_t e = (w >> vb) - sub_lo;
        if (e >= n_slots) return;  // the partition's other sub-range
        if (NV == 0) atomicAdd(l_first + e, 1u);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          uint32_t code = (w >> vsh[v]) & vmk[v];
          if (vnull[v]) {
            if (code == 0) {
              if (v == 0) atomicAdd(l_first + e, 1u);
              continue;
            }
            code -= 1;
          }
          atomicAdd(l_cs + v * E + e, ((unsigned long long)1 << 32) | (unsigned long long)code);
          if (MMODE == 4) atomicOr(l_m0 + v * E + e, 1u << code);
          if (MMODE == 1 || MMODE == 3) atomicMin(l_m0 + v * E + e, code);
          if (MMODE == 2) atomicMax(l_m0 + v * E + e, code);
          if (MMODE == 3) atomicMax(l_m1 + v * E + e, code);
        }
      };
      // a unit of the run = 16 bytes = four words or eight half-words; `n` counts records
      auto unit = [&](const v4i32& q, uint32_t rec0, uint32_t n) {
        const uint32_t ww[4] = {(uint32_t)q.x, (uint32_t)q.y, (uint32_t)q.z, (uint32_t)q.w};
        if (half) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (rec0 + 2 * j < n) one(ww[j] & 0xffffu);
            if (rec0 + 2 * j + 1 < n) one(ww[j] >> 16);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (rec0 + j < n) one(ww[j]);
        }
      };
      const int rs = half ? 3 : 2;
      const int wave = t >> 6, lane = t & 63;
      for (int b = wave; b < g.B; b += kIdxBlock / 64) {
        const uint32_t n = lcnt[b];
        if (!n) continue;
        const v4i32* run = scratch + ((size_t)pp * g.B + b) * g.cap;
        const uint32_t n_units = (n + ((1u << rs) - 1u)) >> rs;
        const uint32_t last = n_units - 1;
        auto at = [&](uint32_t i) -> uint32_t { return i < last ? i : last; };  // clamped: always loadable
        v4i32 c0 = __builtin_nontemporal_load(run + at(lane)), c1 = __builtin_nontemporal_load(run + at(lane + 64));
        for (uint32_t base = 0; base < n_units; base += 128) {
          const uint32_t i = base + lane, nx = i + 128;
          const v4i32 n0 = __builtin_nontemporal_load(run + at(nx)), n1 = __builtin_nontemporal_load(run + at(nx + 64));
          unit(c0, i < n_units ? (i << rs) : n, n);
          unit(c1, i + 64 < n_units ? ((i + 64) << rs) : n, n);
          c0 = n0; c1 = n1;
        }
      }
    }
    __syncthreads();
    for (uint32_t e = t; e < n_slots; e += kIdxBlock) {
      uint32_t cn[kLdsVals] = {0, 0, 0};
      int64_t sm[kLdsVals] = {0, 0, 0};
      int32_t mn[kLdsVals] = {0, 0, 0}, mx[kLdsVals] = {0, 0, 0};
      uint32_t rows = l_first[e];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const unsigned long long cs = l_cs[v * E + e];
        cn[v] = (uint32_t)(cs >> 32);
        sm[v] = (int64_t)cn[v] * (int64_t)c.val_min[v] + (int64_t)(uint32_t)cs;
        if (v == 0) rows += cn[0];
        if (cn[v]) {
          if (MMODE == 4) {
            const uint32_t m = l_m0[v * E + e];
            mn[v] = c.val_min[v] + (int32_t)__builtin_ctz(m);
            mx[v] = c.val_min[v] + 31 - (int32_t)__builtin_clz(m);
          }
          if (MMODE == 1 || MMODE == 3) mn[v] = c.val_min[v] + (int32_t)l_m0[v * E + e];
          if (MMODE == 2) mx[v] = c.val_min[v] + (int32_t)l_m0[v * E + e];
          if (MMODE == 3) mx[v] = c.val_min[v] + (int32_t)l_m1[v * E + e];
        }
      }
      if (!rows) continue;
      switch (g.nk) {
        case 1: idx_emit<1, false>(c, p, out, lo + e, rows, cn, sm, mn, mx); break;
        case 2: idx_emit<2, false>(c, p, out, lo + e, rows, cn, sm, mn, mx); break;
        default: idx_emit<3, false>(c, p, out, lo + e, rows, cn, sm, mn, mx);
      }
    }
  }
}

// records that met a full run: applied one by one, each as the partial of a single row
template <int NK>
__global__ __launch_bounds__(256) void k_idx_spill(IdxGeom g, IdxCols c, DevPlan p, IdxSpill sl, int64_t* __restrict__ out) {
  uint32_t n = *sl.count;
  if (n > sl.cap) n = sl.cap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const v4i32 r = sl.entries[i];
    if (r.x < 0 || (uint32_t)r.x >= g.d) continue;  // the unused tail of a spill block
    const int32_t vv[3] = {r.y, r.z, r.w};
    uint32_t cn[kLdsVals] = {0, 0, 0};
    int64_t sm[kLdsVals] = {0, 0, 0};
    int32_t mn[kLdsVals] = {0, 0, 0}, mx[kLdsVals] = {0, 0, 0};
    for (int v = 0; v < g.nv; ++v) {
      if (c.val_nullable[v] && vv[v] == INT32_MIN) continue;
      cn[v] = 1;
      sm[v] = vv[v];
      mn[v] = mx[v] = vv[v];
    }
    idx_emit<NK, true>(c, p, out, (uint32_t)r.x, 1u, cn, sm, mn, mx);
  }
}

// --------------------------------------------------------------------------------------------------------- host
struct IdxPlanHost {
  IdxGeom g;
  IdxCols c;
  int64_t chunk_rows;
  int64_t rec_bytes, cnt_bytes, scratch_bytes;
  size_t lds1, lds2;
};

bool make_idx_plan_impl(const DevPlan& p, const FragView& fv, int n_cus, int64_t scratch_cap, bool allow_pk, IdxPlanHost* out) {
  IdxPlanHost& h = *out;
  std::memset(&h, 0, sizeof(h));
  LdsArgs a;
  bool need[kLdsVals][4];
  if (tune_knobs().flags & MI355Q_OPT_NO_IDX_PART) return false;
  if (!lds_describe(p, fv, ((int64_t)1 << 31) - 1, 0, &a, need)) return false;
  if (a.baseline || a.n_flt != 0 || a.n_keys < 1) return false;
  // (a.n_vals == 0: only COUNT(*) / key projections — Sort/S001-003: SELECT key, COUNT(*) ... GROUP BY key — 4-byte records
  // that carry the entry index alone, an LDS table of row counts)
  if (p.entry_count <= 65536 || fv.n_frags < 1 || n_cus < 1) return false;
  if (fv.max_frag_rows > 0xfff00000ll) return false;
  for (int v = 0; v < a.n_vals; ++v) {
    if (a.v[v].type != MI355Q_INT32) return false;
    h.c.val_col[v] = a.v[v].col;
    h.c.val_nullable[v] = a.v[v].nullable;
    h.g.mm |= (need[v][2] || need[v][3]) ? 1 : 0;
  }
  for (int k = 0; k < a.n_keys; ++k) {
    if (a.key_type[k] != MI355Q_INT32) return false;
    if (a.key_min[k] <= -(1ll << 30) || a.key_min[k] >= (1ll << 30)) return false;
    if (a.key_card[k] < 1 || a.key_card[k] >= (1ll << 30) || a.key_mul[k] < 1 || a.key_mul[k] >= (1ll << 31)) return false;
    h.c.key_col[k] = a.key_col[k];
    h.c.key_translate[k] = a.key_translate[k];
    h.c.key_min[k] = (uint32_t)(int32_t)a.key_min[k];
    h.c.key_card[k] = (uint32_t)a.key_card[k];
    h.c.key_mul[k] = (uint32_t)a.key_mul[k];
    h.c.key_null[k] = (uint32_t)a.key_null_key[k];
    h.c.key_min64[k] = a.key_min[k];
    h.c.key_null64[k] = a.key_null_key[k];
  }
  for (int k = a.n_keys; k < kLdsKeys; ++k) h.c.key_mul[k] = 1;
  for (int i = 0; i < MI355Q_MAX_TARGETS; ++i) h.c.target_v[i] = i < p.n_targets ? a.target_v[i] : -1;
  IdxGeom& g = h.g;
  g.nk = a.n_keys;
  g.nv = a.n_vals;
  g.rs = a.n_vals == 0 ? 2 : a.n_vals == 1 ? 1 : 0;
  g.d = (uint32_t)p.entry_count;
  // packed records: every value column's range known (DevTarget::arg_rng) and narrow
  uint32_t vb = 0, max_code = 0;
  bool known = allow_pk && !(tune_knobs().flags & MI355Q_OPT_NO_IDX_PACK);
  bool mask_ok = true;  // every column <= 32 values: min / max from a presence mask
  int need_min = 0, need_max = 0;
  for (int v = 0; v < g.nv; ++v) {
    need_min |= need[v][2] ? 1 : 0;
    need_max |= need[v][3] ? 1 : 0;
  }
  for (int v = 0; v < g.nv && known; ++v) {
    const DevTarget* tg = nullptr;
    for (int i = 0; i < p.n_targets && !tg; ++i)
      if (a.target_v[i] == v) tg = &p.targets[i];
    if (!tg || !tg->arg_rng || tg->arg_lo > tg->arg_hi) {
      known = false;
      break;
    }
    const uint64_t card = (uint64_t)((int64_t)tg->arg_hi - (int64_t)tg->arg_lo + 1);
    const uint64_t codes = card + (h.c.val_nullable[v] ? 1 : 0);
    uint32_t bits = 0;
    while (bits < 33 && ((uint64_t)1 << bits) < codes) ++bits;
    if (bits > 20) {
      known = false;
      break;
    }
    h.c.val_min[v] = tg->arg_lo;
    h.c.val_card[v] = (uint32_t)card;
    h.c.val_shift[v] = vb;
    h.c.val_mask[v] = (uint32_t)(((uint64_t)1 << bits) - 1);
    vb += bits;
    if (card > 32) mask_ok = false;
    if ((uint32_t)card - 1 > max_code) max_code = (uint32_t)card - 1;
  }
  // LDS entries of one unit.  Plain records: rows u32 + per value column cnt u32, sum i64 (+ min, max i32).  Packed records
  // (k_idx_aggregate_pk): first u32 + per value column {cnt : sum of codes} u64 (+ a presence mask u32, or min / max codes u32)
  const int mmode = !g.mm ? 0 : mask_ok ? 4 : (need_min ? 1 : 0) | (need_max ? 2 : 0);
  const size_t entry_pk = 4 + (size_t)g.nv * (8 + (mmode == 4 ? 4 : mmode == 3 ? 8 : mmode ? 4 : 0));
  const size_t entry_plain = 4 + (size_t)g.nv * (12 + (g.mm ? 8 : 0));
  size_t entry_bytes = 0;
  uint32_t P = 0;
  for (int attempt = known ? 0 : 1; attempt < 2; ++attempt) {
    const bool pk = attempt == 0;
    entry_bytes = pk ? entry_pk : entry_plain;
    const uint32_t e_max = (uint32_t)(((pk ? kIdxLdsTablePk : kIdxLdsTable) - (size_t)n_cus * 4) / entry_bytes) & ~3u;
    if (e_max < 64) return false;
    const uint64_t units = ((uint64_t)g.d + e_max - 1) / e_max;
    P = 16;  // a line's segments (<= 64) must fit one flusher wave pass
    while (P < 1024 && P < units) P <<= 1;
    while (P < 1024 && P < (uint32_t)n_cus && (uint64_t)P * 64 <= g.d) P <<= 1;  // enough units to occupy the device
    g.P = (int32_t)P;
    g.S1 = (uint32_t)(((uint64_t)g.d + P - 1) / P);
    if (g.S1 < 2) return false;
    g.s1_rcp = (uint32_t)(((uint64_t)1 << 32) / g.S1);
    g.R = (g.S1 + e_max - 1) / e_max;
    if (g.R < 1) g.R = 1;
    g.S2 = ((g.S1 + g.R - 1) / g.R + 3) & ~3u;  // (every array of the LDS table stays 16-byte aligned)
    g.pk = 0;
    g.rs = a.n_vals == 0 ? 2 : a.n_vals == 1 ? 1 : 0;
    if (!pk) break;
    uint32_t eb = 0;
    while (eb < 32 && ((uint64_t)1 << eb) < (uint64_t)g.S1) ++eb;
    // (<= 2^24 entries, partitions of > 256 entries: idx_part_split24's operands fit 24 bits)
    const bool fits24 = g.d <= (1u << 24) && g.S1 > 256u && g.s1_rcp < (1u << 24);
    if (fits24 && g.R <= (uint32_t)kIdxMaxSub && (eb + vb <= 16 || (g.nv >= 1 && eb + vb <= 32))) {
      g.pk = 1;
      g.rs = eb + vb <= 16 ? 3 : 2;
      g.vb = vb;
      g.mmode = mmode;
      break;
    }
  }
  if (g.R > (uint32_t)kIdxMaxSub) return false;
  g.L = kIdxStageUnits / P;
  g.lgL = 0;
  while ((1u << g.lgL) < g.L) ++g.lgL;
  g.B = n_cus;
  const int recs_per_unit = 1 << g.rs;
  int64_t chunk_rows = fv.total_rows > 0 ? fv.total_rows : 1;
  if (chunk_rows > 0xfff00000ll) chunk_rows = 0xfff00000ll;  // 32-bit run positions / LDS counters per chunk
  if (scratch_cap <= 0) scratch_cap = kDefaultScratchCap;
  for (;;) {
    const double per_run = (double)chunk_rows / ((double)P * g.B);  // records
    uint64_t cap = (uint64_t)((per_run * 1.2 + 6.0 * __builtin_sqrt(per_run + 1.0)) / recs_per_unit) + g.L;  // units
    cap = (cap + g.L - 1) / g.L * g.L;  // whole lines
    const bool too_many = (uint64_t)P * g.B * cap >= ((uint64_t)1 << 31);  // 32-bit unit indices in phase 1
    if (!too_many) {
      // packed records: a unit's {cnt : sum of codes} word must hold every record of a partition (B runs of cap units)
      if (g.pk && (uint64_t)g.B * cap * recs_per_unit * (uint64_t)(max_code ? max_code : 1) >= ((uint64_t)1 << 32)) return false;
      g.cap = (uint32_t)cap;
      h.rec_bytes = (int64_t)P * g.B * (int64_t)cap * 16;
      h.cnt_bytes = ((int64_t)P * g.B * 4 + 255) & ~255ll;
      int64_t spill_cap = chunk_rows / 16;
      if (spill_cap < (int64_t)kIdxSpillMin) spill_cap = kIdxSpillMin;
      if (spill_cap > 0x3fffffffll) spill_cap = 0x3fffffffll;
      g.spill_cap = (uint32_t)spill_cap;
      h.scratch_bytes = h.rec_bytes + h.cnt_bytes + 256 + spill_cap * 16;
    }
    if (!too_many && (h.scratch_bytes <= scratch_cap || chunk_rows <= fv.max_frag_rows)) break;
    if (chunk_rows <= fv.max_frag_rows) return false;
    chunk_rows = (int64_t)(chunk_rows * 0.9);
    if (chunk_rows < fv.max_frag_rows) chunk_rows = fv.max_frag_rows;
  }
  if (fv.total_rows > chunk_rows) {  // equal-sized chunks
    const int64_t n_chunks = (fv.total_rows + chunk_rows - 1) / chunk_rows;
    const int64_t even = (fv.total_rows + n_chunks - 1) / n_chunks + fv.max_frag_rows;
    if (even < chunk_rows) chunk_rows = even;
  }
  h.chunk_rows = chunk_rows;
  h.lds1 = (size_t)kIdxStageUnits * 16 + (size_t)P * 12 + 32;
  h.lds2 = (size_t)g.S2 * entry_bytes + (size_t)g.B * 4 + 16;
  return h.lds1 <= 160 * 1024 && h.lds2 <= 160 * 1024;
}

bool make_idx_plan(const DevPlan& p, const FragView& fv, int n_cus, int64_t scratch_cap, IdxPlanHost* out) {
  // (the packed word first; its only own refusal is the bound of the sums of codes)
  return make_idx_plan_impl(p, fv, n_cus, scratch_cap, true, out) || make_idx_plan_impl(p, fv, n_cus, scratch_cap, false, out);
}

template <int NK, int NV, int RS, bool PK>
hipError_t idx_launch_scatter(const IdxPlanHost& h, const FragView& fv, int f0, int nf, v4i32* recs, uint32_t* cnt,
                              const IdxSpill& sl, hipStream_t s) {
  auto k = k_idx_scatter<NK, NV, RS, PK>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds1);
  hipLaunchKernelGGL(k, dim3(h.g.B), dim3(kIdxBlock), h.lds1, s, fv.d_cols + (size_t)f0 * fv.n_cols, fv.d_num_rows + f0, nf,
                     fv.n_cols, h.g, h.c, recs, cnt, sl);
  return hipGetLastError();
}
template <int NK, int NV, bool MM, int RS>
hipError_t idx_launch_aggregate(const IdxPlanHost& h, const DevPlan& p, const v4i32* recs, const uint32_t* cnt, int64_t* out,
                                int n_cus, hipStream_t s) {
  auto k = k_idx_aggregate<NK, NV, MM, RS>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds2);
  const int units = h.g.P * (int)h.g.R;
  hipLaunchKernelGGL(k, dim3(units < n_cus ? units : n_cus), dim3(kIdxBlock), h.lds2, s, h.g, h.c, p, recs, cnt, out);
  return hipGetLastError();
}

template <int NV, int MMODE>
hipError_t idx_launch_aggregate_pk(const IdxPlanHost& h, const DevPlan& p, const v4i32* recs, const uint32_t* cnt, int64_t* out,
                                   int n_cus, hipStream_t s) {
  auto k = k_idx_aggregate_pk<NV, MMODE>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds2);
  const int units = h.g.P * (int)h.g.R;
  hipLaunchKernelGGL(k, dim3(units < n_cus ? units : n_cus), dim3(kIdxBlock), h.lds2, s, h.g, h.c, p, recs, cnt, out);
  return hipGetLastError();
}
template <int NV>
hipError_t idx_aggregate_pk(const IdxPlanHost& h, const DevPlan& p, const v4i32* recs, const uint32_t* cnt, int64_t* out, int n_cus,
                            hipStream_t s) {
  if (NV == 0) return idx_launch_aggregate_pk<NV, 0>(h, p, recs, cnt, out, n_cus, s);
  switch (h.g.mmode) {
    case 0: return idx_launch_aggregate_pk<NV, 0>(h, p, recs, cnt, out, n_cus, s);
    case 1: return idx_launch_aggregate_pk<NV, (NV > 0 ? 1 : 0)>(h, p, recs, cnt, out, n_cus, s);
    case 2: return idx_launch_aggregate_pk<NV, (NV > 0 ? 2 : 0)>(h, p, recs, cnt, out, n_cus, s);
    case 3: return idx_launch_aggregate_pk<NV, (NV > 0 ? 3 : 0)>(h, p, recs, cnt, out, n_cus, s);
    default: return idx_launch_aggregate_pk<NV, (NV > 0 ? 4 : 0)>(h, p, recs, cnt, out, n_cus, s);
  }
}

template <int NK, int NV, int RS, bool PK>
hipError_t idx_run(const IdxPlanHost& h, const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, void* scratch,
                   int n_cus, hipStream_t s, LaunchStats* st) {
  v4i32* recs = (v4i32*)scratch;
  uint32_t* cnt = (uint32_t*)((char*)scratch + h.rec_bytes);
  char* spill_base = (char*)scratch + h.rec_bytes + h.cnt_bytes;
  IdxSpill sl{(uint32_t*)spill_base, (v4i32*)(spill_base + 256), d_err, h.g.spill_cap};
  hipEvent_t* ev_pool = st->ev_pool;
  int ev_i = 0;
  int f = 0;
  while (f < fv.n_frags) {
    int64_t rows = 0;
    int f1 = f;
    while (f1 < fv.n_frags && (f1 == f || rows + fv.h_num_rows[f1] <= h.chunk_rows)) {
      rows += fv.h_num_rows[f1];
      ++f1;
    }
    hipError_t e = hipMemsetAsync(spill_base, 0, 256, s);
    if (e != hipSuccess) return e;
    if (ev_pool && ev_i + 1 < st->n_ev) (void)hipEventRecord(ev_pool[ev_i], s);
    e = idx_launch_scatter<NK, NV, RS, PK>(h, fv, f, f1 - f, recs, cnt, sl, s);
    if (e != hipSuccess) return e;
    if (ev_pool && ev_i + 1 < st->n_ev) {
      (void)hipEventRecord(ev_pool[ev_i + 1], s);
      ev_i += 2;
    }
    st->n_launches += 1;
    if constexpr (PK) {
      e = idx_aggregate_pk<NV>(h, p, recs, cnt, out, n_cus, s);
    } else {
      e = (NV > 0 && h.g.mm) ? idx_launch_aggregate<NK, NV, (NV > 0), RS>(h, p, recs, cnt, out, n_cus, s)
                             : idx_launch_aggregate<NK, NV, false, RS>(h, p, recs, cnt, out, n_cus, s);
    }
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_idx_spill<NK>, dim3(256), dim3(256), 0, s, h.g, h.c, p, sl, out);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    f = f1;
  }
  st->spill_counter32 = (uint32_t*)spill_base;
  st->n_events_used = ev_i;
  return hipSuccess;
}

}  // namespace

bool idx_part_eligible(const DevPlan& p, const FragView& fv, int n_cus) {
  IdxPlanHost h;
  return make_idx_plan(p, fv, n_cus, kDefaultScratchCap, &h);
}

int64_t idx_part_scratch_bytes(const DevPlan& p, const FragView& fv, int n_cus, int64_t cap_bytes) {
  IdxPlanHost h;
  if (!make_idx_plan(p, fv, n_cus, cap_bytes, &h)) return 0;
  return h.scratch_bytes + 64;
}

hipError_t launch_idx_partitioned(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, void* scratch,
                                  int64_t scratch_bytes, int64_t cap_bytes, int n_cus, hipStream_t s, LaunchStats* st) {
  IdxPlanHost h;
  if (!make_idx_plan(p, fv, n_cus, cap_bytes, &h)) return hipErrorInvalidValue;
  if (h.scratch_bytes + 64 > scratch_bytes) return hipErrorInvalidValue;
  st->kernel_name = "k_idx_scatter";
  st->variant = h.g.pk ? (h.g.rs == 3 ? 8 : 7) : 6;  // 6: plain records; 7 / 8: the packed 4- / 2-byte word
  st->n_launches = 0;
  // the plain record of NV value columns, or the packed word (2 or 4 bytes)
#define MQ_IDX_RUN(NK, NV)                                                                                     \
  do {                                                                                                         \
    if (h.g.pk && h.g.rs == 3) return idx_run<NK, NV, 3, true>(h, p, fv, out, d_err, scratch, n_cus, s, st);    \
    if (h.g.pk && NV > 0) return idx_run<NK, NV, (NV > 0 ? 2 : 3), true>(h, p, fv, out, d_err, scratch, n_cus, s, st); \
    return idx_run<NK, NV, (NV == 0 ? 2 : NV == 1 ? 1 : 0), false>(h, p, fv, out, d_err, scratch, n_cus, s, st);       \
  } while (0)
  switch (h.g.nk * 10 + h.g.nv) {
    case 10: MQ_IDX_RUN(1, 0);
    case 20: MQ_IDX_RUN(2, 0);
    case 30: MQ_IDX_RUN(3, 0);
    case 11: MQ_IDX_RUN(1, 1);
    case 12: MQ_IDX_RUN(1, 2);
    case 13: MQ_IDX_RUN(1, 3);
    case 21: MQ_IDX_RUN(2, 1);
    case 22: MQ_IDX_RUN(2, 2);
    case 23: MQ_IDX_RUN(2, 3);
    case 31: MQ_IDX_RUN(3, 1);
    case 32: MQ_IDX_RUN(3, 2);
    default: MQ_IDX_RUN(3, 3);
  }
#undef MQ_IDX_RUN
}

}  // namespace mq
