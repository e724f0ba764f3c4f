// kernels_part.hip — partition-then-aggregate GROUP BY for high-cardinality int64 keys
// (SURVEY cfg3 / cfg3-filtered, the headline workload).
//
// Why: with 10 M groups the output table (20 M x 32 B = 640 MB) is larger than every on-chip
// memory, so updating it row by row (the reference's GPU path: one CAS + atomics per row,
// cuda_mapd_rt.cu:180-246,436-545) turns every row into random HBM read-modify-writes.
// Here the scan range-partitions the surviving rows by the HOME SLOT the reference would
// probe first,  home = MurmurHash3(key) % entry_count  (GroupByRuntime.cpp:20-48), a second
// kernel aggregates each home-slot range in a per-CU LDS table, places the groups with the
// reference's own linear probing rule and writes the range of the output table with plain
// coalesced stores.  No global atomic is issued per row, and the finished buffer is a valid
// linear-probing image of get_group_value: every key sits at or after its home slot with no
// empty slot in between.
//
//   phase 1  k_part_scatter    12 producer waves per workgroup stream the columns (16 B/lane
//                              non-temporal loads, next tile prefetched into registers), filter,
//                              hash, and append the 16-byte record {key, value bits} to this
//                              workgroup's private run of partition p through an LDS staging
//                              line of L = 8192 / P records (128 B at P = 1024, the smallest
//                              write MI355X's HBM absorbs at streaming efficiency —
//                              tools/microbench/scatter.hip); 4 flusher waves send full lines
//                              to the runs as coalesced stores.  Producers and flushers meet
//                              only through three LDS words per partition: no barriers.
//   phase 2  k_part_aggregate  one workgroup per (partition, sub-range): two-choice bucketized
//                              LDS table {key, partial slots} fed with ds atomics; sub-ranges
//                              (R > 1) re-read the partition's runs and keep their own home
//                              range, so an LDS table only ever holds entry_count / (2 P R)
//                              groups.  Emission claims canonical slots in an LDS bitmap and
//                              stores rows + empty rows; a later chunk first re-loads its
//                              range (merge).
//   phase 3  k_spill_merge     the few groups that probe past the end of their range, and
//                              every record that met a full run / full LDS table, are kept
//                              as partial rows in a spill list and merged last with the
//                              reference's CAS insert-or-find, so skew costs speed, never
//                              correctness.
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "fast_common.h"

namespace mq {

using namespace fast;

namespace {

constexpr int kPartBlock = 1024;
constexpr int kStageRecs = 8192;       // staged records per workgroup (128 KB of LDS)
constexpr int kSegRecs = 8;            // records per 128-byte segment (one owner lane each)
constexpr int kMaxInt = 8;             // internal (LDS) partial slots per group
constexpr int kMaxSub = 4;             // sub-ranges per partition
// phase 2 pair counters: 4 KB of rendezvous words (one per workgroup), then one progress word per wave
constexpr int kPairCtrBytes = 4096 + 256 * 16 * 4;
constexpr uint32_t kSpillMin = 1u << 22;  // spill list: at least 4 M partial rows, else 1/16 of a chunk's rows

struct alignas(16) Rec {
  int64_t key;
  int64_t val;
};

// spill list entry: {key, ns_int partial slots} = (1 + ns_int) x 8 bytes, packed

// home slot arithmetic with plan-time reciprocals: q' = mulhi(x, floor(2^32 / d)) is
// floor(x / d) or one less (x / d - x * floor(2^32 / d) / 2^32 < x / 2^32 < 1), so one
// conditional subtraction makes the remainder exact — 2 multiplies instead of a 64-bit
// division
struct HomeMap {
  uint32_t d;        // entry_count
  uint32_t S1;       // home slots per partition
  uint32_t S2;       // home slots per sub-range
  uint32_t R;        // sub-ranges per partition
  uint32_t d_rcp;    // floor(2^32 / d)   (d >= 2)
  uint32_t s1_rcp;   // floor(2^32 / S1)  (S1 >= 2)
};

MQ_D uint32_t home_from_hash(const HomeMap& m, uint32_t h) {
  uint32_t r = h - __umulhi(h, m.d_rcp) * m.d;
  if (r >= m.d) r -= m.d;
  return r;
}
MQ_D uint32_t home_of(const HomeMap& m, int64_t key) {
  return home_from_hash(m, murmur3_u64((uint64_t)key));
}
MQ_D uint32_t part_of(const HomeMap& m, uint32_t home) {
  const uint32_t q = __umulhi(home, m.s1_rcp);
  return q + (home - q * m.S1 >= m.S1 ? 1u : 0u);
}

// Record identity.  Phase 1 has just computed h = MurmurHash3(key) (it names the partition); phase 2
// needs h again for the home slot and the LDS buckets, and was spending a third of its vector
// instructions recomputing it for every record it visits (twice per record with two sub-ranges).
// MurmurHash3_x86_32 over the two words of an int64 key is, for a FIXED low word, a bijection of the
// high word (every step — multiply by an odd constant, rotate, xor, x*5+c, fmix32 — is invertible),
// so the pair (low word, h) identifies the key exactly: records and the LDS tables carry that
// 64-bit "kid" = h << 32 | low word, phase 2 reads h out of it, and the key is rebuilt where a row
// leaves for the output table or the spill list (once per group, not once per record).
MQ_D int64_t kid_of(int64_t key, uint32_t h) { return (int64_t)(((uint64_t)h << 32) | (uint64_t)(uint32_t)key); }
MQ_D uint32_t kid_hash(int64_t kid) { return (uint32_t)((uint64_t)kid >> 32); }
MQ_D int64_t key_of_kid(int64_t kid) {
  const uint32_t lo = (uint32_t)kid;
  uint32_t h = kid_hash(kid);
  // undo fmix32 (x ^= x >> 16 is its own inverse; x ^= x >> 13 needs the 26-bit term as well)
  h ^= h >> 16;
  h *= 0x7ed1b41du;  // 0xc2b2ae35^-1 mod 2^32
  h ^= (h >> 13) ^ (h >> 26);
  h *= 0xa5cb9243u;  // 0x85ebca6b^-1
  h ^= h >> 16;
  h ^= 8u;           // the length
  // h = rotl(h1 ^ k2', 13) * 5 + 0xe6546b64 with h1 = the state after the low word's block
  h = (h - 0xe6546b64u) * 0xcccccccdu;  // 5^-1
  h = (h >> 13) | (h << 19);
  uint32_t k1 = lo * 0xcc9e2d51u;
  k1 = rotl32(k1, 15) * 0x1b873593u;
  const uint32_t h1 = rotl32(k1, 13) * 5u + 0xe6546b64u;
  uint32_t k2 = (h ^ h1) * 0x56ed309bu;  // 0x1b873593^-1
  k2 = ((k2 >> 15) | (k2 << 17)) * 0xdee13bb1u;  // 0xcc9e2d51^-1
  return (int64_t)(((uint64_t)k2 << 32) | lo);
}

struct PartGeom {
  int32_t P, lgL, B;   // partitions, log2(records per staged line), scatter workgroups
  uint32_t L;          // records per line = kStageRecs / P
  uint32_t cap;        // records per (workgroup, partition) run, multiple of L
  uint32_t E;          // LDS table entries in phase 2
  uint32_t b_mult;     // floor((E / 4) * 2^32 / S2): monotone map home-offset -> LDS bucket
  int32_t ns_int;      // distinct partial slots kept per group in LDS
  uint32_t lds_table_bytes;  // keys + slot arrays of the LDS table (16-byte multiple)
  uint32_t slot_off[kMaxInt];  // byte offset of each internal slot array in LDS
  HomeMap hm;
};

struct PartSlots {
  int32_t int_op[kMaxInt];            // op of each internal slot
  int32_t out_map[MI355Q_MAX_SLOTS];  // output slot -> internal slot (-1: key / none)
  int64_t int_init[kMaxInt];          // identity of each internal slot
  // nullable value column: NULL inputs only count for COUNT(*); nn_slot is the internal
  // COUNT_NN slot (non-NULL values), through which "no value seen" turns the output slots
  // flagged SlotProg::null_init back into the NULL sentinel on emission
  int32_t val_nullable;
  int32_t nn_slot;
  int32_t nn_hidden;                  // nn_slot is not mapped by any output slot
  int32_t pad_;
  int64_t null_bits;                  // the value column's NULL as loaded
};

MQ_D int64_t op_identity_dev(int op) {
  switch (op) {
    case SO_MIN_I: return INT64_MAX;
    case SO_MAX_I: return INT64_MIN;
    case SO_MIN_F: return 0x7fefffffffffffffll;            // DBL_MAX
    case SO_MAX_F: return (int64_t)0xffefffffffffffffull;  // -DBL_MAX
    default: return 0;
  }
}
// partial row of ONE raw record for internal slot `op`
MQ_D int64_t raw_partial(int op, int64_t vb, bool is_null) {
  if (op == SO_COUNT) return 1;
  if (op == SO_COUNT_NN) return is_null ? 0 : 1;
  return is_null ? op_identity_dev(op) : vb;
}

struct TableArgs {
  int64_t* out;
  uint32_t entry_count;
  int32_t row_quad;
  SlotProg sp;
  int64_t init[MI355Q_MAX_SLOTS];
};

// Slice-merge mode of phase 2 (multi-device merge): no records; every unit of this rank's
// home range folds the rows of the same table range out of `n_src` tables — the slices the peers sent —
// into its LDS table, plus (the last unit) the pad rows that followed each slice.  Rows are partial rows
// like the ones a later chunk re-loads, so the merge-load code is the same; a row whose home slot is in
// the range but not in the unit goes to the spill list (merged by k_spill_merge), a row whose home slot
// is outside [lo, hi) belongs to another rank and is dropped.
constexpr int kMaxMergeSrc = 16;
struct SliceMerge {
  const int64_t* src[kMaxMergeSrc];   // src[i] + (slot - lo) * row_quad = row `slot` of table i, slot in [lo, hi)
  const int64_t* pads[kMaxMergeSrc];  // pad_rows rows that followed slice i (slots hi, hi + 1, ... wrapping)
  int32_t n_src, pad_rows;
  uint32_t lo, hi;
};

struct SpillList {
  uint32_t* count;     // device word
  int64_t* entries;    // [cap][1 + ns_int]
  int32_t* d_err;      // [0] reference error code, [1] spill list overflow
  uint32_t cap;
  int32_t stride;      // quads per entry = 1 + ns_int
};

// LDS-only barrier: waits for this wave's LDS traffic, NOT for its outstanding global loads,
// so the next tile's column loads stay in flight across the two barriers of a round.
MQ_D void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

MQ_D void spill_append(const SpillList& sl, int64_t key, const int64_t* part, int n) {
  const uint32_t i = atomicAdd(sl.count, 1u);
  if (i >= sl.cap) {
    atomicExch(sl.d_err + 1, 1);
    return;
  }
  int64_t* e = sl.entries + (size_t)i * sl.stride;
  e[0] = key;
  for (int j = 0; j < n; ++j) e[1 + j] = part[j];
}
MQ_D void spill_record(const SpillList& sl, const PartSlots& ps, int ns, int64_t key, int64_t vb) {
  int64_t part[kMaxInt];
  const bool is_null = ps.val_nullable && vb == ps.null_bits;
  for (int j = 0; j < kMaxInt; ++j) part[j] = raw_partial(ps.int_op[j], vb, is_null);
  spill_append(sl, key, part, ns);
}

template <typename VT>
MQ_D int64_t val_bits_of(VT v);
template <>
MQ_D int64_t val_bits_of<int64_t>(int64_t v) { return v; }
template <>
MQ_D int64_t val_bits_of<int32_t>(int32_t v) { return v; }
template <>
MQ_D int64_t val_bits_of<double>(double v) { return dbl_bits(v); }
template <>
MQ_D int64_t val_bits_of<none_t>(none_t) { return 0; }


// ------------------------------------------------------------------------- phase 1
// What phase 1 needs of the plan, kept small so it stays in SGPRs.
struct ScatterArgs {
  int32_t P, lgL, B;
  uint32_t L, cap;
  HomeMap hm;
  int32_t ns_int;
  uint32_t ops_packed;  // internal slot ops, 4 bits each (for the partial row of a spilled record)
  int32_t n_cand;       // heavy-hitter candidate slots (power of two)
  int32_t val_nullable;
  int64_t null_bits;
  int64_t kmin;         // DIRECT partitioning (join probes): partition = (key - kmin) / S1
};

template <typename FT, typename VT>
struct Tile {
  Quad<FT> f;
  Quad<int64_t> k;
  Quad<VT> v;
  int valid;  // rows of this lane's quad that exist (0..4)
};

template <typename FT, typename VT>
MQ_D void load_tile(const int8_t* const* __restrict__ cols, const int64_t* __restrict__ num_rows,
                    int n_cols, int fcol, int kcol, int vcol, int f, int64_t tile_in_frag,
                    Tile<FT, VT>& t) {
  const int8_t* const* fc = cols + (size_t)f * n_cols;
  const int64_t n = num_rows[f];
  const int64_t q = tile_in_frag * kPartBlock + threadIdx.x;
  const int64_t r0 = q << 2;
  t.valid = 0;
  if (r0 >= n) return;
  const int8_t* fb = is_none<FT>::value ? nullptr : fc[fcol];
  const int8_t* kb = fc[kcol];
  const int8_t* vb = is_none<VT>::value ? nullptr : fc[vcol];
  if (r0 + 4 <= n) {
    load_quad<FT>(fb, q, t.f);
    load_quad<int64_t>(kb, q, t.k);
    load_quad<VT>(vb, q, t.v);
    t.valid = 4;
  } else {
    t.valid = (int)(n - r0);
    for (int i = 0; i < 4; ++i) {
      if (i < t.valid) {
        quad_set(t.f, i, load_one<FT>(fb, r0 + i));
        t.k.v[i] = load_one<int64_t>(kb, r0 + i);
        quad_set(t.v, i, load_one<VT>(vb, r0 + i));
      }
    }
  }
}

// range filter on the column's own width (an int32 column is compared with 32-bit bounds;
// make_range_filter clamps lo/hi to the column type)
template <typename T>
MQ_D bool filter_pass_narrow(const RangeFilter& f, T v) { return filter_pass<T>(f, v); }
template <>
MQ_D bool filter_pass_narrow<int32_t>(const RangeFilter& f, int32_t v) {
  bool in = v >= (int32_t)f.lo && v <= (int32_t)f.hi;
  if (f.negate) in = !in;
  if (f.nullable && v == (int32_t)f.null_val) in = false;
  return in;
}

typedef int v4i32_t __attribute__((ext_vector_type(4)));
MQ_D void store_rec_nt(Rec* dst, const Rec& r) {
  v4i32_t x;
  x.x = (int)(uint32_t)r.key;
  x.y = (int)(uint32_t)((uint64_t)r.key >> 32);
  x.z = (int)(uint32_t)r.val;
  x.w = (int)(uint32_t)((uint64_t)r.val >> 32);
  __builtin_nontemporal_store(x, (v4i32_t*)dst);
}

MQ_D Rec load_rec_nt(const Rec* src) {
  const v4i32_t x = __builtin_nontemporal_load((const v4i32_t*)src);
  Rec r;
  r.key = (int64_t)(((uint64_t)(uint32_t)x.y << 32) | (uint64_t)(uint32_t)x.x);
  r.val = (int64_t)(((uint64_t)(uint32_t)x.w << 32) | (uint64_t)(uint32_t)x.z);
  return r;
}

// Cooperative flush of 128-byte segments: `need` marks the lanes whose own segment (index
// seg_base + lane) must leave; 8 lanes write one segment, 8 segments per wave instruction.
// dst_rec = record index in `scratch` of the segment's first record (valid where need).
MQ_D void flush_segments(bool need, uint32_t dst_rec, const Rec* __restrict__ stage, int seg_base,
                         Rec* __restrict__ scratch) {
  const uint64_t mask = __ballot(need);
  if (!mask) return;
  const int lane = threadIdx.x & 63;
  const int n_need = __popcll(mask);
  const int rank_need = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  // forward permute: lane j < n_need receives the lane id of the j-th needed segment
  const int dest = need ? rank_need : n_need + (lane - rank_need);
  const int owner_of_rank = __builtin_amdgcn_ds_permute(dest << 2, lane);
  for (int it = 0; it * 8 < n_need; ++it) {
    const int j = it * 8 + (lane >> 3);
    const int owner = __shfl(owner_of_rank, j & 63, 64);
    const uint32_t o = (uint32_t)__shfl((int)dst_rec, owner, 64);
    if (j < n_need) {
      const Rec r = stage[(size_t)(seg_base + owner) * kSegRecs + (lane & 7)];
      store_rec_nt(scratch + (size_t)o + (lane & 7), r);
    }
  }
}

// Spill list positions for phase 1 are handed out from workgroup-private blocks of
// kSpillBlock entries: one global atomic per block instead of one per record (a skewed input
// spills millions of records; one shared counter would serialise them at ~12 ns each).
// blk = {next, end} packed in one LDS word; ~0 = a lane is fetching the next block.
constexpr uint32_t kSpillBlock = 256;
constexpr unsigned long long kSpillBusy = ~0ull;
MQ_D uint32_t spill_slot_from_block(const SpillList& sl, unsigned long long* blk) {
  for (;;) {
    const unsigned long long w = __hip_atomic_load(blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (w == kSpillBusy) continue;
    const uint32_t next = (uint32_t)w, end = (uint32_t)(w >> 32);
    if (next < end) {
      if (atomicCAS(blk, w, ((unsigned long long)end << 32) | (next + 1)) == w) return next;
    } else if (atomicCAS(blk, w, kSpillBusy) == w) {
      // this lane fetches the next block and publishes it before leaving the loop (the other
      // lanes of the wave spin in this very loop)
      const uint32_t base = atomicAdd(sl.count, kSpillBlock);
      __hip_atomic_store(blk, ((unsigned long long)(base + kSpillBlock) << 32) | (base + 1), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
      return base;
    }
  }
}
MQ_D void spill_raw(const SpillList& sl, const ScatterArgs& g, unsigned long long* blk, int64_t key, int64_t vb) {
  const uint32_t i = spill_slot_from_block(sl, blk);
  if (i >= sl.cap) {
    atomicExch(sl.d_err + 1, 1);
    return;
  }
  const bool is_null = g.val_nullable && vb == g.null_bits;
  int64_t* e = sl.entries + (size_t)i * sl.stride;
  e[0] = key;
  for (int j = 0; j < g.ns_int; ++j) e[1 + j] = raw_partial((int)((g.ops_packed >> (4 * j)) & 15u), vb, is_null);
}

// Role split inside the 16-wave workgroup: waves 0..11 PRODUCE (load, filter, hash, take a
// stream position, write the record into the partition's open staging line), waves 12..15
// FLUSH (poll the per-partition counters, send complete lines to the runs, open the next
// line).  The two sides only meet through three LDS words per partition — there is no
// workgroup barrier in the steady state, so one wave's LDS / HBM latency never stalls another
// (the barrier version of this kernel spent ~17 % of its time parked in s_barrier).
//   cursor[p]   stream positions handed out               (producers: atomicAdd)
//   written[p]  records that have reached the staging area (producers: atomicAdd after the write)
//   flushed[p]  lines already sent to the run = index of the open line (flushers: store after
//               the line has been read)
// A record may only be staged while its line is the open one; otherwise the lane keeps it
// pending (<= 4 per lane) and retries on its next tile.  LDS executes one wave's instructions
// in order, so "written counts it" implies "the record is there", and "flushed moved on"
// implies "the old line has been read".
constexpr int kProdWaves = 12;
constexpr int kFlushWaves = kPartBlock / 64 - kProdWaves;
constexpr int kSegPerFlusher = (kStageRecs / kSegRecs) / (kFlushWaves * 64);  // 4
constexpr int kHotSlots = 256;          // per-workgroup heavy-hitter table (keys with >~0.4 % of the rows)
constexpr uint32_t kHotPromote = 16;    // net sampled sightings (Misra-Gries) before a key is declared hot
constexpr int kHotSampleLg = 4;         // 1 record in 16 votes
constexpr uint32_t kMaxSpins = 1u << 20;  // ~1 s of s_sleep: only a broken protocol gets there

MQ_D uint32_t lds_peek(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
MQ_D void lds_poke(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename FT, typename VT>
MQ_D void load_wave_tile(const int8_t* const* __restrict__ cols, const int64_t* __restrict__ num_rows,
                         int n_cols, int fcol, int kcol, int vcol, int f, int64_t quad, Tile<FT, VT>& t) {
  const int8_t* const* fc = cols + (size_t)f * n_cols;
  const int64_t n = num_rows[f];
  const int64_t r0 = quad << 2;
  t.valid = 0;
  if (r0 >= n) return;
  const int8_t* fb = is_none<FT>::value ? nullptr : fc[fcol];
  const int8_t* kb = fc[kcol];
  const int8_t* vb = is_none<VT>::value ? nullptr : fc[vcol];
  if (r0 + 4 <= n) {
    load_quad<FT>(fb, quad, t.f);
    load_quad<int64_t>(kb, quad, t.k);
    load_quad<VT>(vb, quad, t.v);
    t.valid = 4;
  } else {
    t.valid = (int)(n - r0);
    for (int i = 0; i < 4; ++i) {
      if (i < t.valid) {
        quad_set(t.f, i, load_one<FT>(fb, r0 + i));
        t.k.v[i] = load_one<int64_t>(kb, r0 + i);
        quad_set(t.v, i, load_one<VT>(vb, r0 + i));
      }
    }
  }
}

// one raw record folded into a heavy hitter's partial row (8-byte LDS slots, internal op order)
MQ_D void hot_apply(int op, int64_t* s, int64_t vb, bool is_null) {
  if (op == SO_COUNT) {
    atomicAdd((unsigned long long*)s, 1ull);
    return;
  }
  if (is_null) return;
  switch (op) {
    case SO_COUNT_NN: atomicAdd((unsigned long long*)s, 1ull); break;
    case SO_SUM_I: atomicAdd((unsigned long long*)s, (unsigned long long)vb); break;
    case SO_SUM_F: atomicAdd((double*)s, bits_dbl(vb)); break;
    case SO_MIN_I: atomicMin((long long*)s, (long long)vb); break;
    case SO_MAX_I: atomicMax((long long*)s, (long long)vb); break;
    case SO_MIN_F: a_minmax_f64<true, false, false>(s, bits_dbl(vb), 0.0); break;
    case SO_MAX_F: a_minmax_f64<true, true, false>(s, bits_dbl(vb), 0.0); break;
    default: break;
  }
}

// MODE 0: records are partitioned by their home slot in the group-by table (MurmurHash3) and carry kids;
// MODE 1 (DIRECT): by key range, `(key - kmin) / S1`, keys outside [kmin, kmin + hm.d) are dropped —
// the partitions of a radix join probe (k_part_join / k_part_probe), whose slices then fit LDS / L2;
// MODE 2: by the home slot of a KEYED join table, MurmurHash1(key) % entries (baseline_hash_join_idx,
// JoinHashTableQueryRuntime.cpp:56-94), records carry the key (k_part_probe_keyed).
template <typename FT, typename VT, int MODE = 0>
__global__ __launch_bounds__(kPartBlock) void k_part_scatter(
    const int8_t* const* __restrict__ cols, const int64_t* __restrict__ num_rows, int n_frags,
    int n_cols, RangeFilter flt, int kcol, int vcol, ScatterArgs g, Rec* __restrict__ scratch,
    uint32_t* __restrict__ cnt, SpillList sl) {
  constexpr bool DIRECT = MODE == 1;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  Rec* stage = (Rec*)smem_raw;                                      // [P][L] = kStageRecs records
  uint32_t* cursor = (uint32_t*)(smem_raw + kStageRecs * sizeof(Rec));  // [P]
  uint32_t* written = cursor + g.P;                                 // [P]
  uint32_t* flushed = written + g.P;                                // [P]
  uint32_t* done = flushed + g.P;                                   // producer waves finished
  uint32_t* n_hot = done + 1;                                       // keys promoted so far
  unsigned long long* sp_blk = (unsigned long long*)(((uintptr_t)(n_hot + 1) + 7) & ~(uintptr_t)7);  // spill block {next, end}
  // Heavy hitters.  A key that owns a sizeable share of the rows would overflow its partition's
  // runs and throttle everybody on one staging line, so each workgroup keeps a small table of
  // such keys and folds their records into partial rows right here (spilled once, at the end).
  // Detection is a per-slot Misra-Gries counter fed by 1 record in 16: a key is promoted after
  // kHotPromote net sampled sightings in its slot — uniform keys never get there.  Until the
  // first promotion the per-record cost is zero (one LDS word read per wave tile).  Promotion
  // only changes where LATER records of the key go, so exactness does not depend on the heuristic.
  int64_t* hot_key = (int64_t*)(smem_raw + ((kStageRecs * sizeof(Rec) + (size_t)g.P * 12 + 8 + 15) & ~(size_t)15));
  int64_t* hot_slot = hot_key + kHotSlots;                          // [ns_int][kHotSlots]
  // candidates: {16-bit key fingerprint, 16-bit vote count} per slot, many more slots than
  // the hot table so that a key with ~0.05 % of the rows still out-votes its slot's noise
  uint32_t* cand = (uint32_t*)(hot_slot + (size_t)g.ns_int * kHotSlots);  // [g.n_cand]
  const int t = threadIdx.x, b = blockIdx.x, G = gridDim.x;
  const int wave = t >> 6, lane = t & 63;
  const int lgL = g.lgL;
  const uint32_t Lm1 = g.L - 1;
  for (int i = t; i < 3 * g.P + 2; i += kPartBlock) cursor[i] = 0;
  for (int i = t; i < g.n_cand; i += kPartBlock) cand[i] = 0;
  if (t == 0) *sp_blk = 0;  // next == end: the first spill fetches a block
  for (int i = t; i < kHotSlots; i += kPartBlock) {
    hot_key[i] = kEmptyKey64;
    for (int m = 0; m < g.ns_int; ++m)
      hot_slot[m * kHotSlots + i] = op_identity_dev((int)((g.ops_packed >> (4 * m)) & 15u));
  }
  __syncthreads();

  if (wave < kProdWaves) {
    // ------------------------------------------------------------------ producers
    constexpr int64_t kSuperQuads = (int64_t)kProdWaves * 64;  // quads per workgroup step
    int64_t c_key[4], c_val[4];
    uint32_t c_pid[4], c_slot[4];
    uint32_t c_mask = 0;
    uint32_t tile_no = 0;
    // flattened walk over the fragments in super-tiles of 12 wave-tiles; workgroup b owns the
    // super-tiles == b (mod G), wave w the w-th wave-tile of each
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
    // stage one record if its line is open; true = the record has left this lane
    auto try_stage = [&](int64_t key, int64_t vb, uint32_t p, uint32_t s) -> bool {
      if (lds_peek(&flushed[p]) != (s >> lgL)) return false;
      asm volatile("" ::: "memory");  // compiler order only: the LDS itself runs a wave in order
      stage[((size_t)p << lgL) + (s & Lm1)] = Rec{key, vb};
      asm volatile("" ::: "memory");
      atomicAdd(&written[p], 1u);
      return true;
    };
    auto retry_pending = [&]() {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((c_mask & (1u << i)) && try_stage(c_key[i], c_val[i], c_pid[i], c_slot[i])) c_mask &= ~(1u << i);
      }
    };
    seek();
    Tile<FT, VT> cur;
    cur.valid = 0;
    bool have = f < n_frags;
    if (have) load_wave_tile<FT, VT>(cols, num_rows, n_cols, flt.col, kcol, vcol, f, quad_of(), cur);
    while (have) {
      gt += G;
      seek();
      const bool have_next = f < n_frags;
      Tile<FT, VT> nxt;
      nxt.valid = 0;
      if (have_next) load_wave_tile<FT, VT>(cols, num_rows, n_cols, flt.col, kcol, vcol, f, quad_of(), nxt);

      retry_pending();
      // Heavy-hitter work is wave-uniform: a tile looks at the hot table only once a key has been
      // promoted, and only every 16th tile of a wave votes — all other tiles run the plain path
      // with no per-record cost.
      ++tile_no;
      const bool any_hot = lds_peek(n_hot) != 0;
      const bool voting = (tile_no & ((1u << kHotSampleLg) - 1)) == 1;  // tiles 1, 17, 33, ...: a wave's first tile votes
      auto one_row = [&](int i, auto hot_aware) {
        bool park = false;
        uint32_t p = 0, s = 0;
        int64_t vb = 0, rk = 0;  // rk: what the record carries — the key (DIRECT) or its kid
        if (i < cur.valid && filter_pass_narrow<FT>(flt, quad_get(cur.f, i)) &&
            (!DIRECT || (uint64_t)cur.k.v[i] - (uint64_t)g.kmin < (uint64_t)g.hm.d)) {
          const int64_t key = cur.k.v[i];
          const uint32_t h = MODE == 2 ? murmur1_u64((uint64_t)key) : murmur3_u64((uint64_t)key);
          vb = val_bits_of<VT>(quad_get(cur.v, i));
          bool folded = false;
          if (decltype(hot_aware)::value) {
            const uint32_t hs = (h >> 9) & (kHotSlots - 1);
            const int64_t hk = *(volatile int64_t*)&hot_key[hs];
            if (hk == key) {  // heavy hitter: fold the record into its partial row
              const bool is_null = g.val_nullable && vb == g.null_bits;
              for (int m = 0; m < g.ns_int; ++m)
                hot_apply((int)((g.ops_packed >> (4 * m)) & 15u), &hot_slot[m * kHotSlots + hs], vb, is_null);
              folded = true;
            } else if (voting && hk == kEmptyKey64) {  // hot slot still free: Misra-Gries vote
              const uint32_t cs = h & (uint32_t)(g.n_cand - 1);
              const uint32_t fp = (h >> 16) << 16;
              const uint32_t c = *(volatile uint32_t*)&cand[cs];
              if ((c & 0xffff0000u) == fp && (c & 0xffffu)) {
                const uint32_t votes = (atomicAdd(&cand[cs], 1u) + 1) & 0xffffu;
                if (votes >= kHotPromote && votes < 0x8000u &&
                    atomicCAS((unsigned long long*)&hot_key[hs], (unsigned long long)kEmptyKey64,
                              (unsigned long long)key) == (unsigned long long)kEmptyKey64)
                  atomicAdd(n_hot, 1u);
              } else if ((c & 0xffffu) <= 1 || (c & 0xffffu) >= 0x8000u) {
                *(volatile uint32_t*)&cand[cs] = fp | 1u;  // the old candidate ran out of votes
              } else {
                atomicSub(&cand[cs], 1u);
              }
            }
          }
          if (!folded) {
            p = part_of(g.hm, DIRECT ? (uint32_t)((uint64_t)key - (uint64_t)g.kmin) : home_from_hash(g.hm, h));
            rk = MODE != 0 ? key : kid_of(key, h);
            // (the one key whose kid is the LDS tables' empty mark takes the spill list, like a record
            // that meets a full run)
            s = (MODE == 0 && rk == kEmptyKey64) ? g.cap : atomicAdd(&cursor[p], 1u);
            if (s >= g.cap) spill_raw(sl, g, sp_blk, key, vb);
            else park = !try_stage(rk, vb, p, s);
          }
        }
        // Line not open yet: park the record in pending slot i.  If an older record still
        // waits there, the WHOLE wave waits for the flushers and keeps retrying every
        // pending record of every lane (never spin inside divergent code: the idle lanes of
        // this wave may hold exactly the records another line is waiting for).
        uint32_t spins = 0;
        while (__any(park && (c_mask & (1u << i)))) {
          retry_pending();
          // the record in hand is retried too: its line may have opened meanwhile, and another
          // wave may be waiting for exactly that line to fill
          if (park && try_stage(rk, vb, p, s)) park = false;
          __builtin_amdgcn_s_sleep(2);
          if (++spins > kMaxSpins) {  // cannot happen unless the protocol is broken: bail out
            atomicExch(sl.d_err + 1, 2);
            c_mask = 0;
            break;
          }
        }
        if (park) {
          c_key[i] = rk;
          c_val[i] = vb;
          c_pid[i] = p;
          c_slot[i] = s;
          c_mask |= 1u << i;
        }
      };
      if (any_hot || voting) {
#pragma unroll
        for (int i = 0; i < 4; ++i) one_row(i, std::true_type{});
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) one_row(i, std::false_type{});
      }
      cur = nxt;
      have = have_next;
    }
    // drain this lane's pending records, then report the wave finished
    for (uint32_t spins = 0; __any(c_mask != 0); ++spins) {
      retry_pending();
      __builtin_amdgcn_s_sleep(2);
      if (spins > kMaxSpins) {
        atomicExch(sl.d_err + 1, 3);
        break;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) atomicAdd(done, 1u);
    return;
  }

  // -------------------------------------------------------------------- flushers
  // Flusher lane `fid` owns segments fid + 256 k (k = 0..3); in pass k a wave covers 64
  // consecutive segments, so the segments of one line (<= 64) always share a pass and a wave.
  const int fid = (wave - kProdWaves) * 64 + lane;
  const int lgSpl = lgL - 3;
  uint32_t my_fl[kSegPerFlusher];
  for (int k = 0; k < kSegPerFlusher; ++k) my_fl[k] = 0;
  uint32_t idle = 0;
  for (;;) {
    const bool all_done = lds_peek(done) == (uint32_t)kProdWaves;  // read BEFORE the scan
    bool any_work = false;
#pragma unroll
    for (int k = 0; k < kSegPerFlusher; ++k) {
      const int seg = k * (kFlushWaves * 64) + fid;
      const int p = seg >> lgSpl;
      const uint32_t sidx = (uint32_t)seg & ((1u << lgSpl) - 1);
      const uint32_t w = lds_peek(&written[p]);
      const bool need = (w >> lgL) > my_fl[k] && (my_fl[k] << lgL) < g.cap;
      asm volatile("" ::: "memory");
      flush_segments(need, ((uint32_t)p * g.B + b) * g.cap + (my_fl[k] << lgL) + sidx * kSegRecs, stage,
                     seg - lane, scratch);
      asm volatile("" ::: "memory");
      if (need) {
        my_fl[k] += 1;
        if (sidx == 0) lds_poke(&flushed[p], my_fl[k]);  // after this wave's reads of the line
      }
      any_work |= need;
    }
    if (!__any(any_work)) {
      if (all_done) break;
      __builtin_amdgcn_s_sleep(4);
      if (++idle > kMaxSpins) {  // producers stuck: give up rather than hang the device
        atomicExch(sl.d_err + 1, 4 + (int)lds_peek(done) * 16);
        break;
      }
    } else {
      idle = 0;
    }
  }
  // every producer is done and every complete line is out: drain the partial open lines
#pragma unroll
  for (int k = 0; k < kSegPerFlusher; ++k) {
    const int seg = k * (kFlushWaves * 64) + fid;
    const int p = seg >> lgSpl;
    const uint32_t sidx = (uint32_t)seg & ((1u << lgSpl) - 1);
    const uint32_t w = lds_peek(&written[p]);
    const uint32_t rem = w - (my_fl[k] << lgL);  // < L
    const bool need = sidx * kSegRecs < rem && (my_fl[k] << lgL) + sidx * kSegRecs < g.cap;
    flush_segments(need, ((uint32_t)p * g.B + b) * g.cap + (my_fl[k] << lgL) + sidx * kSegRecs, stage,
                   seg - lane, scratch);
    if (sidx == 0) {
      const uint32_t c = lds_peek(&cursor[p]);
      cnt[(size_t)p * g.B + b] = c < g.cap ? c : g.cap;
    }
  }
  // the unused tail of this workgroup's last spill block must read as "no entry"
  {
    const unsigned long long w = *sp_blk;
    const uint32_t next = (uint32_t)w, end = (uint32_t)(w >> 32);
    for (uint32_t i = next + fid; i < end && i < sl.cap; i += kFlushWaves * 64) sl.entries[(size_t)i * sl.stride] = kEmptyKey64;
  }
  // the heavy hitters' partial rows go to the spill list (merged by k_spill_merge)
  for (int i = fid; i < kHotSlots; i += kFlushWaves * 64) {
    const int64_t key = hot_key[i];
    if (key == kEmptyKey64) continue;
    int64_t part[kMaxInt];
    for (int m = 0; m < kMaxInt; ++m) part[m] = m < g.ns_int ? hot_slot[m * kHotSlots + i] : 0;
    spill_append(sl, key, part, g.ns_int);
  }
}

// ------------------------------------------------------------------------- phase 2
// LDS table of one unit: int64 keys[E] in buckets of four (32 B, two ds_read_b128 per probe),
// then one array per internal slot — COUNT slots are uint32 (a unit never sees 2^32 records
// in one chunk; an older partial that could overflow is routed to the spill list instead),
// every other op keeps the full 8 bytes.  Internal slots are ordered by op id, so a
// compile-time op set (MASK) fixes every array offset; MASK = 0 is the runtime-generic member.
typedef long long v2i64_t __attribute__((ext_vector_type(2)));
typedef long long v2i64_a8_t __attribute__((ext_vector_type(2), aligned(8)));  // (two consecutive 8-byte words, not 16-byte aligned)

MQ_D void lds_apply(int op, char* base, uint32_t e, int64_t vb) {
  switch (op) {
    case SO_COUNT:
    case SO_COUNT_NN: atomicAdd((uint32_t*)base + e, 1u); break;
    case SO_SUM_I: atomicAdd((unsigned long long*)base + e, (unsigned long long)vb); break;
    case SO_SUM_F: atomicAdd((double*)base + e, bits_dbl(vb)); break;
    case SO_MIN_I: atomicMin((long long*)base + e, (long long)vb); break;
    case SO_MAX_I: atomicMax((long long*)base + e, (long long)vb); break;
    case SO_MIN_F: a_minmax_f64<true, false, false>((int64_t*)base + e, bits_dbl(vb), 0.0); break;
    case SO_MAX_F: a_minmax_f64<true, true, false>((int64_t*)base + e, bits_dbl(vb), 0.0); break;
    default: break;
  }
}
// merge a PARTIAL (count / sum / min / max of several rows) into an LDS slot
MQ_D void lds_merge(int op, char* base, uint32_t e, int64_t partial) {
  switch (op) {
    case SO_COUNT:
    case SO_COUNT_NN: atomicAdd((uint32_t*)base + e, (uint32_t)partial); break;
    case SO_SUM_I: atomicAdd((unsigned long long*)base + e, (unsigned long long)partial); break;
    default: lds_apply(op, base, e, partial);
  }
}
MQ_D bool is_count_op(int op) { return op == SO_COUNT || op == SO_COUNT_NN; }
MQ_D int64_t lds_slot_value(int op, const char* base, uint32_t e) {
  return is_count_op(op) ? (int64_t)((const uint32_t*)base)[e] : ((const int64_t*)base)[e];
}
MQ_D void global_merge(int op, int64_t* gslot, int64_t partial) {
  switch (op) {
    case SO_COUNT:
    case SO_COUNT_NN:
    case SO_SUM_I: atomicAdd((unsigned long long*)gslot, (unsigned long long)partial); break;
    case SO_SUM_F: atomicAdd((double*)gslot, bits_dbl(partial)); break;
    case SO_MIN_I: atomicMin((long long*)gslot, (long long)partial); break;
    case SO_MAX_I: atomicMax((long long*)gslot, (long long)partial); break;
    case SO_MIN_F: a_minmax_f64<true, false, false>(gslot, bits_dbl(partial), 0.0); break;
    case SO_MAX_F: a_minmax_f64<true, true, false>(gslot, bits_dbl(partial), 0.0); break;
    default: break;
  }
}

constexpr uint32_t kNoEntry = 0xffffffffu;

// one row's update of entry e, every op of the set
template <int MASK>
MQ_D void apply_row(char* smem, const PartGeom& g, const PartSlots& ps, uint32_t e, int64_t vb) {
  const bool is_null = ps.val_nullable && vb == ps.null_bits;  // only COUNT(*) sees a NULL value
  if (MASK == 0) {
    for (int m = 0; m < g.ns_int; ++m)
      if (!is_null || ps.int_op[m] == SO_COUNT) lds_apply(ps.int_op[m], smem + g.slot_off[m], e, vb);
    return;
  }
  int m = 0;
#pragma unroll
  for (int op = SO_COUNT; op <= SO_COUNT_NN; ++op) {
    if (op == SO_KEY) continue;
    if (MASK & (1 << op)) {
      if (!is_null || op == SO_COUNT) lds_apply(op, smem + g.slot_off[m], e, vb);
      ++m;
    }
  }
}

// LDS table addressing: TWO candidate buckets per key (A from the home slot, B from other
// hash bits), a new key goes to the less loaded one.  With 4-slot buckets at ~64 % fill a
// single-choice table leaves 8 % of the groups outside their home bucket, and since a wave
// walks the probe rounds of its slowest lane, EVERY 256-record step paid 4-5 dependent LDS
// round trips; with two choices 99.9 % of the groups sit in A or B, which the hot loop reads
// up front.  Two lanes that insert the same new key at the same moment may pick different
// buckets; such twins are folded together before emission (fold_twins below).
struct Bucket {
  v2i64_t a, c;
};
// Keys live in two half-bucket arrays — slots {0,1} of every bucket, then slots {2,3} — so a
// 16-byte half-bucket read can start on any of the 16 bank groups of the LDS (a 32-byte
// bucket stride would only ever use 8 of them and double the conflicts of random probes).
MQ_D int64_t* key_slot(int64_t* lkeys, uint32_t n_buckets, uint32_t b, int j) {
  return lkeys + (j < 2 ? (size_t)b * 2 + j : (size_t)n_buckets * 2 + (size_t)b * 2 + (j - 2));
}
MQ_D int64_t* key_of_entry(int64_t* lkeys, uint32_t n_buckets, uint32_t e) {
  return key_slot(lkeys, n_buckets, e >> 2, (int)(e & 3));
}
MQ_D Bucket read_bucket(const int64_t* lkeys, uint32_t n_buckets, uint32_t b) {
  Bucket r;
  r.a = *((const v2i64_t*)lkeys + b);
  r.c = *((const v2i64_t*)lkeys + n_buckets + b);
  return r;
}
MQ_D int find_in(const Bucket& k, int64_t key) {
  return k.a.x == key ? 0 : k.a.y == key ? 1 : k.c.x == key ? 2 : k.c.y == key ? 3 : -1;
}
MQ_D int load_of(const Bucket& k) {
  return (k.a.x != kEmptyKey64) + (k.a.y != kEmptyKey64) + (k.c.x != kEmptyKey64) + (k.c.y != kEmptyKey64);
}
MQ_D int first_empty(const Bucket& k) {
  return k.a.x == kEmptyKey64 ? 0 : k.a.y == kEmptyKey64 ? 1 : k.c.x == kEmptyKey64 ? 2
         : k.c.y == kEmptyKey64 ? 3 : -1;
}
MQ_D uint32_t bucket_b_of(uint32_t h, uint32_t n_buckets) { return __umulhi(h * 2654435761u, n_buckets); }

// insert-or-find: candidate buckets ba / bb, then (both full) linear probing from bb + 1.
// Returns the entry or kNoEntry when the table is full.
MQ_D uint32_t lds_locate(int64_t* lkeys, uint32_t n_buckets, uint32_t ba, uint32_t bb, int64_t key) {
  for (;;) {
    const Bucket A = read_bucket(lkeys, n_buckets, ba), B = read_bucket(lkeys, n_buckets, bb);
    int j = find_in(A, key);
    if (j >= 0) return ba * 4 + (uint32_t)j;
    j = find_in(B, key);
    if (j >= 0) return bb * 4 + (uint32_t)j;
    const int la = load_of(A), lb = load_of(B);
    if (la == 4 && lb == 4) break;
    const bool use_a = la <= lb;
    const uint32_t bt = use_a ? ba : bb;
    const int fe = first_empty(use_a ? A : B);
    const int64_t old = (int64_t)atomicCAS((unsigned long long*)key_slot(lkeys, n_buckets, bt, fe),
                                           (unsigned long long)kEmptyKey64, (unsigned long long)key);
    if (old == kEmptyKey64 || old == key) return bt * 4 + (uint32_t)fe;
    // another key took that slot: look again
  }
  // both candidates full (they stay full): first fit over the following buckets
  uint32_t b = bb + 1 == n_buckets ? 0 : bb + 1;
  for (uint32_t trips = 0; trips < n_buckets;) {
    const Bucket K = read_bucket(lkeys, n_buckets, b);
    const int j = find_in(K, key);
    if (j >= 0) return b * 4 + (uint32_t)j;
    const int fe = first_empty(K);
    if (fe >= 0) {
      const int64_t old = (int64_t)atomicCAS((unsigned long long*)key_slot(lkeys, n_buckets, b, fe),
                                             (unsigned long long)kEmptyKey64, (unsigned long long)key);
      if (old == kEmptyKey64 || old == key) return b * 4 + (uint32_t)fe;
      continue;
    }
    b = b + 1 == n_buckets ? 0 : b + 1;
    ++trips;
  }
  return kNoEntry;
}

template <int MASK>
__global__ __launch_bounds__(kPartBlock) void k_part_aggregate(PartGeom g, const Rec* __restrict__ scratch,
                                                                const uint32_t* __restrict__ cnt,
                                                                PartSlots ps, TableArgs tab, SpillList sl,
                                                                int merge, uint32_t chunk_records_max,
                                                                unsigned long long* __restrict__ dbg,
                                                                unsigned int* __restrict__ pair_ctr, SliceMerge ms) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  int64_t* const lkeys = (int64_t*)smem_raw;
  uint32_t* bitmap = (uint32_t*)(smem_raw + g.lds_table_bytes);  // [(S2 + 31) / 32]
  const uint32_t bm_words = (g.hm.S2 + 31) / 32;
  uint32_t* lcnt = bitmap + bm_words;                            // [B]
  const int ns = g.ns_int;
  const int t = threadIdx.x;
  const int G = gridDim.x;
  const int R = (int)g.hm.R;
  const uint32_t n_buckets = g.E >> 2;
  // sub-ranges of one partition run on workgroups of the same XCD (block id mod 8) at the same
  // time, so the second reader of a run is served by that XCD's L2
  const bool paired = R > 1 && G % (8 * R) == 0;

  uint32_t lo = 0, n_slots = 0;  // this unit's home range [lo, lo + n_slots)
  // MI355Q_TRACE: lane 0 of every workgroup accumulates the cycles of each phase
  long long t_mark = dbg ? clock64() : 0;
  unsigned long long t_acc[5] = {0, 0, 0, 0, 0};
  auto mark = [&](int ph) {
    if (dbg) {
      const long long now = clock64();
      t_acc[ph] += (unsigned long long)(now - t_mark);
      t_mark = now;
    }
  };
  auto bucket_of = [&](uint32_t x) -> uint32_t { return __umulhi(x, g.b_mult); };
  // Four records per lane per step: all four bucket reads are issued before any is consumed
  // (LDS latency overlaps 4x), hits — the common case once a unit's table is warm — update
  // their slots straight away, misses fall back to the insert-or-find loop afterwards.
  auto insert4 = [&](const Rec& r0, const Rec& r1, const Rec& r2, const Rec& r3, uint32_t n_valid) {
    const Rec* rr[4] = {&r0, &r1, &r2, &r3};
    uint32_t ba[4], bb[4];
    uint32_t in_mask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t h = kid_hash(rr[j]->key);  // the records carry kids (kid_of), not keys
      const uint32_t x = home_from_hash(g.hm, h) - lo;
      const bool in = (uint32_t)j < n_valid && x < n_slots;  // else: past the run's end / another sub-range
      ba[j] = in ? bucket_of(x) : 0u;
      bb[j] = in ? bucket_b_of(h, n_buckets) : 0u;
      in_mask |= (in ? 1u : 0u) << j;
    }
    Bucket ka[4], kb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ka[j] = read_bucket(lkeys, n_buckets, ba[j]);
      kb[j] = read_bucket(lkeys, n_buckets, bb[j]);
    }
    uint32_t miss = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t key = rr[j]->key;
      const int ha = find_in(ka[j], key), hb = find_in(kb[j], key);
      if (in_mask & (1u << j)) {
        if (ha >= 0 || hb >= 0) {
          const uint32_t e = ha >= 0 ? ba[j] * 4 + (uint32_t)ha : bb[j] * 4 + (uint32_t)hb;
          apply_row<MASK>(smem_raw, g, ps, e, rr[j]->val);
        } else {
          miss |= 1u << j;
        }
      }
    }
    // misses (first touch of a group, or both candidate buckets full): one loop per lane
    while (miss) {
      const int j = __builtin_ctz(miss);
      miss &= miss - 1;
      const int64_t key = j == 0 ? r0.key : j == 1 ? r1.key : j == 2 ? r2.key : r3.key;
      const int64_t val = j == 0 ? r0.val : j == 1 ? r1.val : j == 2 ? r2.val : r3.val;
      const uint32_t a0 = j == 0 ? ba[0] : j == 1 ? ba[1] : j == 2 ? ba[2] : ba[3];
      const uint32_t b0 = j == 0 ? bb[0] : j == 1 ? bb[1] : j == 2 ? bb[2] : bb[3];
      const uint32_t e = lds_locate(lkeys, n_buckets, a0, b0, key);
      if (e != kNoEntry) apply_row<MASK>(smem_raw, g, ps, e, val);
      else spill_record(sl, ps, ns, key_of_kid(key), val);
    }
  };

  for (int it = 0;; ++it) {
    int p, r;
    if (paired) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
      r = j % R;
      p = (j / R) * 8 + xcd + (G / R) * it;
    } else {
      const int v = blockIdx.x + G * it;
      p = v / R;
      r = v % R;
    }
    if (p >= g.P) break;
    {
      const uint64_t p_lo = (uint64_t)p * g.hm.S1;
      const uint64_t p_hi = p_lo + g.hm.S1 < g.hm.d ? p_lo + g.hm.S1 : g.hm.d;
      uint64_t a = p_lo + (uint64_t)r * g.hm.S2, z = a + g.hm.S2;
      if (z > p_hi) z = p_hi;
      if (a > z) a = z;
      if (ms.n_src > 0) {  // slice merge: only the part of the unit inside this rank's home range
        if (a < ms.lo) a = ms.lo;
        if (z > ms.hi) z = ms.hi;
        if (a > z) a = z;
      }
      lo = (uint32_t)a;
      n_slots = (uint32_t)(z - a);
    }
    if (ms.n_src > 0 && n_slots == 0) continue;  // a unit of another rank's range (uniform per workgroup)
    for (uint32_t e = t; e < g.E; e += kPartBlock) {
      lkeys[e] = kEmptyKey64;  // (initialisation: the key layout does not matter here)
      for (int m = 0; m < ns; ++m) {
        if (is_count_op(ps.int_op[m])) ((uint32_t*)(smem_raw + g.slot_off[m]))[e] = 0;
        else ((int64_t*)(smem_raw + g.slot_off[m]))[e] = ps.int_init[m];
      }
    }
    for (uint32_t w = t; w < bm_words; w += kPartBlock) bitmap[w] = 0;
    for (int b = t; b < g.B; b += kPartBlock) lcnt[b] = cnt ? cnt[(size_t)p * g.B + b] : 0u;
    __syncthreads();
    mark(0);
    // Pair rendezvous (speed only): the two sub-range workgroups of a partition — same XCD, blocks b and
    // b ^ 8 — read the SAME runs in the same order; started together, the second reader of a line finds
    // it in the XCD's L2 instead of fetching it from HBM again.  One counter per pair, bumped by both at
    // this point of every unit; bounded wait, a partner that never shows up only costs the pacing.
    if (pair_ctr && paired && R == 2) {
      if (t == 0) {
        unsigned int* c = pair_ctr + (blockIdx.x & ~8u);
        atomicAdd(c, 1u);
        const unsigned int need = 2u * (unsigned int)(it + 1);
        unsigned int spins = 0;
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1u << 14)) break;
        }
      }
      __syncthreads();
    }
    if (n_slots) {
      // one partial row (a table row of an earlier chunk, or of a peer's slice) into the LDS table
      auto merge_row = [&](const int64_t* row, bool foreign_possible) {
        {
          const int64_t key = row[0];
          if (key == kEmptyKey64) return;
          int64_t part[kMaxInt];
          for (int m = 0; m < kMaxInt; ++m) part[m] = 0;
          bool big = false;  // a 32-bit LDS counter could wrap
          bool any_value = false;  // some nullable value slot already holds a value
          for (int j = tab.sp.n - 1; j >= 0; --j) {
            const int m = ps.out_map[j];
            if (m >= 0) {
              int64_t v = row[1 + j];
              if (tab.sp.null_init[j]) {
                if (v == tab.init[j]) v = ps.int_init[m];  // NULL so far (the slot's own sentinel): contributes nothing
                else any_value = true;
              }
              part[m] = v;
              if (is_count_op(ps.int_op[m]) &&
                  (uint64_t)part[m] + chunk_records_max >= 0xffffffffull) big = true;
            }
          }
          // the hidden non-NULL counter (no COUNT(col) / AVG in the output): only zero / non-zero
          // matters
          if (ps.nn_slot >= 0 && ps.nn_hidden) part[ps.nn_slot] = any_value ? 1 : 0;
          const uint32_t h = murmur3_u64((uint64_t)key);
          const uint32_t home = home_from_hash(g.hm, h);
          if (foreign_possible && (home < ms.lo || home >= ms.hi)) return;  // another rank's key
          const uint32_t x = home - lo;
          const int64_t kid = kid_of(key, h);
          const uint32_t e = (x < n_slots && !big && kid != kEmptyKey64)
                                 ? lds_locate(lkeys, n_buckets, bucket_of(x), bucket_b_of(h, n_buckets), kid)
                                 : kNoEntry;
          if (e == kNoEntry) {  // probed in from another range / no room / huge count: merged last
            spill_append(sl, key, part, ns);
            return;
          }
          for (int m = 0; m < ns; ++m) lds_merge(ps.int_op[m], smem_raw + g.slot_off[m], e, part[m]);
        }
      };
      if (merge) {
        // groups of earlier chunks: re-load this range of the table as partial rows
        for (uint32_t s = t; s < n_slots; s += kPartBlock) merge_row(tab.out + (size_t)(lo + s) * tab.row_quad, false);
      }
      for (int i = 0; i < ms.n_src; ++i) {
        for (uint32_t s = t; s < n_slots; s += kPartBlock)
          merge_row(ms.src[i] + (size_t)(lo + s - ms.lo) * tab.row_quad, true);
        if (lo + n_slots == ms.hi)  // the unit that ends the range also takes the pads
          for (uint32_t s = t; s < (uint32_t)ms.pad_rows; s += kPartBlock) merge_row(ms.pads[i] + (size_t)s * tab.row_quad, true);
      }
      mark(1);
      // one wave per run; four 16-byte record loads per lane, the next four already in flight
      const int wave = t >> 6, lane = t & 63;
      for (int b = wave; b < g.B; b += kPartBlock / 64) {
        const uint32_t n = lcnt[b];
        if (!n) continue;
        const Rec* run = scratch + ((size_t)p * g.B + b) * g.cap;
        const uint32_t last = n - 1;
        auto at = [&](uint32_t i) -> uint32_t { return i < last ? i : last; };  // clamped: always loadable
        Rec c0 = run[at(lane)], c1 = run[at(lane + 64)], c2 = run[at(lane + 128)], c3 = run[at(lane + 192)];
        for (uint32_t base = 0; base < n; base += 256) {
          const uint32_t i = base + lane, nx = i + 256;
          const Rec n0 = run[at(nx)], n1 = run[at(nx + 64)], n2 = run[at(nx + 128)], n3 = run[at(nx + 192)];
          insert4(c0, c1, c2, c3, i < n ? (n - i + 63) / 64 : 0u);
          c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
      }
    }
    __syncthreads();
    mark(2);
    // fold twins: an entry that is not in its key's bucket A may have a twin earlier in the
    // key's probe sequence (A, then B, then the buckets after B) — created when two lanes
    // inserted the same new key at the same moment.  The later copy is added to the earlier
    // one and cleared.
    for (uint32_t e = t; e < g.E && n_slots; e += kPartBlock) {
      const int64_t key = *key_of_entry(lkeys, n_buckets, e);  // a kid
      if (key == kEmptyKey64) continue;
      const uint32_t h = kid_hash(key);
      const uint32_t ba = bucket_of(home_from_hash(g.hm, h) - lo), bb = bucket_b_of(h, n_buckets);
      const uint32_t be = e >> 2;
      if (be == ba) continue;
      uint32_t twin = kNoEntry;
      int j = find_in(read_bucket(lkeys, n_buckets, ba), key);
      if (j >= 0) {
        twin = ba * 4 + (uint32_t)j;
      } else if (be != bb) {
        // overflow entry: B, then the buckets between B and its own
        for (uint32_t b = bb; b != be; b = b + 1 == n_buckets ? 0 : b + 1) {
          j = find_in(read_bucket(lkeys, n_buckets, b), key);
          if (j >= 0) {
            twin = b * 4 + (uint32_t)j;
            break;
          }
        }
      }
      if (twin == kNoEntry) continue;
      for (int m = 0; m < ns; ++m)
        lds_merge(ps.int_op[m], smem_raw + g.slot_off[m], twin, lds_slot_value(ps.int_op[m], smem_raw + g.slot_off[m], e));
      *key_of_entry(lkeys, n_buckets, e) = kEmptyKey64;
    }
    __syncthreads();
    // emit: claim the first free slot at or after the home slot (the reference's probing
    // rule, GroupByRuntime.cpp:25-48) in the LDS bitmap, then store the finished row
    for (uint32_t e = t; e < g.E && n_slots; e += kPartBlock) {
      const int64_t kid = *key_of_entry(lkeys, n_buckets, e);
      if (kid == kEmptyKey64) continue;
      const int64_t key = key_of_kid(kid);
      uint32_t s = home_from_hash(g.hm, kid_hash(kid)) - lo;
      bool placed = false;
      while (s < n_slots) {
        const uint32_t bit = 1u << (s & 31);
        const uint32_t old = atomicOr(&bitmap[s >> 5], bit);
        if (!(old & bit)) {
          placed = true;
          break;
        }
        // skip the occupied run inside this word
        const uint32_t free_above = ~(old | bit) & ~((bit << 1) - 1u);
        s = free_above ? (s & ~31u) + (uint32_t)__builtin_ctz(free_above) : (s | 31u) + 1u;
      }
      if (!placed) {  // probes past the end of the range: merged last, canonically
        int64_t part[kMaxInt];
        for (int m = 0; m < kMaxInt; ++m)
          part[m] = m < ns ? lds_slot_value(ps.int_op[m], smem_raw + g.slot_off[m], e) : 0;
        spill_append(sl, key, part, ns);
        continue;
      }
      int64_t* row = tab.out + (size_t)(lo + s) * tab.row_quad;
      row[0] = key;
      const bool no_value = ps.nn_slot >= 0 &&
                            lds_slot_value(SO_COUNT_NN, smem_raw + g.slot_off[ps.nn_slot], e) == 0;
      for (int j = 0; j < tab.sp.n; ++j) {
        const int m = ps.out_map[j];
        int64_t v = m >= 0 ? lds_slot_value(ps.int_op[m], smem_raw + g.slot_off[m], e)
                           : (tab.sp.op[j] == SO_KEY ? key : tab.init[j]);
        if (tab.sp.null_init[j] && no_value) v = tab.init[j];  // SUM / MIN / MAX of no value: the slot's NULL
        row[1 + j] = v;
      }
    }
    __syncthreads();
    mark(3);
    for (uint32_t s = t; s < n_slots; s += kPartBlock) {
      if (bitmap[s >> 5] & (1u << (s & 31))) continue;
      int64_t* row = tab.out + (size_t)(lo + s) * tab.row_quad;
      row[0] = kEmptyKey64;
      for (int j = 0; j < tab.sp.n; ++j) row[1 + j] = tab.init[j];
    }
    __syncthreads();
    mark(4);
  }
  if (dbg && t == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(dbg + i, t_acc[i]);
  }
}

// ------------------------------------------------------------------------- phase 3
// merge one partial row (internal slots) into the output table with the reference's insert-or-find
MQ_D void merge_partial_global(const PartSlots& ps, const TableArgs& tab, const SpillList& sl, int64_t key,
                               const int64_t* part) {
  int64_t* slots = baseline_find_or_insert(tab.out, tab.entry_count, tab.row_quad, 8, key);
  if (!slots) {
    atomicCAS(sl.d_err, 0, -1);  // out of group slots: the caller resizes and retries
    return;
  }
  const bool no_value = ps.nn_slot >= 0 && part[ps.nn_slot] == 0;
  for (int j = 0; j < tab.sp.n; ++j) {
    const int m = ps.out_map[j];
    if (m < 0) {
      if (tab.sp.op[j] == SO_KEY) MQ_STORE64(slots + j, key);
      continue;
    }
    if (!tab.sp.null_init[j]) {
      global_merge(tab.sp.op[j], slots + j, part[m]);
      continue;
    }
    if (no_value) continue;  // the partial holds no value: the slot keeps what it has (NULL or not)
    const int64_t v = part[m];
    switch (tab.sp.op[j]) {  // slot starts at the NULL sentinel: first value overwrites it
      case SO_SUM_I: a_sum_i64_skip<true>(slots + j, v, tab.init[j]); break;
      case SO_SUM_F: a_sum_f64_skip<true>(slots + j, bits_dbl(v), bits_dbl(tab.init[j])); break;
      case SO_MIN_I: a_min_i64_skip<true>(slots + j, v, tab.init[j]); break;
      case SO_MAX_I: a_max_i64_skip<true>(slots + j, v, tab.init[j]); break;
      case SO_MIN_F: a_minmax_f64<true, false, true>(slots + j, bits_dbl(v), bits_dbl(tab.init[j])); break;
      case SO_MAX_F: a_minmax_f64<true, true, true>(slots + j, bits_dbl(v), bits_dbl(tab.init[j])); break;
      default: break;
    }
  }
}

// partial (op) partial in LDS, 8-byte slots
MQ_D void lds_fold_partial(int op, int64_t* s, int64_t v) {
  switch (op) {
    case SO_COUNT:
    case SO_COUNT_NN:
    case SO_SUM_I: atomicAdd((unsigned long long*)s, (unsigned long long)v); break;
    case SO_SUM_F: atomicAdd((double*)s, bits_dbl(v)); break;
    case SO_MIN_I: atomicMin((long long*)s, (long long)v); break;
    case SO_MAX_I: atomicMax((long long*)s, (long long)v); break;
    case SO_MIN_F: a_minmax_f64<true, false, false>(s, bits_dbl(v), 0.0); break;
    case SO_MAX_F: a_minmax_f64<true, true, false>(s, bits_dbl(v), 0.0); break;
    default: break;
  }
}

// The spill list is dominated by repeats of a few keys when it is long (a moderately hot key
// that overflowed its runs): every workgroup first folds its 1024-entry slice in a small LDS
// table, so the output table sees one insert-or-find per distinct key and slice instead of one
// contended atomic chain per entry.
constexpr int kSpillSlice = 1024;
constexpr int kSpillLds = 2048;  // LDS table entries (50 % fill at worst)
__global__ __launch_bounds__(256) void k_spill_merge(PartSlots ps, TableArgs tab, SpillList sl, int ns) {
  __shared__ int64_t s_key[kSpillLds];
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];  // [ns][kSpillLds] partial slots
  int64_t* s_part = (int64_t*)smem_raw;
  uint32_t n = *sl.count;
  if (n > sl.cap) n = sl.cap;
  for (uint32_t base = blockIdx.x * kSpillSlice; base < n; base += gridDim.x * kSpillSlice) {
    for (int i = threadIdx.x; i < kSpillLds; i += 256) {
      s_key[i] = kEmptyKey64;
      for (int m = 0; m < ns; ++m) s_part[m * kSpillLds + i] = ps.int_init[m];
    }
    __syncthreads();
    const uint32_t end = base + kSpillSlice < n ? base + kSpillSlice : n;
    for (uint32_t i = base + threadIdx.x; i < end; i += 256) {
      const int64_t* e = sl.entries + (size_t)i * sl.stride;
      const int64_t ekey = e[0];
      if (ekey == kEmptyKey64) continue;  // padding of a partly used spill block
      uint32_t h = murmur3_u64((uint64_t)ekey) & (kSpillLds - 1);
      for (;;) {  // at most kSpillSlice distinct keys in kSpillLds slots: always terminates
        int64_t k = *(volatile int64_t*)&s_key[h];
        if (k == kEmptyKey64)
          k = (int64_t)atomicCAS((unsigned long long*)&s_key[h], (unsigned long long)kEmptyKey64,
                                 (unsigned long long)ekey);
        if (k == kEmptyKey64 || k == ekey) break;
        h = (h + 1) & (kSpillLds - 1);
      }
      for (int m = 0; m < ns; ++m) lds_fold_partial(ps.int_op[m], &s_part[m * kSpillLds + h], e[1 + m]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kSpillLds; i += 256) {
      const int64_t key = s_key[i];
      if (key == kEmptyKey64) continue;
      int64_t part[kMaxInt];
      for (int m = 0; m < kMaxInt; ++m) part[m] = m < ns ? s_part[m * kSpillLds + i] : 0;
      merge_partial_global(ps, tab, sl, key, part);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------- host side
int64_t op_identity(int op) {
  switch (op) {
    case SO_MIN_I: return INT64_MAX;
    case SO_MAX_I: return INT64_MIN;
    case SO_MIN_F: return dbl_bits(1.7976931348623157e308);
    case SO_MAX_F: return dbl_bits(-1.7976931348623157e308);
    default: return 0;
  }
}

// ------------------------------------------------------------------------- radix join probe
// Semi-join + aggregate (`SELECT SUM(fact.v), COUNT(*) FROM fact JOIN dim ON fact.k = dim.k`, the
// inner side reduced to the presence bitmap of its perfect table): a direct probe of a 12.5 MB
// bitmap is bound by ~7 x 10^10 random reads/s (cfg4).  Partitioning the fact rows by key range
// first makes every partition's bitmap slice fit LDS, so the probes never leave the CU:
//   phase 1  k_part_scatter<.., DIRECT>   16 B read + 16 B written per row
//   phase 2  k_part_join                  16 B read per row, LDS bit test, register accumulators
struct JoinPartArgs {
  int32_t P, B;
  uint32_t cap, S1;        // run capacity; keys per partition (multiple of 32)
  int64_t kmin;
  uint64_t range;          // max - min + 1
  const uint32_t* bitmap;  // 1 bit per key of [kmin, kmin + range)
  int64_t bm_words;
  int32_t n_slots;
  int32_t op[4];           // per output slot: 0 COUNT(*), 1 SUM(fact value)
  int64_t null_sum;        // NULL_BIGINT: skipped by the (non-grouped) SUM, and its slot's sentinel
};

// block-wide sums of the three accumulators, then one merge per output slot per workgroup
MQ_D void join_part_epilogue(const JoinPartArgs& a, long long sum, unsigned long long n_match,
                             unsigned long long n_nn, int64_t* __restrict__ out, long long* s_red) {
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
    n_match += __shfl_down(n_match, off, 64);
    n_nn += __shfl_down(n_nn, off, 64);
  }
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_red[wave * 3 + 0] = sum;
    s_red[wave * 3 + 1] = (long long)n_match;
    s_red[wave * 3 + 2] = (long long)n_nn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t_sum = 0, t_match = 0, t_nn = 0;
    for (int w = 0; w < n_waves; ++w) {
      t_sum += s_red[w * 3];
      t_match += s_red[w * 3 + 1];
      t_nn += s_red[w * 3 + 2];
    }
    for (int j = 0; j < a.n_slots; ++j) {
      if (a.op[j] == 0) {
        if (t_match) atomicAdd((unsigned long long*)(out + j), (unsigned long long)t_match);
      } else if (t_nn) {
        // the slot is NULL until the first non-NULL contribution (non-grouped SUM, skip_val)
        int64_t old = MQ_LOAD64(out + j);
        for (;;) {
          const int64_t nv = old == a.null_sum ? t_sum : old + t_sum;
          const int64_t seen = (int64_t)MQ_CAS64(out + j, old, nv);
          if (seen == old) break;
          old = seen;
        }
      }
    }
  }
}

__global__ __launch_bounds__(kPartBlock) void k_part_join(JoinPartArgs a, const Rec* __restrict__ scratch,
                                                           const uint32_t* __restrict__ cnt,
                                                           int64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  uint32_t* s_bm = (uint32_t*)smem_raw;                     // [S1 / 32]
  long long* s_red = (long long*)(smem_raw + (a.S1 >> 3));  // [16 waves x 3]
  const int t = threadIdx.x;
  const uint32_t nw = a.S1 >> 5;
  long long sum = 0;
  unsigned long long n_match = 0, n_nn = 0;
  for (int p = blockIdx.x; p < a.P; p += gridDim.x) {
    const int64_t w0 = ((int64_t)p * a.S1) >> 5;
    for (uint32_t i = t; i < nw; i += kPartBlock) s_bm[i] = w0 + i < a.bm_words ? a.bitmap[w0 + i] : 0u;
    __syncthreads();
    const int64_t base = a.kmin + (int64_t)p * a.S1;
    for (int b = 0; b < a.B; ++b) {
      const uint32_t n = cnt[(size_t)p * a.B + b];
      const Rec* run = scratch + ((size_t)p * a.B + b) * a.cap;
      for (uint32_t i0 = 0; i0 < n; i0 += 4 * kPartBlock) {
        Rec r[4];
        // four independent 16-byte loads in flight per lane: UNCONDITIONAL loads of a clamped index (a
        // load under `if (i < n)` sits in its own exec-masked block and the compiler waits for it
        // before issuing the next one), the out-of-range lanes are discarded afterwards
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t i = i0 + u * kPartBlock + t;
          r[u] = load_rec_nt(run + (i < n ? i : n - 1));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t x = (uint32_t)(r[u].key - base);
          const bool live = i0 + u * kPartBlock + t < n;  // (the loaded values are never patched: a write to
                                                          //  r[u] right after its load makes the compiler wait there)
          if (live && x < a.S1 && ((s_bm[x >> 5] >> (x & 31)) & 1u)) {
            ++n_match;
            if (r[u].val != a.null_sum) {
              sum += r[u].val;
              ++n_nn;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  join_part_epilogue(a, sum, n_match, n_nn, out, s_red);
}

// records that overflowed their runs and the partial rows of heavy-hitter keys: probe the global
// bitmap (entries: key, SUM partial, COUNT partial, COUNT_NN partial)
__global__ __launch_bounds__(256) void k_join_spill(JoinPartArgs a, SpillList sl, int64_t* __restrict__ out) {
  __shared__ long long s_red[4 * 3];
  uint32_t n = *sl.count;
  if (n > sl.cap) n = sl.cap;
  long long sum = 0;
  unsigned long long n_match = 0, n_nn = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int64_t* e = sl.entries + (size_t)i * sl.stride;
    const int64_t key = e[0];
    if (key == kEmptyKey64) continue;
    const uint64_t off = (uint64_t)key - (uint64_t)a.kmin;
    if (off < a.range && ((a.bitmap[off >> 5] >> (off & 31)) & 1u)) {
      sum += e[1];
      n_match += (unsigned long long)e[2];
      n_nn += (unsigned long long)e[3];
    }
  }
  join_part_epilogue(a, sum, n_match, n_nn, out, s_red);
}

// ------------------------------------------------------------------------- payload probe
// Joins whose targets READ the inner side — SUM / COUNT of an inner column, one-to-many tables
// (one joined row per match), LEFT joins — through the same radix route as the semi-join above.
// What makes it possible is an aggregate pushed into the join table: the non-grouped targets only
// need, per key k of a perfect-hash table,
//     cnt[k]   rows of the inner table with that key           (0 / 1 for a one-to-one table)
//     wsum[k]  the sum of the inner column over those rows, NULLs skipped
//     wnn[k]   how many of them are not NULL
// because  COUNT(*) = sum_rows cnt[key],  SUM(fact.v) = sum_rows v * cnt[key],
// SUM(dim.w) = sum_rows wsum[key],  COUNT(dim.w) = sum_rows wnn[key]  — exactly what one joined row
// per match (hash_join_idx / the one-to-many loop, JoinHashTableQueryRuntime.cpp:56-163,
// HashJoinRuntime.cpp:654-1110) adds up to, integer arithmetic, any order.  The arrays are built once
// per (join table, inner column) from the reference-layout table (k_join_payload) and cached on the
// join handle; phase 2 keeps a partition's slice of them in LDS (4-16 bytes per key), so a probe costs
// LDS reads instead of 2-5 dependent random HBM reads (offsets, counts, payload run, inner column).
// LEFT joins add the unmatched outer rows afterwards from totals over the whole outer table:
//     COUNT(*) = J + (N - M),  SUM(fact.v) = SVc + (SV_all - SVm)   (J joined rows, M matched outer rows).
// Key ranges too wide for LDS slices (cfg4: 100 M inner keys = 98 K keys per partition) keep the
// slice in the XCD's L2 instead: one 16-byte entry per key, and all workgroups of an XCD walk the
// SAME partition at the same time (k_part_probe_l2), so its 1.5 MB slice is fetched from HBM once and
// every probe after that is an L2 hit — 265 G random reads/s against 54 G/s from HBM
// (tools/microbench/gather.hip, profiles/r02_microbench_gather.txt).
struct Pay16 {
  int64_t wsum;
  uint32_t cnt, wnn;
};
struct ProbeArgs {
  int32_t P, B, R;
  uint32_t cap, S1, S2;    // run capacity; keys per partition; keys per sub-range (multiple of 32)
  int64_t kmin;
  uint64_t range;          // max - min + 1
  const uint32_t* cnt_k;   // [range]
  const int64_t* wsum_k;   // [range] or null
  const uint32_t* wnn_k;   // [range] or null (inner column NOT NULL: wnn = cnt)
  const Pay16* pay16;      // [range] the same three as one 16-byte entry per key (L2 mode), or null
  const int64_t* pay8;     // [range] L2 mode, one-to-one table without NULLs among the matching inner values:
                           // the inner value itself, INT64_MIN where no row has the key (divergent 16-byte
                           // loads run at half the rate of 8-byte ones: 133 vs 265 G probes/s)
  const int64_t* kkeys;    // keyed tables (MODE 2): the key of every table slot, EMPTY_KEY_64 where free;
                           // pay16 is then indexed by SLOT, `range` is the table's entry count, and pay8 is
                           // INTERLEAVED with the keys — {key, inner value or INT64_MIN}[entries], 16 B per
                           // slot: one gather finds the key and its payload (two arrays cost two cache lines
                           // per probe, and a 3 MB slice next to the record stream does not stay in a 4 MB L2)
  int64_t null_sum;        // NULL_BIGINT: skipped by the non-grouped SUM over the outer value
  uint32_t range_rcp;      // floor(2^32 / range) (keyed: slot = hash % range without a division)
};
MQ_D uint32_t probe_slot_of(const ProbeArgs& a, uint32_t h) {
  const uint32_t n = (uint32_t)a.range;
  uint32_t r = h - __umulhi(h, a.range_rcp) * n;
  if (r >= n) r -= n;
  return r;
}
// accumulators (device words, wrapping 64-bit adds)
enum ProbeAcc { PA_J = 0, PA_M, PA_SVC, PA_SVM, PA_NNVC, PA_NNVM, PA_SW, PA_NNW, PA_N };

MQ_D void probe_reduce_store(unsigned long long* acc, const unsigned long long* v, unsigned long long* s_red) {
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6, lane = threadIdx.x & 63;
  for (int k = 0; k < PA_N; ++k) {
    unsigned long long x = v[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) s_red[wave * PA_N + k] = x;
  }
  __syncthreads();
  if (threadIdx.x < PA_N) {
    unsigned long long tot = 0;
    for (int w = 0; w < n_waves; ++w) tot += s_red[w * PA_N + threadIdx.x];
    if (tot) atomicAdd(acc + threadIdx.x, tot);
  }
}

__global__ __launch_bounds__(kPartBlock) void k_part_probe(ProbeArgs a, const Rec* __restrict__ scratch,
                                                            const uint32_t* __restrict__ cnt,
                                                            unsigned long long* __restrict__ acc) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // LDS slice of one unit: wsum[S2] (8 B) | cnt[S2] | wnn[S2] | reduction scratch
  int64_t* s_ws = (int64_t*)smem_raw;
  uint32_t* s_cnt = (uint32_t*)(smem_raw + (a.wsum_k ? (size_t)a.S2 * 8 : 0));
  uint32_t* s_nn = s_cnt + a.S2;
  unsigned long long* s_red = (unsigned long long*)(s_nn + (a.wnn_k ? a.S2 : 0));
  const int t = threadIdx.x;
  unsigned long long v[PA_N];
  for (int k = 0; k < PA_N; ++k) v[k] = 0;
  const int units = a.P * a.R;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int p = u / a.R, r = u % a.R;
    const uint64_t off0 = (uint64_t)p * a.S1 + (uint64_t)r * a.S2;  // first key offset of the unit
    uint32_t nk = a.S2;                                             // keys of this unit
    if ((uint64_t)r * a.S2 + nk > a.S1) nk = a.S1 > (uint64_t)r * a.S2 ? (uint32_t)(a.S1 - (uint64_t)r * a.S2) : 0u;
    for (uint32_t i = t; i < a.S2; i += kPartBlock) {
      const bool live = i < nk && off0 + i < a.range;
      s_cnt[i] = live ? a.cnt_k[off0 + i] : 0u;
      if (a.wsum_k) s_ws[i] = live ? a.wsum_k[off0 + i] : 0;
      if (a.wnn_k) s_nn[i] = live ? a.wnn_k[off0 + i] : 0u;
    }
    __syncthreads();
    const int64_t base = a.kmin + (int64_t)off0;
    for (int b = 0; b < a.B; ++b) {
      const uint32_t n = cnt[(size_t)p * a.B + b];
      const Rec* run = scratch + ((size_t)p * a.B + b) * a.cap;
      for (uint32_t i0 = 0; i0 < n; i0 += 4 * kPartBlock) {
        Rec rec[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // four independent 16-byte loads in flight per lane (clamped, unconditional)
          const uint32_t i = i0 + q * kPartBlock + t;
          rec[q] = load_rec_nt(run + (i < n ? i : n - 1));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t x = (uint32_t)(rec[q].key - base);
          if (i0 + q * kPartBlock + t < n && x < nk) {
            const unsigned long long c = s_cnt[x];
            if (c) {
              const bool nn = rec[q].val != a.null_sum;
              v[PA_J] += c;
              v[PA_M] += 1;
              if (nn) {
                v[PA_SVC] += (unsigned long long)rec[q].val * c;
                v[PA_SVM] += (unsigned long long)rec[q].val;
                v[PA_NNVC] += c;
                v[PA_NNVM] += 1;
              }
              if (a.wsum_k) v[PA_SW] += (unsigned long long)s_ws[x];
              v[PA_NNW] += a.wnn_k ? (unsigned long long)s_nn[x] : c;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  probe_reduce_store(acc, v, s_red);
}

// L2 mode: workgroup (xcd, g) of the grid's 8 x G takes the runs g, g + G, ... of the partitions
// xcd, xcd + 8, ...  (block b runs on XCD b % 8 — observed, not promised: a different placement only
// costs the L2 hits, never correctness)
// Pacing (speed only): the workgroups of an XCD group stay within two consecutive partitions, so at
// most two slices (2 x 1.5 MB) compete for the XCD's 4 MB of L2 — without it the group spreads over
// many partitions and the probes fall back to Infinity-Cache speed (measured: 73 G probes/s instead of
// the 265 G/s of L2 hits).  `pace[xcd]` counts finished (workgroup, partition) pairs; a workgroup starts
// its iteration i once everybody has finished iteration i - 2.  Bounded wait: a group that is not
// fully resident only loses the pacing.
template <int BLOCK, bool PAY8, int UQ>
__global__ __launch_bounds__(BLOCK) void k_part_probe_l2(ProbeArgs a, const Rec* __restrict__ scratch,
                                                        const uint32_t* __restrict__ cnt,
                                                        unsigned long long* __restrict__ acc,
                                                        unsigned int* __restrict__ pace) {
  __shared__ unsigned long long s_red[(BLOCK / 64) * PA_N];
  const int xcd = blockIdx.x & 7, g = blockIdx.x >> 3, G = gridDim.x >> 3;
  const int t = threadIdx.x;
  unsigned long long v[PA_N];
  for (int k = 0; k < PA_N; ++k) v[k] = 0;
  bool pacing = pace != nullptr;
  int it = 0;
  for (int p = xcd; p < a.P; p += 8, ++it) {
    if (pacing && it >= 2) {
      if (t == 0) {
        const unsigned int need = (unsigned int)(it - 1) * (unsigned int)G;
        unsigned int spins = 0;
        while (__hip_atomic_load(pace + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > (1u << 14)) break;  // ~10 ms: the group is not running together
        }
      }
      __syncthreads();
    }
    for (int b = g; b < a.B; b += G) {
      const uint32_t n = cnt[(size_t)p * a.B + b];
      const Rec* run = scratch + ((size_t)p * a.B + b) * a.cap;
      if (!n) continue;
      // records of the NEXT step are already in flight while this step's gathers run; every load is an
      // unconditional load of a clamped index (a load under `if (i < n)` gets its own exec-masked block
      // and a wait right behind it)
      Rec rec[UQ];
#pragma unroll
      for (int q = 0; q < UQ; ++q) {
        const uint32_t i = q * BLOCK + t;
        rec[q] = load_rec_nt(run + (i < n ? i : n - 1));
      }
      for (uint32_t i0 = 0; i0 < n; i0 += UQ * BLOCK) {
        Rec nxt[UQ];
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
          const uint32_t i = i0 + UQ * BLOCK + q * BLOCK + t;
          nxt[q] = load_rec_nt(run + (i < n ? i : n - 1));
        }
        Pay16 pe[UQ];
        bool hit[UQ];
#pragma unroll
        for (int q = 0; q < UQ; ++q) {  // four independent gathers in flight (index 0 for keys outside the range)
          const uint64_t off = (uint64_t)rec[q].key - (uint64_t)a.kmin;
          hit[q] = i0 + q * BLOCK + t < n && off < a.range;
          if (PAY8) {
            const int64_t w = a.pay8[hit[q] ? off : 0ull];
            const uint32_t present = w != INT64_MIN;
            pe[q] = Pay16{present ? w : 0, present, present};
          } else {
            pe[q] = a.pay16[hit[q] ? off : 0ull];
          }
        }
#pragma unroll
        for (int q = 0; q < UQ; ++q) {
          const unsigned long long c = hit[q] ? pe[q].cnt : 0u;
          if (c) {
            const bool nn = rec[q].val != a.null_sum;
            v[PA_J] += c;
            v[PA_M] += 1;
            if (nn) {
              v[PA_SVC] += (unsigned long long)rec[q].val * c;
              v[PA_SVM] += (unsigned long long)rec[q].val;
              v[PA_NNVC] += c;
              v[PA_NNVM] += 1;
            }
            v[PA_SW] += (unsigned long long)pe[q].wsum;
            v[PA_NNW] += pe[q].wnn;
          }
        }
#pragma unroll
        for (int q = 0; q < UQ; ++q) rec[q] = nxt[q];
      }
    }
    if (pacing) {
      __syncthreads();
      if (t == 0) atomicAdd(pace + xcd, 1u);
    }
  }
  probe_reduce_store(acc, v, s_red);
}

// Keyed (baseline) join tables — sparse keys — through the same route: the outer rows are partitioned by
// the HOME SLOT of their key in the join table (scatter MODE 2), so a partition only ever probes one
// contiguous slice of the table: its keys (8 B per slot, `kkeys`) and the per-slot payload stay in the
// XCD's L2 while all workgroups of the XCD walk that partition.  Probing is the reference's linear probe
// (get_matching_slot, JoinHashTableQueryRuntime.cpp:40-54): stop at the key or at an empty slot.
// PM: 0 = keys (kkeys, 8 B per slot), then the slot's 16-byte payload entry; 1 = {key, inner value} interleaved (pay8);
//     2 = KEYS ONLY — a one-to-one table none of whose inner columns is read (cfg4 Query A on sparse keys): a match
//     counts one joined row and nothing else is fetched, so a partition's slice is 8 B per slot — 1.6 MB for cfg4's 200 M
//     slots, which an XCD's 4 MB L2 does keep next to the record stream (the 16-byte slices, 3.1 MB, hit 73 %:
//     profiles/r03_cfg4_join_pmc_before.txt)
// Probe scheduling (round 4).  Every lane keeps K probe sequences in flight and each round fetches, for every sequence, W
// consecutive slots; a sequence that ends (key found, or an empty slot) is REPLACED in the same round by the wave's next
// record, taken from a per-wave ring of records in LDS that the wave itself fills with coalesced chunks of its run —
// nothing is shared between waves, so the inner loop has no workgroup barrier, and a long probe sequence (the table is
// half full: 1.5 slots on average, but the longest of a few hundred sequences is ~10) delays only its own lane slot.
// Measured on cfg4's sparse variant, 3.33 B probes per launch (profiles/r04_cfg4_keyed_probe_variants_call*.jsonl, the
// SQ / TCP counters next to them):
//   one slot per dependent gather, every record's sequence walked on its own: a wave follows its slowest lane, ~10
//     dependent L2 round trips per wave-probe, 82 % of the wave cycles in s_waitcnt                             50.9 ms
//   the four sequences of a lane advancing together in windows of 32 bytes, still finishing together            ~41 ms
//   rings, K = 1 / 2 / 4 with 32-byte windows (2 - 4 gathers per round): ~40 ms whatever K — no longer latency, the
//     L2's request rate: every gather of a round is a request of its own, 2.7 per probe
//   four LANES per sequence, one aligned 64-byte line per round (1.0 request per probe, TCP_TCC_READ_REQ): 38.7 ms,
//     now bound by instruction issue (4.8 VALU + 3.5 SALU wave instructions per probe: sixteen probes per instruction)
//   rings, ONE 16-byte gather per sequence and round — two keys (PM 0 / 2), K = 4:                              27 ms
//   {key, value} slots (PM 1) gain nothing from narrower rounds (36 - 44 ms): their 3.1 MB slice does not stay in
//     the XCD's L2 next to the record stream, the probes run at the Infinity Cache's gather rate; K = 2, W = 2 is the
//     fastest of the measured members there.
constexpr int kProbeRing = 512, kProbeChunk = 256;  // records per wave: the ring (two chunks), one coalesced chunk
constexpr size_t kProbeKeyedLds = 16 * PA_N * sizeof(unsigned long long) + (size_t)16 * kProbeRing * sizeof(Rec);
#define MQ_WAVE_LDS_SYNC()               \
  do {                                   \
    asm volatile("" ::: "memory");       \
    __builtin_amdgcn_wave_barrier();     \
    asm volatile("" ::: "memory");       \
  } while (0)

template <int PM, int K, int W>
__global__ __launch_bounds__(1024) void k_part_probe_keyed(ProbeArgs a, const Rec* __restrict__ scratch,
                                                            const uint32_t* __restrict__ cnt,
                                                            unsigned long long* __restrict__ acc,
                                                            unsigned int* __restrict__ pace) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int CH = kProbeChunk, RING = kProbeRing;
  unsigned long long* s_red = (unsigned long long*)smem_raw;  // [16 * PA_N]
  const int xcd = blockIdx.x & 7, g = blockIdx.x >> 3, G = gridDim.x >> 3;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // (wave-uniform: the chunk bookkeeping stays in scalar registers)
  v4i32_t* ring = (v4i32_t*)(smem_raw + 16 * PA_N * sizeof(unsigned long long)) + wave * RING;
  const uint32_t entries = (uint32_t)a.range;
  // PM 1 / 2: a match is exactly one joined row, so J = M, SVC = SVM, NNVC = NNVM — three of the eight sums are kept
  // per lane: matches, SUM(outer value) and COUNT(outer value) over the matches, plus SUM(inner value) for PM 1
  unsigned long long v[PA_N];
  for (int k = 0; k < PA_N; ++k) v[k] = 0;
  int it = 0;
  for (int p = xcd; p < a.P; p += 8) {
   for (int r = 0; r < a.R; ++r, ++it) {
    // pacing as in k_part_probe_l2: the XCD's workgroups stay within two consecutive (partition, pass) units
    if (pace && it >= 2) {
      if (t == 0) {
        const unsigned int need = (unsigned int)(it - 1) * (unsigned int)G;
        unsigned int spins = 0;
        while (__hip_atomic_load(pace + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(16);
          if (++spins > (1u << 14)) break;
        }
      }
      __syncthreads();
    }
    // a.R > 1: the partition's slot range is walked in R passes (each re-reads the partition's records and keeps the keys
    // whose home slot is in the pass's sub-range)
    const uint32_t sub_lo = (uint32_t)p * a.S1 + (uint32_t)r * a.S2, sub_hi = sub_lo + a.S2;
    // the wave's share of the unit: chunk c of the runs b = g, g + G, ... with c = wave (mod 16)
    int b = g;
    uint32_t c = (uint32_t)wave;
    uint32_t n_run = b < a.B ? cnt[(size_t)p * a.B + b] : 0u;
    v4i32_t pre[4];          // the next chunk, on its way from HBM: records lane + 64 j
    uint32_t pre_count = 0;  // > 0: `pre` holds a chunk that is not in the ring yet
    bool more = true;
    uint32_t avail = 0, taken = 0;  // records put into / taken out of the wave's ring (wave-uniform, running totals)
    int64_t key[K], val[K];
    uint32_t cur[K], pslot[K];
    uint32_t active = 0, paying = 0, stale = 0;  // stale: rounds since one of the lane's sequences ended
#pragma unroll
    for (int q = 0; q < K; ++q) {
      key[q] = 0;
      val[q] = 0;
      cur[q] = 0;
      pslot[q] = 0;
    }
    for (;;) {
      if (pre_count && avail - taken <= (uint32_t)(RING - CH)) {  // room for a chunk: move it into the ring
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t i = (uint32_t)lane + 64u * j;
          if (i < pre_count) ring[(avail + i) & (RING - 1)] = pre[j];
        }
        avail += pre_count;
        pre_count = 0;
        MQ_WAVE_LDS_SYNC();
      }
      // free sequences take the next records of the ring, in lane order
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const bool want = !(active & (1u << q));
        const unsigned long long m = __ballot(want);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t room = avail - taken, wanted = (uint32_t)__popcll(m);
        const bool get = want && rank < room;
        const v4i32_t x = ring[(get ? taken + rank : taken) & (RING - 1)];
        if (get) {
          key[q] = (int64_t)(((uint64_t)(uint32_t)x.y << 32) | (uint64_t)(uint32_t)x.x);
          val[q] = (int64_t)(((uint64_t)(uint32_t)x.w << 32) | (uint64_t)(uint32_t)x.z);
          const uint32_t home = probe_slot_of(a, murmur1_u64((uint64_t)key[q]));
          if (a.R == 1 || (home >= sub_lo && home < sub_hi)) {  // (else: another pass's record)
            cur[q] = home;
            active |= 1u << q;
          }
        }
        taken += wanted < room ? wanted : room;
      }
      MQ_WAVE_LDS_SYNC();
      const bool busy = __any(active != 0);
      if (!busy && taken == avail && !pre_count && !more) break;
      // one window per sequence, all in flight together (unconditional loads — a free sequence re-reads its last window
      // out of the cache: loads under a lane-dependent `if` are serialised by the compiler, one s_waitcnt each;
      // r02_l2_probe_variants.jsonl)
      int64_t wk[K][W], wvv[K][W];
      Pay16 pe[K];
      if (busy) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
          if (PM != 1 && W == 2) {  // two keys with one 16-byte load (the key array carries a spare key behind its end)
            const v2i64_a8_t kk = *(const MQ_GLOBAL v2i64_a8_t*)(a.kkeys + cur[q]);
            wk[q][0] = kk.x;
            wk[q][1] = cur[q] + 1 < entries ? kk.y : a.kkeys[0];
            wvv[q][0] = wvv[q][1] = 0;
            if (PM == 0) pe[q] = a.pay16[pslot[q]];
            continue;
          }
#pragma unroll
          for (int j = 0; j < W; ++j) {
            uint32_t at = cur[q] + (uint32_t)j;
            if (at >= entries) at -= entries;  // (the probe sequence wraps at the table's end)
            if (PM == 1) {
              const v2i64_t kp = ((const MQ_GLOBAL v2i64_t*)a.pay8)[at];
              wk[q][j] = kp.x;
              wvv[q][j] = kp.y;
            } else {
              wk[q][j] = a.kkeys[at];
              wvv[q][j] = 0;
            }
          }
          if (PM == 0) pe[q] = a.pay16[pslot[q]];  // the payload entry of a sequence that found its slot last round
        }
      }
      if (!pre_count && more) {  // the next chunk (issued behind the windows: the round does not wait for it)
        while (b < a.B && (uint64_t)c * CH >= n_run) {
          b += G;
          c = (uint32_t)wave;
          n_run = b < a.B ? cnt[(size_t)p * a.B + b] : 0u;
        }
        if (b >= a.B) {
          more = false;
        } else {
          const uint32_t count = n_run - c * CH < (uint32_t)CH ? n_run - c * CH : (uint32_t)CH;
          const Rec* base = scratch + ((size_t)p * a.B + b) * a.cap + (size_t)c * CH;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t i = (uint32_t)lane + 64u * j;
            pre[j] = __builtin_nontemporal_load((const v4i32_t*)(base + (i < count ? i : count - 1)));
          }
          pre_count = count;
          c += 16;
        }
      }
      if (!busy) continue;
      bool ended = false;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const uint32_t bit = 1u << q;
        if (!(active & bit)) continue;
        if (PM == 0 && (paying & bit)) {
          const unsigned long long cc = pe[q].cnt;
          if (cc) {
            const bool nn = val[q] != a.null_sum;
            v[PA_J] += cc;
            v[PA_M] += 1;
            if (nn) {
              v[PA_SVC] += (unsigned long long)val[q] * cc;
              v[PA_SVM] += (unsigned long long)val[q];
              v[PA_NNVC] += cc;
              v[PA_NNVM] += 1;
            }
            v[PA_SW] += (unsigned long long)pe[q].wsum;
            v[PA_NNW] += pe[q].wnn;
          }
          paying &= ~bit;
          active &= ~bit;
          pslot[q] = 0;
          ended = true;
          continue;
        }
        int state = 0;  // 1 found, 2 empty slot: no match
        uint32_t at_hit = 0;
        int64_t w0 = 0;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if (state) continue;
          if (wk[q][j] == key[q]) {
            at_hit = cur[q] + (uint32_t)j;
            w0 = wvv[q][j];
            state = 1;
          } else if (wk[q][j] == kEmptyKey64) {
            state = 2;
          }
        }
        if (state == 1 && PM == 0) {
          if (at_hit >= entries) at_hit -= entries;
          pslot[q] = at_hit;
          paying |= bit;
        } else if (state) {
          if (state == 1 && (PM == 2 || w0 != INT64_MIN)) {
            v[PA_M] += 1;
            if (val[q] != a.null_sum) {
              v[PA_SVM] += (unsigned long long)val[q];
              v[PA_NNVM] += 1;
            }
            if (PM == 1) v[PA_SW] += (unsigned long long)w0;
          }
          active &= ~bit;
          ended = true;
        } else {
          cur[q] += W;
          if (cur[q] >= entries) cur[q] -= entries;
        }
      }
      // (a table without a free slot and a key that is not in it: no sequence of the lane has ended for a whole walk)
      stale = ended ? 0u : stale + 1u;
      if (stale > entries / W + 1) {
        active = 0;
        paying = 0;
        stale = 0;
      }
    }
    if (pace) {
      __syncthreads();
      if (t == 0) atomicAdd(pace + xcd, 1u);
    }
   }
  }
  if (PM != 0) {  // one joined row per match
    v[PA_J] = v[PA_M];
    v[PA_SVC] = v[PA_SVM];
    v[PA_NNVC] = v[PA_NNVM];
    if (PM == 1) v[PA_NNW] = v[PA_M];
  }
  __syncthreads();
  probe_reduce_store(acc, v, s_red);
}

// keys + per-slot payload of a keyed join table with ONE 8-byte key component: one-to-one
// `{key, row id}[entries]`, one-to-many `keys[entries] | offsets | counts | payloads` (include/mi355q.h)
__global__ __launch_bounds__(256) void k_join_payload_keyed(const int64_t* __restrict__ table, int hash_type,
                                                            int64_t entries, const int64_t* __restrict__ w,
                                                            int64_t* __restrict__ kkeys, Pay16* __restrict__ pay16,
                                                            int64_t* __restrict__ pay8, int32_t* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool any_null = false;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < entries; x += stride) {
    uint32_t c = 0, nn = 0;
    unsigned long long sum = 0;
    int64_t key;
    if (hash_type == 1) {
      key = table[2 * x];
      const int64_t id = table[2 * x + 1];
      if (key != kEmptyKey64 && id >= 0) {
        c = 1;
        if (w) {
          const int64_t val = w[id];
          if (val != INT64_MIN) {
            sum = (unsigned long long)val;
            nn = 1;
          }
        }
      }
    } else {
      key = table[x];
      const int32_t* offsets = (const int32_t*)(table + entries);
      const int32_t off = offsets[x];
      const int32_t n = offsets[entries + x];
      if (key != kEmptyKey64 && off >= 0 && n > 0) {
        c = (uint32_t)n;
        if (w) {
          const int32_t* ids = offsets + 2 * entries + off;
          for (int32_t i = 0; i < n; ++i) {
            const int64_t val = w[ids[i]];
            if (val != INT64_MIN) {
              sum += (unsigned long long)val;
              ++nn;
            }
          }
        }
      }
    }
    if (!w) nn = c;
    kkeys[x] = key;
    if (pay8) {  // interleaved {key, value}: see ProbeArgs::kkeys
      pay8[2 * x] = key;
      pay8[2 * x + 1] = c ? (int64_t)sum : INT64_MIN;
    }
    pay16[x] = Pay16{(int64_t)sum, c, nn};
    any_null |= nn != c;
  }
  if (__any(any_null) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}

// run overflows and heavy-hitter partial rows {key, SUM(v) partial, COUNT partial, COUNT_NN partial}
// against the global arrays
__global__ __launch_bounds__(256) void k_probe_spill(ProbeArgs a, SpillList sl, unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long s_red[4 * PA_N];
  uint32_t n = *sl.count;
  if (n > sl.cap) n = sl.cap;
  unsigned long long v[PA_N];
  for (int k = 0; k < PA_N; ++k) v[k] = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int64_t* e = sl.entries + (size_t)i * sl.stride;
    const int64_t key = e[0];
    if (key == kEmptyKey64) continue;
    uint64_t off;
    if (a.kkeys) {  // keyed table: linear probe for the slot
      const uint32_t entries = (uint32_t)a.range;
      uint32_t h = probe_slot_of(a, murmur1_u64((uint64_t)key));
      bool found = false;
      for (uint32_t trips = 0; trips < entries; ++trips) {
        const int64_t k = a.kkeys[h];
        if (k == key) {
          found = true;
          break;
        }
        if (k == kEmptyKey64) break;
        h = h + 1 == entries ? 0 : h + 1;
      }
      if (!found) continue;
      off = h;
    } else {
      off = (uint64_t)key - (uint64_t)a.kmin;
      if (off >= a.range) continue;
    }
    unsigned long long c, ws, wn;
    if (a.pay8) {
      const int64_t w = a.kkeys ? a.pay8[2 * off + 1] : a.pay8[off];
      c = w != INT64_MIN;
      ws = c ? (unsigned long long)w : 0ull;
      wn = c;
    } else if (a.pay16) {
      const Pay16 pe = a.pay16[off];
      c = pe.cnt;
      ws = (unsigned long long)pe.wsum;
      wn = pe.wnn;
    } else {
      c = a.cnt_k[off];
      ws = a.wsum_k ? (unsigned long long)a.wsum_k[off] : 0ull;
      wn = a.wnn_k ? (unsigned long long)a.wnn_k[off] : c;
    }
    if (!c) continue;
    const unsigned long long rows = (unsigned long long)e[2], rows_nn = (unsigned long long)e[3];
    v[PA_J] += rows * c;
    v[PA_M] += rows;
    v[PA_SVC] += (unsigned long long)e[1] * c;
    v[PA_SVM] += (unsigned long long)e[1];
    v[PA_NNVC] += rows_nn * c;
    v[PA_NNVM] += rows_nn;
    v[PA_SW] += rows * ws;
    v[PA_NNW] += rows * wn;
  }
  probe_reduce_store(acc, v, s_red);
}

// totals over the whole outer value column (LEFT joins): acc2 = {sum of non-NULL values, their count}
template <typename VT>
__global__ __launch_bounds__(256) void k_outer_totals(const int8_t* const* __restrict__ cols,
                                                      const int64_t* __restrict__ num_rows, int n_frags, int n_cols,
                                                      int vcol, int64_t null_sum, unsigned long long* __restrict__ acc2) {
  __shared__ unsigned long long s_red[4 * 2];
  unsigned long long sum = 0, nn = 0;
  scan_fragments<none_t, none_t, VT>(cols, num_rows, n_frags, n_cols, 0, 0, vcol, [&](none_t, none_t, VT val) {
    const int64_t x = (int64_t)val;
    if (x != null_sum) {
      sum += (unsigned long long)x;
      ++nn;
    }
  });
  for (int off = 32; off > 0; off >>= 1) {
    sum += __shfl_down(sum, off, 64);
    nn += __shfl_down(nn, off, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_red[wave * 2] = sum;
    s_red[wave * 2 + 1] = nn;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned long long tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_red[w * 2 + threadIdx.x];
    if (tot) atomicAdd(acc2 + threadIdx.x, tot);
  }
}

// the output slots from the accumulators (one thread).  op: 0 COUNT(*), 1 SUM(outer v), 2 SUM(inner w),
// 3 COUNT(outer v), 4 COUNT(inner w).  A non-grouped SUM that saw no value stays NULL (skip_val
// aggregates start at the sentinel, OutputBufferInitialization.cpp:79-81).
struct ProbeFinish {
  int32_t n_slots, left;
  int32_t op[4];
  int64_t n_outer_rows;    // N
  int64_t null_sum;
};
__global__ void k_probe_finish(ProbeFinish f, const unsigned long long* __restrict__ acc,
                               const unsigned long long* __restrict__ acc2, int64_t* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  const unsigned long long J = acc[PA_J], M = acc[PA_M];
  unsigned long long sv = acc[PA_SVC], nnv = acc[PA_NNVC];
  unsigned long long rows = J;
  if (f.left) {  // every unmatched outer row appears once, inner columns NULL
    rows = J + ((unsigned long long)f.n_outer_rows - M);
    sv = acc[PA_SVC] + (acc2[0] - acc[PA_SVM]);
    nnv = acc[PA_NNVC] + (acc2[1] - acc[PA_NNVM]);
  }
  for (int j = 0; j < f.n_slots; ++j) {
    switch (f.op[j]) {
      case 0: out[j] = (int64_t)rows; break;
      case 1: out[j] = nnv ? (int64_t)sv : f.null_sum; break;
      case 2: out[j] = acc[PA_NNW] ? (int64_t)acc[PA_SW] : f.null_sum; break;
      case 3: out[j] = (int64_t)nnv; break;
      default: out[j] = (int64_t)acc[PA_NNW]; break;
    }
  }
}

// cnt / wsum / wnn of every key slot of a perfect-hash join table (layouts: include/mi355q.h):
// one-to-one `int32 slot[entries]` (-1 empty), one-to-many `offsets | counts | payloads`.
// flags[0] is set when some matching inner value is NULL (then wnn differs from cnt and has to be
// carried along).  w == nullptr: only the counts are wanted.
__global__ __launch_bounds__(256) void k_join_payload(const int32_t* __restrict__ table, int hash_type,
                                                      int64_t entries, const int64_t* __restrict__ w,
                                                      uint32_t* __restrict__ cnt_k, int64_t* __restrict__ wsum_k,
                                                      uint32_t* __restrict__ wnn_k, Pay16* __restrict__ pay16,
                                                      int64_t* __restrict__ pay8, int32_t* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool any_null = false;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < entries; x += stride) {
    uint32_t c = 0, nn = 0;
    unsigned long long sum = 0;
    if (hash_type == 0) {
      const int32_t id = table[x];
      if (id >= 0) {
        c = 1;
        if (w) {
          const int64_t v = w[id];
          if (v != INT64_MIN) {
            sum = (unsigned long long)v;
            nn = 1;
          }
        }
      }
    } else {
      const int32_t off = table[x];
      const int32_t n = table[entries + x];
      if (off >= 0 && n > 0) {
        c = (uint32_t)n;
        if (w) {
          const int32_t* ids = table + 2 * entries + off;
          for (int32_t i = 0; i < n; ++i) {
            const int64_t v = w[ids[i]];
            if (v != INT64_MIN) {
              sum += (unsigned long long)v;
              ++nn;
            }
          }
        }
      }
    }
    if (!w) nn = c;  // no inner column: nothing can be NULL
    if (pay8) pay8[x] = c ? (int64_t)sum : INT64_MIN;  // (only read when the table is one-to-one and nn == c everywhere)
    if (pay16) {
      pay16[x] = Pay16{(int64_t)sum, c, nn};
    } else if (!pay8) {
      cnt_k[x] = c;
      if (w) {
        wsum_k[x] = (int64_t)sum;
        wnn_k[x] = nn;
      }
    }
    any_null |= nn != c;
  }
  if (__any(any_null) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}

struct PartPlanHost {
  PartGeom g;
  PartSlots ps;
  int64_t chunk_rows;     // max rows per chunk
  int64_t rec_bytes;      // runs
  int64_t cnt_bytes;      // run lengths
  int64_t scratch_bytes;  // runs + lengths + spill list
  size_t lds1, lds2;
  uint32_t spill_cap;     // spill list entries
  int n_cand;             // heavy-hitter candidate slots in phase 1
  int op_mask;            // set of internal ops (bit per SlotOp)
  // phase 1 of chunk i + 1 next to phase 2 of chunk i: two buffers of `buf_bytes` (runs + lengths +
  // spill list) in the scratch, g.B scatter workgroups and grid2 aggregate workgroups share the device
  int overlap;            // 0 = one phase after the other on the whole device
  int grid2;              // workgroups of phase 2
  int64_t buf_bytes;      // one buffer set (256-byte multiple)
};

constexpr size_t kLdsTableBudget = 150 * 1024;
// phase 1 of the next chunk on this many CUs while phase 2 runs on the rest (0 = off); an input is cut into at
// least kOverlapMinChunks chunks then, so that all but the first scatter and the last aggregate have company
constexpr int kDefaultOverlapCus = 0;
constexpr int kOverlapMinChunks = 6;
constexpr int64_t kOverlapMinRows = 256ll << 20;

bool make_part_plan(const DevPlan& p, const FastShape& fs, const FragView& fv, int n_cus,
                    int64_t scratch_cap, PartPlanHost* out) {
  PartPlanHost& h = *out;
  if (p.entry_count < 64 || p.entry_count >= ((int64_t)1 << 32)) return false;
  if (fv.max_frag_rows > 0xfff00000ll) return false;  // 32-bit LDS counters per chunk
  // internal slots: distinct ops only (COUNT(*) and AVG's count share one LDS counter)
  int n_int = 0;
  for (int j = 0; j < MI355Q_MAX_SLOTS; ++j) h.ps.out_map[j] = -1;
  for (int j = 0; j < kMaxInt; ++j) {
    h.ps.int_op[j] = -1;
    h.ps.int_init[j] = 0;
  }
  for (int j = 0; j < fs.sp.n; ++j) {
    const int op = fs.sp.op[j];
    if (op == SO_KEY) continue;
    int m = -1;
    for (int k = 0; k < n_int; ++k)
      if (h.ps.int_op[k] == op) m = k;
    if (m < 0) {
      if (n_int >= kMaxInt) return false;
      m = n_int++;
      h.ps.int_op[m] = op;
      h.ps.int_init[m] = op_identity(op);
    }
    h.ps.out_map[j] = m;
  }
  if (n_int == 0) return false;
  // nullable value column with a SUM / MIN / MAX: a hidden counter of non-NULL values tells
  // emission whether the slot holds a value or is still NULL (COUNT(col) / AVG already keep one)
  h.ps.val_nullable = fs.sp.val_nullable;
  h.ps.null_bits = fs.sp.null_bits;
  h.ps.nn_slot = -1;
  h.ps.nn_hidden = 0;
  h.ps.pad_ = 0;
  bool any_null_init = false;
  for (int j = 0; j < fs.sp.n; ++j) any_null_init |= fs.sp.null_init[j] != 0;
  if (fs.sp.val_nullable && any_null_init) {
    bool have = false;
    for (int k = 0; k < n_int; ++k) have |= h.ps.int_op[k] == SO_COUNT_NN;
    if (!have) {
      if (n_int >= kMaxInt) return false;
      h.ps.int_op[n_int] = SO_COUNT_NN;
      h.ps.int_init[n_int] = 0;
      ++n_int;
      h.ps.nn_hidden = 1;
    }
  }
  // internal slots in ascending op order (the compile-time op sets of phase 2 rely on it)
  for (int a = 0; a < n_int; ++a)
    for (int b = a + 1; b < n_int; ++b)
      if (h.ps.int_op[b] < h.ps.int_op[a]) {
        const int oa = h.ps.int_op[a], ob = h.ps.int_op[b];
        h.ps.int_op[a] = ob;
        h.ps.int_op[b] = oa;
        h.ps.int_init[a] = op_identity(ob);
        h.ps.int_init[b] = op_identity(oa);
        for (int j = 0; j < MI355Q_MAX_SLOTS; ++j) {
          if (h.ps.out_map[j] == a) h.ps.out_map[j] = b;
          else if (h.ps.out_map[j] == b) h.ps.out_map[j] = a;
        }
      }
  h.op_mask = 0;
  for (int m = 0; m < n_int; ++m) {
    h.op_mask |= 1 << h.ps.int_op[m];
    if (h.ps.int_op[m] == SO_COUNT_NN && any_null_init) h.ps.nn_slot = m;
  }
  h.g.ns_int = n_int;
  size_t entry_bytes = 8;
  for (int m = 0; m < n_int; ++m)
    entry_bytes += (h.ps.int_op[m] == SO_COUNT || h.ps.int_op[m] == SO_COUNT_NN) ? 4 : 8;
  const uint32_t e_max = (uint32_t)(kLdsTableBudget / entry_bytes) & ~3u;
  // expected groups: the caller sizes the table at ~2 x NDV (50 % fill, docs results.rst)
  const uint64_t d = (uint64_t)p.entry_count;
  const uint64_t groups = d / 2 > 0 ? d / 2 : 1;
  const uint64_t per_unit = (uint64_t)(0.8 * e_max);
  const uint64_t units = (groups + per_unit - 1) / per_unit;
  uint32_t P = 16;  // a line's segments (<= 64) must fit one flusher wave pass
  while (P < 1024 && P < units) P <<= 1;
  // phase 2 runs one workgroup per (partition, sub-range): with few groups the LDS table would
  // allow a handful of partitions, which leaves most CUs idle while the records are read back
  while (P < 1024 && P < (uint32_t)n_cus && (uint64_t)P * 4 <= d) P <<= 1;
  const uint32_t R = (uint32_t)((units + P - 1) / P);
  if (R > (uint32_t)kMaxSub) return false;
  h.g.P = (int32_t)P;
  h.g.L = kStageRecs / P;
  h.g.lgL = 0;
  while ((1u << h.g.lgL) < h.g.L) ++h.g.lgL;
  HomeMap& hm = h.g.hm;
  hm.d = (uint32_t)d;
  hm.S1 = (uint32_t)((d + P - 1) / P);
  hm.R = R < 1 ? 1 : R;
  hm.S2 = (hm.S1 + hm.R - 1) / hm.R;
  hm.d_rcp = (uint32_t)(((uint64_t)1 << 32) / hm.d);
  hm.s1_rcp = hm.S1 >= 2 ? (uint32_t)(((uint64_t)1 << 32) / hm.S1) : 0u;
  if (hm.S1 < 2) return false;
  uint64_t e_want = hm.S2;  // an LDS entry per home slot when it fits (50 % fill)
  if (e_want > e_max) e_want = e_max;
  if (e_want < 64) e_want = 64;
  h.g.E = ((uint32_t)e_want + 3) & ~3u;  // keeps every slot array 16-byte aligned
  h.g.lds_table_bytes = (uint32_t)((size_t)h.g.E * entry_bytes);
  {
    uint32_t off = h.g.E * 8;  // 8-byte slot arrays first, then the 4-byte counters
    for (int m = 0; m < kMaxInt; ++m) h.g.slot_off[m] = 0;
    auto is_cnt = [&](int m) { return h.ps.int_op[m] == SO_COUNT || h.ps.int_op[m] == SO_COUNT_NN; };
    for (int m = 0; m < n_int; ++m)
      if (!is_cnt(m)) { h.g.slot_off[m] = off; off += h.g.E * 8; }
    for (int m = 0; m < n_int; ++m)
      if (is_cnt(m)) { h.g.slot_off[m] = off; off += h.g.E * 4; }
  }
  {
    const uint64_t m = ((uint64_t)(h.g.E / 4) << 32) / hm.S2;
    h.g.b_mult = (uint32_t)(m > 0xffffffffull ? 0xffffffffull : m);
  }
  h.g.B = n_cus;  // one 1024-lane workgroup per CU
  h.grid2 = n_cus;
  h.overlap = 0;
  {
    int ov = tune_knobs().overlap_cus;
    if (ov == 0) ov = kDefaultOverlapCus;
    // (the 16-workgroup granularity keeps phase 2's sub-range pairs on one XCD; small devices — the host
    // simulation's 8 "CUs" — take any split)
    if (ov > 0 && ov < n_cus && fv.total_rows >= (tune_knobs().overlap_cus > 0 ? 2 * fv.max_frag_rows : kOverlapMinRows) &&
        fv.n_frags >= 2) {
      h.overlap = 1;
      h.g.B = ov;
      h.grid2 = n_cus - ov;
      scratch_cap /= 2;
    }
  }
  // chunking: worst case every row survives the filter; shrink the chunk until the runs
  // (1.2 x mean + 6 sigma + a line of slack per run) fit the scratch cap, never below one
  // fragment
  int64_t chunk_rows = fv.total_rows > 0 ? fv.total_rows : 1;
  if (h.overlap) {
    const int64_t want = (fv.total_rows + kOverlapMinChunks - 1) / kOverlapMinChunks + fv.max_frag_rows;
    if (want < chunk_rows) chunk_rows = want;
    if (chunk_rows < fv.max_frag_rows) chunk_rows = fv.max_frag_rows;
  }
  if (chunk_rows > 0xfff00000ll) chunk_rows = 0xfff00000ll;  // 32-bit LDS counters per chunk
  for (;;) {
    const double per_run = (double)chunk_rows / ((double)P * h.g.B);
    uint64_t cap = (uint64_t)(per_run * 1.2 + 6.0 * __builtin_sqrt(per_run + 1.0)) + h.g.L;
    cap = (cap + h.g.L - 1) / h.g.L * h.g.L;  // whole lines
    if (cap > 0x7fffffffull) return false;
    if ((uint64_t)P * h.g.B * cap >= ((uint64_t)1 << 32)) {  // 32-bit record indices in phase 1
      if (chunk_rows <= fv.max_frag_rows) return false;
      chunk_rows = (int64_t)(chunk_rows * 0.97);
      if (chunk_rows < fv.max_frag_rows) chunk_rows = fv.max_frag_rows;
      continue;
    }
    h.g.cap = (uint32_t)cap;
    h.rec_bytes = (int64_t)P * h.g.B * (int64_t)cap * (int64_t)sizeof(Rec);
    h.cnt_bytes = (((int64_t)P * h.g.B * 4 + 255) & ~255ll) + kPairCtrBytes;  // run lengths + the pair counters of phase 2
    // spill list: room for 1/16 of the chunk's rows (skewed keys overflow their runs by a few
    // per cent of the records), at least kSpillMin entries
    int64_t spill_cap = chunk_rows / 16;
    if (spill_cap < (int64_t)kSpillMin) spill_cap = kSpillMin;
    if (spill_cap > 0x7fffffffll) spill_cap = 0x7fffffffll;
    h.spill_cap = (uint32_t)spill_cap;
    const int64_t spill_bytes = 256 + spill_cap * 8 * (int64_t)(1 + n_int);
    h.buf_bytes = (h.rec_bytes + h.cnt_bytes + spill_bytes + 255) & ~255ll;
    h.scratch_bytes = h.overlap ? 2 * h.buf_bytes : h.rec_bytes + h.cnt_bytes + spill_bytes;
    if (h.buf_bytes <= scratch_cap || chunk_rows <= fv.max_frag_rows) break;
    chunk_rows = (int64_t)(chunk_rows * 0.97);
    if (chunk_rows < fv.max_frag_rows) chunk_rows = fv.max_frag_rows;
  }
  // equal-sized chunks: 10 B rows under a 3.4 B-row limit are three chunks of 3.3 B, not two full
  // ones and a sliver (every chunk pays the table re-load, the emission and the launch tails)
  if (fv.total_rows > chunk_rows) {
    const int64_t n_chunks = (fv.total_rows + chunk_rows - 1) / chunk_rows;
    const int64_t even = (fv.total_rows + n_chunks - 1) / n_chunks + fv.max_frag_rows;  // fragments are not split
    if (even < chunk_rows) chunk_rows = even;
  }
  h.chunk_rows = chunk_rows;
  // staging lines + cursor / written / flushed + done + heavy-hitter table
  {
    const size_t fixed = kStageRecs * sizeof(Rec) + (size_t)P * 12 + 48 + (size_t)kHotSlots * (8 + 8 * (size_t)n_int);
    h.n_cand = 2048;
    while (h.n_cand > 64 && fixed + (size_t)h.n_cand * 4 > 160 * 1024) h.n_cand >>= 1;
    h.lds1 = fixed + (size_t)h.n_cand * 4;
    if (h.lds1 > 160 * 1024) return false;
  }
  h.lds2 = (size_t)h.g.E * entry_bytes + (size_t)((hm.S2 + 31) / 32) * 4 + (size_t)h.g.B * 4;
  return h.lds2 <= 160 * 1024;
}

// second stream and the four ordering events of the overlapped pipeline, one set per device, created on first use and
// kept (like the per-device workspace)
struct OverlapRes {
  hipStream_t s2 = nullptr;
  hipEvent_t ev_scat[2] = {nullptr, nullptr}, ev_agg[2] = {nullptr, nullptr};
};
OverlapRes* overlap_resources() {
  static OverlapRes res[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  OverlapRes& r = res[dev < 0 ? 0 : dev % 64];
  if (!r.s2) {
    if (hipStreamCreateWithFlags(&r.s2, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (int i = 0; i < 2; ++i)
      if (hipEventCreateWithFlags(&r.ev_scat[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&r.ev_agg[i], hipEventDisableTiming) != hipSuccess)
        return nullptr;
  }
  return &r;
}

template <typename FT, typename VT>
hipError_t launch_scatter_t(int grid, size_t lds, hipStream_t s, const FragView& fv, int f0, int nf,
                            const RangeFilter& flt, int kcol, int vcol, const ScatterArgs& g,
                            Rec* scratch, uint32_t* cnt, const SpillList& sl) {
  const int8_t* const* cols = fv.d_cols + (size_t)f0 * fv.n_cols;
  const int64_t* rows = fv.d_num_rows + f0;
  // opt in to > 64 KB of dynamic LDS (gfx950: 160 KB per workgroup)
  (void)hipFuncSetAttribute((const void*)k_part_scatter<FT, VT>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_part_scatter<FT, VT>), dim3(grid), dim3(kPartBlock), lds, s, cols, rows, nf,
                     fv.n_cols, flt, kcol, vcol, g, scratch, cnt, sl);
  return hipGetLastError();
}

template <typename FT>
hipError_t launch_scatter_v(const FastShape& fs, int grid, size_t lds, hipStream_t s,
                            const FragView& fv, int f0, int nf, int kcol, const ScatterArgs& g,
                            Rec* scratch, uint32_t* cnt, const SpillList& sl) {
  const int vcol = fs.vcol < 0 ? 0 : fs.vcol;
  if (fs.vcol < 0)
    return launch_scatter_t<FT, none_t>(grid, lds, s, fv, f0, nf, fs.flt, kcol, vcol, g, scratch, cnt, sl);
  if (fs.vtype == MI355Q_INT64)
    return launch_scatter_t<FT, int64_t>(grid, lds, s, fv, f0, nf, fs.flt, kcol, vcol, g, scratch, cnt, sl);
  if (fs.vtype == MI355Q_INT32)
    return launch_scatter_t<FT, int32_t>(grid, lds, s, fv, f0, nf, fs.flt, kcol, vcol, g, scratch, cnt, sl);
  return launch_scatter_t<FT, double>(grid, lds, s, fv, f0, nf, fs.flt, kcol, vcol, g, scratch, cnt, sl);
}

}  // namespace

bool part_supported(const DevPlan& p, const FragView& fv, int n_cus) {
  FastShape fs;
  if (!grouped_fast_shape(p, fv, &fs)) return false;
  PartPlanHost h;
  return make_part_plan(p, fs, fv, n_cus, kDefaultScratchCap, &h);
}

int64_t part_scratch_bytes(const DevPlan& p, const FragView& fv, int n_cus, int64_t cap_bytes) {
  FastShape fs;
  if (!grouped_fast_shape(p, fv, &fs)) return 0;
  if (cap_bytes <= 0) cap_bytes = kDefaultScratchCap;
  PartPlanHost h;
  if (!make_part_plan(p, fs, fv, n_cus, cap_bytes, &h)) return 0;
  return h.scratch_bytes + 64;
}

hipError_t launch_baseline_partitioned(const DevPlan& p, const FragView& fv, int64_t* out,
                                       int32_t* d_err, void* scratch, int64_t scratch_bytes,
                                       int64_t cap_bytes, int n_cus, hipStream_t s,
                                       LaunchStats* st) {
  FastShape fs;
  if (!grouped_fast_shape(p, fv, &fs)) return hipErrorInvalidValue;
  if (cap_bytes <= 0) cap_bytes = kDefaultScratchCap;
  PartPlanHost h;
  // same inputs as part_scratch_bytes -> the same plan
  if (!make_part_plan(p, fs, fv, n_cus, cap_bytes, &h)) return hipErrorInvalidValue;
  if (h.scratch_bytes + 64 > scratch_bytes) return hipErrorInvalidValue;
  hipEvent_t* ev_pool = st->ev_pool;
  const int n_ev = st->n_ev;
  Rec* recs = (Rec*)scratch;
  uint32_t* cnt = (uint32_t*)((char*)scratch + h.rec_bytes);
  char* spill_base = (char*)scratch + h.rec_bytes + h.cnt_bytes;
  SpillList sl{(uint32_t*)spill_base, (int64_t*)(spill_base + 256), d_err, h.spill_cap, 1 + h.g.ns_int};
  hipError_t e = hipMemsetAsync(spill_base, 0, 256, s);
  if (e != hipSuccess) return e;
  TableArgs tab{};
  tab.out = out;
  tab.entry_count = (uint32_t)p.entry_count;
  tab.row_quad = p.row_quad;
  tab.sp = fs.sp;
  for (int j = 0; j < MI355Q_MAX_SLOTS; ++j) tab.init[j] = p.init_vals[j];
  st->kernel_name = "k_part_scatter";
  st->variant = 2;
  st->n_launches = 0;
  // phase-2 member: the common op sets are compiled in, everything else runs the generic one
  auto agg_kernel = k_part_aggregate<0>;
  switch (h.op_mask) {
    case 1 << SO_COUNT: agg_kernel = k_part_aggregate<(1 << SO_COUNT)>; break;
    case (1 << SO_COUNT) | (1 << SO_SUM_F): agg_kernel = k_part_aggregate<((1 << SO_COUNT) | (1 << SO_SUM_F))>; break;
    case (1 << SO_COUNT) | (1 << SO_SUM_I): agg_kernel = k_part_aggregate<((1 << SO_COUNT) | (1 << SO_SUM_I))>; break;
    case 1 << SO_SUM_F: agg_kernel = k_part_aggregate<(1 << SO_SUM_F)>; break;
    case 1 << SO_SUM_I: agg_kernel = k_part_aggregate<(1 << SO_SUM_I)>; break;
    // nullable value column: COUNT(*), AVG(col) [, COUNT(col)]
    case (1 << SO_COUNT) | (1 << SO_SUM_F) | (1 << SO_COUNT_NN):
      agg_kernel = k_part_aggregate<((1 << SO_COUNT) | (1 << SO_SUM_F) | (1 << SO_COUNT_NN))>;
      break;
    case (1 << SO_COUNT) | (1 << SO_SUM_I) | (1 << SO_COUNT_NN):
      agg_kernel = k_part_aggregate<((1 << SO_COUNT) | (1 << SO_SUM_I) | (1 << SO_COUNT_NN))>;
      break;
    default: break;
  }
  (void)hipFuncSetAttribute((const void*)agg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h.lds2);
  // MI355Q_OPT_TRACE: per-phase cycle counters of phase 2 live in the spill header's tail
  const uint32_t opt_flags = tune_knobs().flags;
  unsigned long long* dbg = (opt_flags & MI355Q_OPT_TRACE) ? (unsigned long long*)(spill_base + 64) : nullptr;
  unsigned int* pair_ctr = (n_cus <= 256 && !(opt_flags & MI355Q_OPT_NO_PAIR_RENDEZVOUS))
                               ? (unsigned int*)((char*)scratch + h.rec_bytes + h.cnt_bytes - kPairCtrBytes) : nullptr;
  ScatterArgs sa{};
  sa.P = h.g.P;
  sa.lgL = h.g.lgL;
  sa.B = h.g.B;
  sa.L = h.g.L;
  sa.cap = h.g.cap;
  sa.hm = h.g.hm;
  sa.ns_int = h.g.ns_int;
  sa.ops_packed = 0;
  sa.n_cand = h.n_cand;
  for (int m = 0; m < h.g.ns_int; ++m) sa.ops_packed |= (uint32_t)(h.ps.int_op[m] & 15) << (4 * m);
  sa.val_nullable = h.ps.val_nullable;
  sa.null_bits = h.ps.null_bits;
  int f = 0;
  int ev_i = 0;
  int chunk = 0;
  if (h.overlap) {
    // ---- phase 1 of chunk i + 1 on g.B CUs next to phase 2 of chunk i on the other grid2 ----
    // stream s: scatter(0), scatter(1), ...  (scatter(i) waits until aggregate(i - 2) has let go of its buffer)
    // stream s2: aggregate(0) + spill merge(0), aggregate(1) + ...   (aggregate(i) waits for scatter(i); one after
    // the other, because chunk i + 1 re-loads the table rows chunk i wrote)
    // Enqueued in an order that is also a valid sequential schedule (the host simulation runs a launch when it is
    // enqueued): scatter(0); then scatter(i + 1), aggregate(i) for every i.
    OverlapRes* ov = overlap_resources();
    if (!ov) return hipErrorInvalidValue;
    struct Chunk { int f0, nf; int64_t rows; };
    std::vector<Chunk> chunks;
    while (f < fv.n_frags) {
      int64_t rows = 0;
      int f1 = f;
      while (f1 < fv.n_frags && (f1 == f || rows + fv.h_num_rows[f1] <= h.chunk_rows)) {
        rows += fv.h_num_rows[f1];
        ++f1;
      }
      chunks.push_back({f, f1 - f, rows});
      f = f1;
    }
    const int n = (int)chunks.size();
    auto buf_recs = [&](int b) { return (Rec*)((char*)scratch + (int64_t)b * h.buf_bytes); };
    auto buf_cnt = [&](int b) { return (uint32_t*)((char*)scratch + (int64_t)b * h.buf_bytes + h.rec_bytes); };
    auto buf_spill = [&](int b) { return (char*)scratch + (int64_t)b * h.buf_bytes + h.rec_bytes + h.cnt_bytes; };
    auto buf_sl = [&](int b) {
      char* sb = buf_spill(b);
      return SpillList{(uint32_t*)sb, (int64_t*)(sb + 256), d_err, h.spill_cap, 1 + h.g.ns_int};
    };
    auto scatter = [&](int i) -> hipError_t {
      const int b = i & 1;
      hipError_t e2;
      if (i >= 2 && (e2 = hipStreamWaitEvent(s, ov->ev_agg[b], 0)) != hipSuccess) return e2;
      if ((e2 = hipMemsetAsync(buf_spill(b), 0, 256, s)) != hipSuccess) return e2;
      if (ev_pool && ev_i + 1 < n_ev) (void)hipEventRecord(ev_pool[ev_i], s);
      const Chunk& c = chunks[i];
      if (fs.fil_type == 0)
        e2 = launch_scatter_v<none_t>(fs, h.g.B, h.lds1, s, fv, c.f0, c.nf, p.group_col, sa, buf_recs(b), buf_cnt(b), buf_sl(b));
      else if (fs.fil_type == MI355Q_INT32)
        e2 = launch_scatter_v<int32_t>(fs, h.g.B, h.lds1, s, fv, c.f0, c.nf, p.group_col, sa, buf_recs(b), buf_cnt(b), buf_sl(b));
      else if (fs.fil_type == MI355Q_INT8)
        e2 = launch_scatter_v<int8_t>(fs, h.g.B, h.lds1, s, fv, c.f0, c.nf, p.group_col, sa, buf_recs(b), buf_cnt(b), buf_sl(b));
      else
        e2 = launch_scatter_v<int64_t>(fs, h.g.B, h.lds1, s, fv, c.f0, c.nf, p.group_col, sa, buf_recs(b), buf_cnt(b), buf_sl(b));
      if (e2 != hipSuccess) return e2;
      if (ev_pool && ev_i + 1 < n_ev) {
        (void)hipEventRecord(ev_pool[ev_i + 1], s);
        ev_i += 2;
      }
      st->n_launches += 1;
      return hipEventRecord(ov->ev_scat[b], s);
    };
    auto aggregate = [&](int i) -> hipError_t {
      const int b = i & 1;
      const Chunk& c = chunks[i];
      hipError_t e2;
      if ((e2 = hipStreamWaitEvent(ov->s2, ov->ev_scat[b], 0)) != hipSuccess) return e2;
      unsigned int* pc = (h.grid2 <= 256 && !(opt_flags & MI355Q_OPT_NO_PAIR_RENDEZVOUS))
                             ? (unsigned int*)((char*)buf_cnt(b) + h.cnt_bytes - kPairCtrBytes) : nullptr;
      if (pc && (e2 = hipMemsetAsync(pc, 0, kPairCtrBytes, ov->s2)) != hipSuccess) return e2;
      const int units = h.g.P * (int)h.g.hm.R;
      const int g2 = units < h.grid2 ? units : h.grid2;
      hipLaunchKernelGGL(agg_kernel, dim3(g2), dim3(kPartBlock), h.lds2, ov->s2, h.g, buf_recs(b), buf_cnt(b), h.ps,
                         tab, buf_sl(b), i > 0 ? 1 : 0, (uint32_t)(c.rows > 0xfff00000ll ? 0xfff00000ll : c.rows),
                         (unsigned long long*)nullptr, pc, SliceMerge{});
      if ((e2 = hipGetLastError()) != hipSuccess) return e2;
      (void)hipFuncSetAttribute((const void*)k_spill_merge, hipFuncAttributeMaxDynamicSharedMemorySize,
                                h.g.ns_int * kSpillLds * 8);
      hipLaunchKernelGGL(k_spill_merge, dim3(512), dim3(256), (size_t)h.g.ns_int * kSpillLds * 8, ov->s2, h.ps, tab,
                         buf_sl(b), h.g.ns_int);
      if ((e2 = hipGetLastError()) != hipSuccess) return e2;
      return hipEventRecord(ov->ev_agg[b], ov->s2);
    };
    // everything enqueued on s so far (table init by the caller, the memsets above) precedes the first aggregate
    // through ev_scat[0]
    if ((e = scatter(0)) != hipSuccess) return e;
    for (int i = 0; i < n; ++i) {
      if (i + 1 < n && (e = scatter(i + 1)) != hipSuccess) return e;
      if ((e = aggregate(i)) != hipSuccess) return e;
    }
    if ((e = hipStreamWaitEvent(s, ov->ev_agg[(n - 1) & 1], 0)) != hipSuccess) return e;
    st->spill_counter32 = (uint32_t*)buf_spill((n - 1) & 1);
    st->n_events_used = ev_i;
    return hipSuccess;
  }
  while (f < fv.n_frags) {
    int64_t rows = 0;
    int f1 = f;
    while (f1 < fv.n_frags && (f1 == f || rows + fv.h_num_rows[f1] <= h.chunk_rows)) {
      rows += fv.h_num_rows[f1];
      ++f1;
    }
    if (ev_pool && ev_i + 1 < n_ev) (void)hipEventRecord(ev_pool[ev_i], s);
    if (fs.fil_type == 0)
      e = launch_scatter_v<none_t>(fs, h.g.B, h.lds1, s, fv, f, f1 - f, p.group_col, sa, recs, cnt, sl);
    else if (fs.fil_type == MI355Q_INT32)
      e = launch_scatter_v<int32_t>(fs, h.g.B, h.lds1, s, fv, f, f1 - f, p.group_col, sa, recs, cnt, sl);
    else if (fs.fil_type == MI355Q_INT8)
      e = launch_scatter_v<int8_t>(fs, h.g.B, h.lds1, s, fv, f, f1 - f, p.group_col, sa, recs, cnt, sl);
    else
      e = launch_scatter_v<int64_t>(fs, h.g.B, h.lds1, s, fv, f, f1 - f, p.group_col, sa, recs, cnt, sl);
    if (e != hipSuccess) return e;
    if (ev_pool && ev_i + 1 < n_ev) {
      (void)hipEventRecord(ev_pool[ev_i + 1], s);
      ev_i += 2;
    }
    st->n_launches += 1;
    const int units = h.g.P * (int)h.g.hm.R;
    const int grid2 = units < h.grid2 ? units : h.grid2;
    if (pair_ctr) {
      e = hipMemsetAsync(pair_ctr, 0, kPairCtrBytes, s);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(agg_kernel, dim3(grid2), dim3(kPartBlock), h.lds2, s, h.g, recs, cnt, h.ps,
                       tab, sl, chunk > 0 ? 1 : 0, (uint32_t)(rows > 0xfff00000ll ? 0xfff00000ll : rows), dbg,
                       pair_ctr, SliceMerge{});
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    (void)hipFuncSetAttribute((const void*)k_spill_merge, hipFuncAttributeMaxDynamicSharedMemorySize,
                              h.g.ns_int * kSpillLds * 8);
    hipLaunchKernelGGL(k_spill_merge, dim3(512), dim3(256), (size_t)h.g.ns_int * kSpillLds * 8, s, h.ps, tab, sl,
                       h.g.ns_int);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // spilled_rows reports the last chunk's list; the word is re-armed for the next chunk
    if (f1 < fv.n_frags) {
      e = hipMemsetAsync(spill_base, 0, 4, s);
      if (e != hipSuccess) return e;
    }
    f = f1;
    ++chunk;
  }
  st->spill_counter32 = (uint32_t*)spill_base;
  st->n_events_used = ev_i;
  if (dbg) {
    unsigned long long h_dbg[16] = {0};
    uint32_t h_sp = 0;
    (void)hipMemcpy(&h_sp, spill_base, 4, hipMemcpyDeviceToHost);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h_dbg, dbg, sizeof(h_dbg), hipMemcpyDeviceToHost);
    const double wg = (double)(h.g.P * (int)h.g.hm.R < n_cus ? h.g.P * (int)h.g.hm.R : n_cus);
    std::fprintf(stderr, "[mi355q] phase 2 Mcycles per workgroup: init %.3f  merge-load %.3f  records %.3f  emit %.3f  empties %.3f | spills %u\n",
                 h_dbg[0] / wg / 1e6, h_dbg[1] / wg / 1e6, h_dbg[2] / wg / 1e6, h_dbg[3] / wg / 1e6, h_dbg[4] / wg / 1e6, h_sp);
  }
  return hipSuccess;
}

// ---------------------------------------------------------------- slice merge (multi-device): host side
namespace {

// The slot program of a table known only by its layout (mi355q_result_create: no input columns): the
// merge of partial rows does not care which column a slot was fed from.
bool slice_merge_plan(const DevPlan& p, int n_cus, FastShape* fs, PartPlanHost* h) {
  if (p.desc_type != MI355Q_GROUP_BY_BASELINE_HASH || p.n_group != 1 || p.key_width != 8 || p.slot_width != 8 ||
      p.key_quad != 1)
    return false;
  DevPlan q = p;
  q.join_col = -1;
  q.n_quals = 0;
  for (int i = 0; i < q.n_targets; ++i) {
    DevTarget& t = q.targets[i];
    t.table = 0;
    if (t.arg_f32) return false;
    if (t.agg == MI355Q_PROJECT_KEY || (t.agg == MI355Q_COUNT && !t.skip_null)) continue;
    t.col = 0;
    t.arg_type = t.arg_fp ? MI355Q_DOUBLE : MI355Q_INT64;
  }
  FragView fv{};
  fv.total_rows = 1;
  fv.max_frag_rows = 1;
  if (!grouped_fast_shape(q, fv, fs)) return false;
  return make_part_plan(q, *fs, fv, n_cus, kDefaultScratchCap, h);
}

uint32_t slice_merge_spill_cap(int64_t lo, int64_t hi) {
  // strays of clusters that cross a unit boundary, partial counts too large for a 32-bit LDS counter, rows
  // the two-choice table had no room for: a small fraction of the range (an overflow is reported through
  // d_err, the caller then folds with mi355q_shard_merge_range)
  const int64_t c = (hi - lo) / 4 + 65536;
  return (uint32_t)(c > 0x7fffffffll ? 0x7fffffffll : c);
}

}  // namespace

int64_t slice_merge_scratch_bytes(const DevPlan& p, int64_t lo, int64_t hi, int n_cus) {
  FastShape fs;
  PartPlanHost h;
  if (!slice_merge_plan(p, n_cus, &fs, &h)) return 0;
  return 256 + (int64_t)slice_merge_spill_cap(lo, hi) * 8 * (1 + h.g.ns_int) + 64;
}

// Folds rows [lo, hi) of `n_src` tables (src[i] = the address of row `lo` of table i) and the pad_rows
// rows that followed each into rows [lo, hi) of `out`, keeping only keys whose home slot is in [lo, hi).
// Rows [lo, hi) of `out` are overwritten (canonical emission per unit); strays are CAS-merged afterwards
// and may land anywhere from their home slot on, like any late insert.
hipError_t launch_slice_merge(const DevPlan& p, int64_t* out, const int64_t* const* src, const int64_t* const* pads,
                              int n_src, int pad_rows, int64_t lo, int64_t hi, int32_t* d_err, void* scratch,
                              int64_t scratch_bytes, int n_cus, hipStream_t s) {
  if (n_src < 1 || n_src > kMaxMergeSrc || lo < 0 || hi <= lo || hi > p.entry_count) return hipErrorInvalidValue;
  FastShape fs;
  PartPlanHost h;
  if (!slice_merge_plan(p, n_cus, &fs, &h)) return hipErrorInvalidValue;
  const uint32_t spill_cap = slice_merge_spill_cap(lo, hi);
  if (256 + (int64_t)spill_cap * 8 * (1 + h.g.ns_int) > scratch_bytes) return hipErrorInvalidValue;
  char* spill_base = (char*)scratch;
  SpillList sl{(uint32_t*)spill_base, (int64_t*)(spill_base + 256), d_err, spill_cap, 1 + h.g.ns_int};
  hipError_t e = hipMemsetAsync(spill_base, 0, 256, s);
  if (e != hipSuccess) return e;
  TableArgs tab{};
  tab.out = out;
  tab.entry_count = (uint32_t)p.entry_count;
  tab.row_quad = p.row_quad;
  tab.sp = fs.sp;
  for (int j = 0; j < MI355Q_MAX_SLOTS; ++j) tab.init[j] = p.init_vals[j];
  SliceMerge ms{};
  for (int i = 0; i < n_src; ++i) {
    ms.src[i] = src[i];
    ms.pads[i] = pads ? pads[i] : nullptr;
  }
  ms.n_src = n_src;
  ms.pad_rows = pads ? pad_rows : 0;
  ms.lo = (uint32_t)lo;
  ms.hi = (uint32_t)hi;
  auto agg_kernel = k_part_aggregate<0>;  // no records in this mode: the op-set members gain nothing
  (void)hipFuncSetAttribute((const void*)agg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds2);
  const int units = h.g.P * (int)h.g.hm.R;
  const int grid2 = units < n_cus ? units : n_cus;
  // 32-bit LDS counters: n_src partial counts are added up per group; a partial that could make the sum
  // wrap goes through the 64-bit spill merge instead
  const uint32_t big_from = 0xffffffffu - 0xffffffffu / (uint32_t)(n_src + 1);
  hipLaunchKernelGGL(agg_kernel, dim3(grid2), dim3(kPartBlock), h.lds2, s, h.g, (const Rec*)nullptr,
                     (const uint32_t*)nullptr, h.ps, tab, sl, 0, big_from, (unsigned long long*)nullptr,
                     (unsigned int*)nullptr, ms);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  (void)hipFuncSetAttribute((const void*)k_spill_merge, hipFuncAttributeMaxDynamicSharedMemorySize,
                            h.g.ns_int * kSpillLds * 8);
  hipLaunchKernelGGL(k_spill_merge, dim3(512), dim3(256), (size_t)h.g.ns_int * kSpillLds * 8, s, h.ps, tab, sl,
                     h.g.ns_int);
  return hipGetLastError();
}

// ---------------------------------------------------------------- radix join probe: host side
namespace {

struct JoinPartHost {
  ScatterArgs sa;
  JoinPartArgs ja;
  int64_t chunk_rows, rec_bytes, cnt_bytes, scratch_bytes;
  size_t lds1, lds2;
  uint32_t spill_cap;
  int vcol;  // fact value column or -1
};

// the plan shapes this family takes: non-grouped, INNER one-to-one perfect join on a NOT NULL
// int64 key with the presence bitmap available, targets COUNT(*) / SUM(fact int64 NOT NULL col)
bool make_join_part_plan(const DevPlan& p, const FragView& fv, int n_cus, int64_t scratch_cap, JoinPartHost* out) {
  JoinPartHost& h = *out;
  if (p.desc_type != MI355Q_NON_GROUPED_AGGREGATE || p.join_col < 0 || p.n_quals != 0) return false;
  if (p.join_hash_type != 0 || !p.join_bitmap || p.join_n_keys != 1 || p.join_kind != MI355Q_JOIN_INNER) return false;
  if (p.join_type != MI355Q_INT64 || p.join_nullable || p.n_targets > 4) return false;
  if (fv.max_frag_rows > 0xfff00000ll) return false;
  h.vcol = -1;
  h.ja = JoinPartArgs{};
  h.ja.n_slots = p.n_targets;
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    if (t.slot != i) return false;
    if (t.agg == MI355Q_COUNT && t.col < 0) {
      h.ja.op[i] = 0;
    } else if (t.agg == MI355Q_SUM && t.table == 0 && t.arg_type == MI355Q_INT64 && !t.arg_nullable) {
      if (h.vcol >= 0 && h.vcol != t.col) return false;
      h.vcol = t.col;
      h.ja.op[i] = 1;
    } else {
      return false;
    }
  }
  if (!all_aligned16(fv, p.join_col) || (h.vcol >= 0 && !all_aligned16(fv, h.vcol))) return false;
  const __int128 range128 = (__int128)p.join_max - (__int128)p.join_min + 1;
  if (range128 < 64 || range128 >= ((__int128)1 << 32)) return false;
  const uint64_t range = (uint64_t)range128;
  // partitions: at least one per CU, bitmap slice within ~120 KB of LDS
  uint32_t P = 16;
  auto s1_of = [&](uint32_t parts) { return (uint32_t)((((range + parts - 1) / parts) + 31) & ~(uint64_t)31); };
  // as many partitions as the scatter supports: with 256 the 12 producer waves of a workgroup
  // contend on 256 LDS cursors and stage 512-byte lines (9.1 ms per 1.43 B rows); 1024
  // partitions = the geometry the scatter was tuned on
  while (P < 1024 && (P < 4 * (uint32_t)n_cus || s1_of(P) / 8 > 120 * 1024) && s1_of(P * 2) >= 64) P <<= 1;
  const uint32_t S1 = s1_of(P);
  if (S1 / 8 > 150 * 1024 || S1 < 32) return false;
  ScatterArgs& sa = h.sa;
  sa = ScatterArgs{};
  sa.P = (int32_t)P;
  sa.L = kStageRecs / P;
  sa.lgL = 0;
  while ((1u << sa.lgL) < sa.L) ++sa.lgL;
  sa.B = n_cus;
  sa.hm.d = (uint32_t)range;
  sa.hm.S1 = S1;
  sa.hm.S2 = S1;
  sa.hm.R = 1;
  sa.hm.d_rcp = (uint32_t)(((uint64_t)1 << 32) / sa.hm.d);
  sa.hm.s1_rcp = (uint32_t)(((uint64_t)1 << 32) / S1);
  sa.kmin = p.join_min;
  // partial rows of spilled / heavy-hitter records: SUM, COUNT, COUNT of non-NULL values
  sa.ns_int = 3;
  sa.ops_packed = (uint32_t)SO_SUM_I | ((uint32_t)SO_COUNT << 4) | ((uint32_t)SO_COUNT_NN << 8);
  sa.val_nullable = 1;  // the non-grouped SUM skips NULL_BIGINT even on a NOT NULL column
  sa.null_bits = INT64_MIN;
  int64_t chunk_rows = fv.total_rows > 0 ? fv.total_rows : 1;
  if (chunk_rows > 0xfff00000ll) chunk_rows = 0xfff00000ll;
  if (scratch_cap <= 0) scratch_cap = kDefaultScratchCap;
  for (;;) {
    const double per_run = (double)chunk_rows / ((double)P * sa.B);
    uint64_t cap = (uint64_t)(per_run * 1.2 + 6.0 * __builtin_sqrt(per_run + 1.0)) + sa.L;
    cap = (cap + sa.L - 1) / sa.L * sa.L;
    const bool too_many = cap > 0x7fffffffull || (uint64_t)P * sa.B * cap >= ((uint64_t)1 << 32);
    int64_t spill_cap = chunk_rows / 16;
    if (spill_cap < (int64_t)kSpillMin) spill_cap = kSpillMin;
    if (spill_cap > 0x7fffffffll) spill_cap = 0x7fffffffll;
    h.rec_bytes = (int64_t)P * sa.B * (int64_t)cap * (int64_t)sizeof(Rec);
    h.cnt_bytes = ((int64_t)P * sa.B * 4 + 255) & ~255ll;
    const int64_t spill_bytes = 256 + spill_cap * 8 * (int64_t)(1 + sa.ns_int);
    h.scratch_bytes = h.rec_bytes + h.cnt_bytes + spill_bytes;
    if (!too_many && (h.scratch_bytes <= scratch_cap || chunk_rows <= fv.max_frag_rows)) {
      sa.cap = (uint32_t)cap;
      h.spill_cap = (uint32_t)spill_cap;
      break;
    }
    if (chunk_rows <= fv.max_frag_rows) return false;
    chunk_rows = (int64_t)(chunk_rows * 0.97);
    if (chunk_rows < fv.max_frag_rows) chunk_rows = fv.max_frag_rows;
  }
  h.chunk_rows = chunk_rows;
  {
    const size_t fixed = kStageRecs * sizeof(Rec) + (size_t)P * 12 + 48 + (size_t)kHotSlots * (8 + 8 * (size_t)sa.ns_int);
    sa.n_cand = 2048;
    while (sa.n_cand > 64 && fixed + (size_t)sa.n_cand * 4 > 160 * 1024) sa.n_cand >>= 1;
    h.lds1 = fixed + (size_t)sa.n_cand * 4;
    if (h.lds1 > 160 * 1024) return false;
  }
  h.lds2 = (size_t)(S1 / 8) + 16 * 3 * 8;
  JoinPartArgs& ja = h.ja;
  ja.P = (int32_t)P;
  ja.B = sa.B;
  ja.cap = sa.cap;
  ja.S1 = S1;
  ja.kmin = p.join_min;
  ja.range = range;
  ja.bitmap = p.join_bitmap;
  ja.bm_words = (int64_t)((range + 31) / 32);
  ja.null_sum = INT64_MIN;
  return true;
}

}  // namespace

bool join_part_supported(const DevPlan& p, const FragView& fv, int n_cus) {
  JoinPartHost h;
  return make_join_part_plan(p, fv, n_cus, kDefaultScratchCap, &h);
}

int64_t join_part_scratch_bytes(const DevPlan& p, const FragView& fv, int n_cus, int64_t cap_bytes) {
  JoinPartHost h;
  if (!make_join_part_plan(p, fv, n_cus, cap_bytes, &h)) return 0;
  return h.scratch_bytes + 64;
}

hipError_t launch_join_partitioned(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err,
                                   void* scratch, int64_t scratch_bytes, int64_t cap_bytes, int n_cus,
                                   hipStream_t s, LaunchStats* st) {
  JoinPartHost h;
  if (!make_join_part_plan(p, fv, n_cus, cap_bytes, &h)) return hipErrorInvalidValue;
  if (h.scratch_bytes + 64 > scratch_bytes) return hipErrorInvalidValue;
  Rec* recs = (Rec*)scratch;
  uint32_t* cnt = (uint32_t*)((char*)scratch + h.rec_bytes);
  char* spill_base = (char*)scratch + h.rec_bytes + h.cnt_bytes;
  SpillList sl{(uint32_t*)spill_base, (int64_t*)(spill_base + 256), d_err, h.spill_cap, 1 + h.sa.ns_int};
  st->kernel_name = "k_part_scatter";
  st->variant = 2;
  st->n_launches = 0;
  (void)hipFuncSetAttribute((const void*)k_part_join, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds2);
  RangeFilter flt = no_filter();
  const int vcol = h.vcol < 0 ? 0 : h.vcol;
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
    if (st->ev_pool && ev_i + 1 < st->n_ev) (void)hipEventRecord(st->ev_pool[ev_i], s);
    const int8_t* const* cols = fv.d_cols + (size_t)f * fv.n_cols;
    const int64_t* nrows = fv.d_num_rows + f;
    if (h.vcol >= 0) {
      (void)hipFuncSetAttribute((const void*)k_part_scatter<none_t, int64_t, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds1);
      hipLaunchKernelGGL((k_part_scatter<none_t, int64_t, 1>), dim3(h.sa.B), dim3(kPartBlock), h.lds1, s, cols,
                         nrows, f1 - f, fv.n_cols, flt, p.join_col, vcol, h.sa, recs, cnt, sl);
    } else {
      (void)hipFuncSetAttribute((const void*)k_part_scatter<none_t, none_t, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds1);
      hipLaunchKernelGGL((k_part_scatter<none_t, none_t, 1>), dim3(h.sa.B), dim3(kPartBlock), h.lds1, s, cols,
                         nrows, f1 - f, fv.n_cols, flt, p.join_col, vcol, h.sa, recs, cnt, sl);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (st->ev_pool && ev_i + 1 < st->n_ev) {
      (void)hipEventRecord(st->ev_pool[ev_i + 1], s);
      ev_i += 2;
    }
    st->n_launches += 1;
    const int grid2 = h.ja.P < n_cus ? h.ja.P : n_cus;
    hipLaunchKernelGGL(k_part_join, dim3(grid2), dim3(kPartBlock), h.lds2, s, h.ja, recs, cnt, out);
    hipLaunchKernelGGL(k_join_spill, dim3(256), dim3(256), 0, s, h.ja, sl, out);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    f = f1;
  }
  st->spill_counter32 = (uint32_t*)spill_base;
  st->n_events_used = ev_i;
  return hipSuccess;
}

// ---------------------------------------------------------------- payload probe: host side
hipError_t launch_join_payload_build(const void* table, int hash_type, int64_t entries, const void* inner_col,
                                     uint32_t* cnt_k, int64_t* wsum_k, uint32_t* wnn_k, void* pay16, int64_t* pay8,
                                     int32_t* d_flags, int n_cus, hipStream_t s) {
  int64_t blocks = (entries + 255) / 256;
  if (blocks > (int64_t)n_cus * 16) blocks = (int64_t)n_cus * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_join_payload, dim3((unsigned)blocks), dim3(256), 0, s, (const int32_t*)table, hash_type, entries,
                     (const int64_t*)inner_col, cnt_k, wsum_k, wnn_k, (Pay16*)pay16, pay8, d_flags);
  return hipGetLastError();
}

hipError_t launch_join_payload_keyed_build(const void* table, int hash_type, int64_t entries, const void* inner_col,
                                           int64_t* kkeys, void* pay16, int64_t* pay8, int32_t* d_flags, int n_cus,
                                           hipStream_t s) {
  int64_t blocks = (entries + 255) / 256;
  if (blocks > (int64_t)n_cus * 16) blocks = (int64_t)n_cus * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_join_payload_keyed, dim3((unsigned)blocks), dim3(256), 0, s, (const int64_t*)table, hash_type,
                     entries, (const int64_t*)inner_col, kkeys, (Pay16*)pay16, pay8, d_flags);
  return hipGetLastError();
}

namespace {

struct ProbePartHost {
  ScatterArgs sa;
  ProbeArgs pa;
  ProbeFinish fin;
  int64_t chunk_rows, rec_bytes, cnt_bytes, scratch_bytes;
  size_t lds1, lds2;
  uint32_t spill_cap;
  int vcol;    // outer value column or -1
  int wcol;    // inner column or -1
  bool l2_mode;  // slices in L2 (k_part_probe_l2) instead of LDS (k_part_probe)
  bool keyed;    // keyed join table: partitioned by hash slot, probed in L2 (k_part_probe_keyed)
  bool keys_only;  // ... one-to-one and no inner column read: the probe fetches the slot keys alone (8 B per slot)
};

constexpr size_t kProbeLdsBudget = 144 * 1024;

// plan shapes: non-grouped, no quals, perfect-hash table (one-to-one or one-to-many), one NOT NULL
// int64 key, INNER or LEFT; targets COUNT(*), SUM / COUNT of ONE plain int64 outer column and of ONE
// plain int64 inner column
bool make_probe_plan(const DevPlan& p, const FragView& fv, const JoinPayloadView& pay, int n_cus, int64_t scratch_cap,
                     ProbePartHost* out) {
  ProbePartHost& h = *out;
  if (p.desc_type != MI355Q_NON_GROUPED_AGGREGATE || p.join_col < 0 || p.n_quals != 0) return false;
  if (p.join_hash_type < 0 || p.join_hash_type > 3 || p.join_n_keys != 1) return false;
  h.keyed = p.join_hash_type == 1 || p.join_hash_type == 3;
  h.keys_only = false;
  if (h.keyed && p.join_width != 8) return false;
  if (p.join_kind != MI355Q_JOIN_INNER && p.join_kind != MI355Q_JOIN_LEFT) return false;
  if (p.join_type != MI355Q_INT64 || p.join_nullable || p.n_targets > 4) return false;
  if (fv.max_frag_rows > 0xfff00000ll) return false;
  h.vcol = h.wcol = -1;
  h.fin = ProbeFinish{};
  h.fin.n_slots = p.n_targets;
  h.fin.left = p.join_kind == MI355Q_JOIN_LEFT;
  h.fin.n_outer_rows = fv.total_rows;
  h.fin.null_sum = INT64_MIN;
  for (int i = 0; i < p.n_targets; ++i) {
    const DevTarget& t = p.targets[i];
    if (t.slot != i) return false;
    if (t.agg == MI355Q_COUNT && t.col < 0) {
      h.fin.op[i] = 0;
      continue;
    }
    if ((t.agg != MI355Q_SUM && t.agg != MI355Q_COUNT) || t.arg_type != MI355Q_INT64) return false;
    if (t.table == 0) {
      if (h.vcol >= 0 && h.vcol != t.col) return false;
      h.vcol = t.col;
      h.fin.op[i] = t.agg == MI355Q_SUM ? 1 : 3;
    } else {
      if (h.wcol >= 0 && h.wcol != t.col) return false;
      h.wcol = t.col;
      h.fin.op[i] = t.agg == MI355Q_SUM ? 2 : 4;
    }
  }
  if (!all_aligned16(fv, p.join_col) || (h.vcol >= 0 && !all_aligned16(fv, h.vcol))) return false;
  // perfect tables: the partitions are key ranges; keyed tables: ranges of table slots
  const __int128 range128 = h.keyed ? (__int128)p.join_entries : (__int128)p.join_max - (__int128)p.join_min + 1;
  if (range128 < 64 || range128 >= ((__int128)1 << 32)) return false;
  const uint64_t range = (uint64_t)range128;
  // the payload arrays must be there for exactly this inner column (api.cpp builds them first)
  if (pay.entries != (int64_t)range) return false;
  if (h.wcol >= 0 && pay.inner_col != (const void*)p.inner_cols[h.wcol]) return false;
  const bool need_nn = h.wcol >= 0 && pay.has_nulls;
  const size_t entry_bytes = 4 + (h.wcol >= 0 ? 8 : 0) + (need_nn ? 4 : 0);
  uint32_t P = 16;
  auto s1_of = [&](uint32_t parts) { return (uint32_t)((((range + parts - 1) / parts) + 31) & ~(uint64_t)31); };
  // geometry and mode are decided on the widest entry this plan could need (so they do not depend on
  // whether the inner column turned out to hold NULLs); the LDS slices then use the actual width
  const size_t worst_bytes = 4 + (h.wcol >= 0 ? 12 : 0);
  while (P < 1024 && (P < 4 * (uint32_t)n_cus || (size_t)s1_of(P) * worst_bytes > kProbeLdsBudget) && s1_of(P * 2) >= 64) P <<= 1;
  const uint32_t S1 = s1_of(P);
  if (S1 < 32) return false;
  uint32_t R = (uint32_t)(((size_t)S1 * entry_bytes + kProbeLdsBudget - 1) / kProbeLdsBudget);
  if (R < 1) R = 1;
  // every LDS sub-range re-reads the partition's records: beyond 3 the slice stays in L2 instead
  // (16-byte entries, at most 2 MB per partition so two or three live slices fit the XCD's 4 MB)
  h.l2_mode = h.keyed || ((size_t)S1 * worst_bytes + kProbeLdsBudget - 1) / kProbeLdsBudget > 3;
  if (h.keyed) {
    // slice = keys (8 B per slot) + payload (8 or 16 B per slot): passes of at most 3.5 MB of it.  Measured on
    // cfg4's sparse 200 M-slot table (3.1 MB per partition, 3.2 B rows, profiles/r02_keyed_probe_passes.jsonl):
    // one pass 89 ms, two passes of 1.6 MB 114 ms (the second read of the records costs more than the smaller
    // slice gives), direct probe 140 ms; without pacing the XCD group 113 / 129 ms
    if (!pay.kkeys || (!pay.pay16 && !pay.pay8)) return false;
    h.keys_only = p.join_hash_type == 1 && h.wcol < 0;
    R = (uint32_t)(((size_t)S1 * (h.keys_only ? 8 : 16) + ((size_t)7 << 19) - 1) / ((size_t)7 << 19));
    if (R < 1) R = 1;
    {  // tests: several passes on a small table
      const int v = tune_knobs().probe_keyed_passes;
      if (v >= 1 && v <= 4) R = (uint32_t)v;
    }
    if (R > 4) return false;
  } else if (h.l2_mode) {
    if ((size_t)S1 * sizeof(Pay16) > ((size_t)2 << 20)) return false;
    if (!pay.pay16 && !pay.pay8) return false;
    R = 1;
  } else if (!pay.cnt_k || (h.wcol >= 0 && !pay.wsum_k)) {
    return false;
  }
  const uint32_t S2 = (uint32_t)((((uint64_t)S1 + R - 1) / R + 31) & ~(uint64_t)31);
  ScatterArgs& sa = h.sa;
  sa = ScatterArgs{};
  sa.P = (int32_t)P;
  sa.L = kStageRecs / P;
  sa.lgL = 0;
  while ((1u << sa.lgL) < sa.L) ++sa.lgL;
  sa.B = n_cus;
  sa.hm.d = (uint32_t)range;
  sa.hm.S1 = S1;
  sa.hm.S2 = S1;
  sa.hm.R = 1;
  sa.hm.d_rcp = (uint32_t)(((uint64_t)1 << 32) / sa.hm.d);
  sa.hm.s1_rcp = (uint32_t)(((uint64_t)1 << 32) / S1);
  sa.kmin = h.keyed ? 0 : p.join_min;
  sa.ns_int = 3;  // spilled / heavy-hitter partial rows: SUM, COUNT, COUNT of non-NULL values
  sa.ops_packed = (uint32_t)SO_SUM_I | ((uint32_t)SO_COUNT << 4) | ((uint32_t)SO_COUNT_NN << 8);
  sa.val_nullable = 1;
  sa.null_bits = INT64_MIN;
  int64_t chunk_rows = fv.total_rows > 0 ? fv.total_rows : 1;
  if (chunk_rows > 0xfff00000ll) chunk_rows = 0xfff00000ll;
  if (scratch_cap <= 0) scratch_cap = kDefaultScratchCap;
  for (;;) {
    const double per_run = (double)chunk_rows / ((double)P * sa.B);
    uint64_t cap = (uint64_t)(per_run * 1.2 + 6.0 * __builtin_sqrt(per_run + 1.0)) + sa.L;
    cap = (cap + sa.L - 1) / sa.L * sa.L;
    const bool too_many = cap > 0x7fffffffull || (uint64_t)P * sa.B * cap >= ((uint64_t)1 << 32);
    int64_t spill_cap = chunk_rows / 16;
    if (spill_cap < (int64_t)kSpillMin) spill_cap = kSpillMin;
    if (spill_cap > 0x7fffffffll) spill_cap = 0x7fffffffll;
    h.rec_bytes = (int64_t)P * sa.B * (int64_t)cap * (int64_t)sizeof(Rec);
    h.cnt_bytes = ((int64_t)P * sa.B * 4 + 255) & ~255ll;
    const int64_t spill_bytes = 256 + spill_cap * 8 * (int64_t)(1 + sa.ns_int);
    h.scratch_bytes = h.rec_bytes + h.cnt_bytes + spill_bytes;
    if (!too_many && (h.scratch_bytes <= scratch_cap || chunk_rows <= fv.max_frag_rows)) {
      sa.cap = (uint32_t)cap;
      h.spill_cap = (uint32_t)spill_cap;
      break;
    }
    if (chunk_rows <= fv.max_frag_rows) return false;
    chunk_rows = (int64_t)(chunk_rows * 0.97);
    if (chunk_rows < fv.max_frag_rows) chunk_rows = fv.max_frag_rows;
  }
  if (fv.total_rows > chunk_rows) {
    const int64_t n_chunks = (fv.total_rows + chunk_rows - 1) / chunk_rows;
    const int64_t even = (fv.total_rows + n_chunks - 1) / n_chunks + fv.max_frag_rows;
    if (even < chunk_rows) chunk_rows = even;
  }
  h.chunk_rows = chunk_rows;
  {
    const size_t fixed = kStageRecs * sizeof(Rec) + (size_t)P * 12 + 48 + (size_t)kHotSlots * (8 + 8 * (size_t)sa.ns_int);
    sa.n_cand = 2048;
    while (sa.n_cand > 64 && fixed + (size_t)sa.n_cand * 4 > 160 * 1024) sa.n_cand >>= 1;
    h.lds1 = fixed + (size_t)sa.n_cand * 4;
    if (h.lds1 > 160 * 1024) return false;
  }
  h.lds2 = h.l2_mode ? 0 : (size_t)S2 * entry_bytes + (size_t)16 * PA_N * 8 + 64;
  if (h.lds2 > 160 * 1024) return false;
  ProbeArgs& pa = h.pa;
  pa = ProbeArgs{};
  pa.P = (int32_t)P;
  pa.B = sa.B;
  pa.R = (int32_t)R;
  pa.cap = sa.cap;
  pa.S1 = S1;
  pa.S2 = S2;
  pa.kmin = h.keyed ? 0 : p.join_min;
  pa.range = range;
  pa.cnt_k = h.l2_mode ? nullptr : pay.cnt_k;
  pa.wsum_k = (!h.l2_mode && h.wcol >= 0) ? pay.wsum_k : nullptr;
  pa.wnn_k = (!h.l2_mode && need_nn) ? pay.wnn_k : nullptr;
  pa.kkeys = h.keyed ? pay.kkeys : nullptr;
  pa.pay8 = (h.l2_mode && pay.pay8 && (p.join_hash_type == 0 || p.join_hash_type == 1) && !pay.has_nulls) ? pay.pay8
                                                                                                            : nullptr;
  pa.pay16 = (h.l2_mode && !pa.pay8) ? (const Pay16*)pay.pay16 : nullptr;
  if (h.l2_mode && !pa.pay8 && !pa.pay16) return false;
  pa.null_sum = INT64_MIN;
  pa.range_rcp = (range >= 2 && range < ((uint64_t)1 << 32)) ? (uint32_t)(((uint64_t)1 << 32) / range) : 0u;
  return true;
}

}  // namespace

bool join_probe_supported(const DevPlan& p, const FragView& fv, const JoinPayloadView& pay, int n_cus) {
  ProbePartHost h;
  return make_probe_plan(p, fv, pay, n_cus, kDefaultScratchCap, &h);
}

// which inner column (if any) the payload of this plan has to be built for; false = shape not taken
bool join_probe_wants(const DevPlan& p, const FragView& fv, int n_cus, int* inner_col, int* l2_mode) {
  JoinPayloadView fake{};
  const bool keyed = p.join_hash_type == 1 || p.join_hash_type == 3;
  const __int128 range128 = keyed ? (__int128)p.join_entries : (__int128)p.join_max - (__int128)p.join_min + 1;
  if (range128 < 64 || range128 >= ((__int128)1 << 32)) return false;
  fake.entries = (int64_t)range128;
  fake.cnt_k = (const uint32_t*)16;  // shape test only: the pointers are not followed
  fake.wsum_k = (const int64_t*)16;
  fake.wnn_k = (const uint32_t*)16;
  fake.pay16 = (const void*)16;
  fake.pay8 = nullptr;
  fake.kkeys = (const int64_t*)16;
  fake.has_nulls = 1;
  int wcol = -1;
  for (int i = 0; i < p.n_targets; ++i)
    if (p.targets[i].table == 1 && p.targets[i].col >= 0) wcol = p.targets[i].col;
  fake.inner_col = wcol >= 0 ? (const void*)p.inner_cols[wcol] : nullptr;
  ProbePartHost h;
  // (the geometry — partitions, slice sizes, L2 mode — depends on the CU count: the device's own, not a nominal 256)
  if (!make_probe_plan(p, fv, fake, n_cus, kDefaultScratchCap, &h)) return false;
  *inner_col = h.wcol;
  *l2_mode = h.keyed ? 2 : h.l2_mode ? 1 : 0;
  return true;
}

int64_t join_probe_scratch_bytes(const DevPlan& p, const FragView& fv, const JoinPayloadView& pay, int n_cus,
                                 int64_t cap_bytes) {
  ProbePartHost h;
  if (!make_probe_plan(p, fv, pay, n_cus, cap_bytes, &h)) return 0;
  return h.scratch_bytes + 64 + 512;
}

hipError_t launch_join_probe(const DevPlan& p, const FragView& fv, const JoinPayloadView& pay, int64_t* out,
                             int32_t* d_err, void* scratch, int64_t scratch_bytes, int64_t cap_bytes, int n_cus,
                             hipStream_t s, LaunchStats* st) {
  ProbePartHost h;
  if (!make_probe_plan(p, fv, pay, n_cus, cap_bytes, &h)) return hipErrorInvalidValue;
  if (h.scratch_bytes + 64 + 512 > scratch_bytes) return hipErrorInvalidValue;
  Rec* recs = (Rec*)scratch;
  uint32_t* cnt = (uint32_t*)((char*)scratch + h.rec_bytes);
  char* spill_base = (char*)scratch + h.rec_bytes + h.cnt_bytes;
  SpillList sl{(uint32_t*)spill_base, (int64_t*)(spill_base + 256), d_err, h.spill_cap, 1 + h.sa.ns_int};
  // accumulators live behind the spill list
  unsigned long long* acc = (unsigned long long*)((char*)scratch + h.scratch_bytes + 64 - 64);
  acc = (unsigned long long*)(((uintptr_t)acc + 63) & ~(uintptr_t)63);
  unsigned long long* acc2 = acc + PA_N;
  hipError_t e = hipMemsetAsync(acc, 0, (PA_N + 2) * 8 + 64, s);
  if (e != hipSuccess) return e;
  st->kernel_name = "k_part_scatter";
  st->variant = 3;
  st->n_launches = 0;
  (void)hipFuncSetAttribute((const void*)k_part_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds2);
  RangeFilter flt = no_filter();
  const int vcol = h.vcol < 0 ? 0 : h.vcol;
  int ev_i = 0;
  int f = 0;
  while (f < fv.n_frags) {
    int64_t rows = 0;
    int f1 = f;
    while (f1 < fv.n_frags && (f1 == f || rows + fv.h_num_rows[f1] <= h.chunk_rows)) {
      rows += fv.h_num_rows[f1];
      ++f1;
    }
    e = hipMemsetAsync(spill_base, 0, 256, s);
    if (e != hipSuccess) return e;
    if (st->ev_pool && ev_i + 1 < st->n_ev) (void)hipEventRecord(st->ev_pool[ev_i], s);
    const int8_t* const* cols = fv.d_cols + (size_t)f * fv.n_cols;
    const int64_t* nrows = fv.d_num_rows + f;
    auto scatter = [&](auto kern) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h.lds1);
      hipLaunchKernelGGL(kern, dim3(h.sa.B), dim3(kPartBlock), h.lds1, s, cols, nrows, f1 - f, fv.n_cols, flt, p.join_col,
                         vcol, h.sa, recs, cnt, sl);
    };
    if (h.keyed) {
      if (h.vcol >= 0) scatter(k_part_scatter<none_t, int64_t, 2>);
      else scatter(k_part_scatter<none_t, none_t, 2>);
    } else {
      if (h.vcol >= 0) scatter(k_part_scatter<none_t, int64_t, 1>);
      else scatter(k_part_scatter<none_t, none_t, 1>);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (st->ev_pool && ev_i + 1 < st->n_ev) {
      (void)hipEventRecord(st->ev_pool[ev_i + 1], s);
      ev_i += 2;
    }
    st->n_launches += 1;
    unsigned int* pace = (h.l2_mode && !(tune_knobs().flags & MI355Q_OPT_PROBE_NO_PACING)) ? (unsigned int*)(acc2 + 2) : nullptr;
    if (pace) {
      e = hipMemsetAsync(pace, 0, 64, s);
      if (e != hipSuccess) return e;
    }
    if (h.keyed) {
      // one 1024-lane workgroup per CU: 16 waves, each with its own ring of records in LDS
      const dim3 pg(n_cus);
      auto probe = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kProbeKeyedLds);
        hipLaunchKernelGGL(kern, pg, dim3(1024), kProbeKeyedLds, s, h.pa, recs, cnt, acc, pace);
      };
      // (sequences per lane, slots per round: profiles/r04_cfg4_keyed_probe_variants_call*.jsonl)
      if (h.keys_only) probe(k_part_probe_keyed<2, 4, 2>);
      else if (h.pa.pay8) probe(k_part_probe_keyed<1, 2, 2>);
      else probe(k_part_probe_keyed<0, 2, 2>);
    } else if (h.l2_mode) {
      // one 1024-lane workgroup per CU (measured, 3.2 B rows: 2048 workgroups of 256 lanes stream the
      // records at 2.1 TB/s and take 52.8 ms; 256 of 1024 lanes stream at 6.5 TB/s and take 25.7 ms;
      // four or eight records per lane per step: no difference)
      if (h.pa.pay8)
        hipLaunchKernelGGL((k_part_probe_l2<1024, true, 4>), dim3(n_cus), dim3(1024), 0, s, h.pa, recs, cnt, acc, pace);
      else
        hipLaunchKernelGGL((k_part_probe_l2<1024, false, 4>), dim3(n_cus), dim3(1024), 0, s, h.pa, recs, cnt, acc, pace);
    } else {
      const int units = h.pa.P * h.pa.R;
      const int grid2 = units < n_cus ? units : n_cus;
      hipLaunchKernelGGL(k_part_probe, dim3(grid2), dim3(kPartBlock), h.lds2, s, h.pa, recs, cnt, acc);
    }
    hipLaunchKernelGGL(k_probe_spill, dim3(256), dim3(256), 0, s, h.pa, sl, acc);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    f = f1;
  }
  if (h.fin.left && h.vcol >= 0) {
    int64_t want = (fv.total_rows / 4 + kBlock - 1) / kBlock;
    if (want < 1) want = 1;
    const int grid = (int)(want < (int64_t)n_cus * 2 ? want : (int64_t)n_cus * 2);
    hipLaunchKernelGGL(k_outer_totals<int64_t>, dim3(grid), dim3(kBlock), 0, s, fv.d_cols, fv.d_num_rows, fv.n_frags,
                       fv.n_cols, h.vcol, (int64_t)INT64_MIN, acc2);
  }
  hipLaunchKernelGGL(k_probe_finish, dim3(1), dim3(64), 0, s, h.fin, acc, acc2, out);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  st->spill_counter32 = (uint32_t*)spill_base;
  st->n_events_used = ev_i;
  return hipSuccess;
}

}  // namespace mq
