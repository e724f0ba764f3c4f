// kernels_sort.hip — ORDER BY <one target> [ASC | DESC] [NULLS FIRST | LAST] LIMIT k over a grouped
// result, on the device, so a dashboard-style "top n groups" never copies the 640 MB table to
// the host.
//
// Reference: ResultSet::sort (ResultSet.cpp:801-851) -> baselineSort -> baseline_sort
// (ResultSetSortImpl.cu: order-entry column, thrust::sort_by_key of entry indices) and
// TopKSort.cu (partition NULLs, radix sort, take the first n).  Here the k best entries are
// SELECTED instead of sorting everything:
//   1. k_topk_keys      one order-preserving uint64 per entry (empty entries sort last, NULLs
//                       first or last, DESC by complementing)
//   2. k_topk_hist x 6  MSB-first 11-bit radix histograms of the entries that still match the
//      k_topk_pick x 6  prefix found so far; each pick fixes 11 more bits of the k-th key
//   3. k_topk_collect   entries below the k-th key, plus as many equal to it as are still needed
//   4. k_topk_finish    one workgroup: bitonic sort of the <= 4096 candidates in LDS, then the
//                       rows are gathered in order
// Ties at the k-th position are broken arbitrarily, as in the reference.
#include <hip/hip_runtime.h>
#include <cstring>

#include "kernels.h"
#include "rowfunc.h"

namespace mq {

namespace {

constexpr int kBlock = 256;
constexpr int kRadixBits = 11;
constexpr int kBins = 1 << kRadixBits;
constexpr int kPasses = 6;  // 6 x 11 = 66 >= 64 bits
constexpr uint64_t kEmptySortKey = ~0ull;
constexpr uint64_t kNullLastKey = ~0ull - 1;

struct TopkState {  // device words shared by the passes
  unsigned long long prefix;      // bits of the k-th key fixed so far (left aligned)
  unsigned long long remaining;   // how many keys with this prefix are still wanted
  unsigned long long n_below;     // candidates written by collect
  unsigned long long n_ties;      // ties taken
  unsigned int hist[kBins];
};

// order-preserving pattern of a double; -0.0 and +0.0 are ONE value (ResultSetComparator compares doubles numerically:
// with distinct patterns a multi-key ORDER BY would stop at the first key where the reference goes on to the next)
MQ_D uint64_t ordered_f64(double d) {
  const uint64_t b = d == 0.0 ? 0ull : (uint64_t)dbl_bits(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

MQ_D uint64_t order_key_of(const DevPlan& p, const DevTarget& t, const int64_t* row, int idx_target_as_key,
                           int64_t null_pattern, bool fp_result, bool desc, bool nulls_first) {
  if (is_empty_row(p, row, idx_target_as_key)) return kEmptySortKey;
  uint64_t u;
  bool is_null = false;
  if (t.agg == MI355Q_PROJECT_KEY && t.slot < 0) {
    const int64_t v = row_key_component(row, p.key_width, t.key_idx);
    is_null = v == null_pattern;
    u = (uint64_t)v ^ 0x8000000000000000ull;
  } else if (p.slot_width == 4) {  // compact layouts: COUNT(*) or a projected key, 32-bit
    const int64_t v = ((const int32_t*)(row + p.key_quad))[t.slot];
    is_null = t.agg == MI355Q_PROJECT_KEY && v == null_pattern;
    u = (uint64_t)v ^ 0x8000000000000000ull;
  } else {
    const int64_t* s = row + p.key_quad + t.slot;
    if (t.arg_f32 && t.agg != MI355Q_AVG && t.agg != MI355Q_COUNT) {  // float bits in the low half
      is_null = t.skip_null && (int32_t)s[0] == (int32_t)null_pattern;
      u = ordered_f64((double)bits_flt((int32_t)s[0]));
    } else if (t.agg == MI355Q_AVG) {
      const int64_t cnt = s[1];
      is_null = cnt == 0;  // pair_to_double: count 0 -> NULL
      const double sum = t.arg_f32 ? (double)bits_flt((int32_t)s[0]) : t.arg_fp ? bits_dbl(s[0]) : (double)s[0];
      const double d = is_null ? 0.0 : sum / (double)cnt;
      u = ordered_f64(d);
    } else if (fp_result) {
      is_null = (t.skip_null) && s[0] == null_pattern;
      u = ordered_f64(bits_dbl(s[0]));
    } else {
      is_null = (t.skip_null || t.agg == MI355Q_PROJECT_KEY) && s[0] == null_pattern;
      u = (uint64_t)s[0] ^ 0x8000000000000000ull;
    }
  }
  if (is_null) return nulls_first ? 0ull : kNullLastKey;
  if (desc) u = ~u;
  // keep the reserved patterns free: 0 (NULLS FIRST), ~0 - 1 (NULLS LAST), ~0 (empty).  Known deviation: the two
  // smallest and the three largest patterns of the value range fold onto their neighbour (INT64_MIN / INT64_MIN + 1 and
  // INT64_MAX - 2 .. INT64_MAX for an integer target), i.e. they tie where the reference orders them; a nullable
  // integer target never holds INT64_MIN (it is the NULL sentinel) and no finite double reaches the fp patterns.
  if (u == 0ull) u = 1ull;
  if (u >= kNullLastKey) u = kNullLastKey - 1;
  return u;
}

__global__ __launch_bounds__(kBlock) void k_topk_keys(DevPlan p, int idx_target_as_key, int target,
                                                       int64_t null_pattern, int fp_result, int desc,
                                                       int nulls_first, const int64_t* __restrict__ buf,
                                                       uint64_t* __restrict__ keys, TopkState* st,
                                                       unsigned long long k) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < p.entry_count; e += stride) {
    keys[e] = order_key_of(p, p.targets[target], buf + e * p.row_quad, idx_target_as_key, null_pattern,
                           fp_result != 0, desc != 0, nulls_first != 0);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->prefix = 0;
    st->remaining = k;
    st->n_below = 0;
    st->n_ties = 0;
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < kBins; i += kBlock) st->hist[i] = 0;
}

// histogram of digit `pass` (MSB first) over the keys whose higher digits equal the prefix
__global__ __launch_bounds__(kBlock) void k_topk_hist(const uint64_t* __restrict__ keys, int64_t n, int pass,
                                                       TopkState* st) {
  __shared__ unsigned int s_hist[kBins];
  for (int i = threadIdx.x; i < kBins; i += kBlock) s_hist[i] = 0;
  __syncthreads();
  const int hi_bits = pass * kRadixBits;                 // bits already fixed
  const int shift = 64 - hi_bits - kRadixBits;           // may be negative in the last pass
  const uint64_t prefix = st->prefix;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
    const uint64_t u = keys[e];
    if (hi_bits && (u >> (64 - hi_bits)) != (prefix >> (64 - hi_bits))) continue;
    const unsigned int d = shift >= 0 ? (unsigned int)(u >> shift) & (kBins - 1)
                                      : (unsigned int)(u << -shift) & (kBins - 1);
    atomicAdd(&s_hist[d], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += kBlock)
    if (s_hist[i]) atomicAdd(&st->hist[i], s_hist[i]);
}

// one workgroup: find the digit where the running count reaches `remaining`
__global__ __launch_bounds__(kBlock) void k_topk_pick(int pass, TopkState* st) {
  __shared__ unsigned long long s_part[kBlock];
  const int per = kBins / kBlock;  // 8 consecutive bins per lane
  unsigned long long local = 0;
  for (int i = 0; i < per; ++i) local += st->hist[threadIdx.x * per + i];
  s_part[threadIdx.x] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long want = st->remaining;
    unsigned long long acc = 0;
    int lane = 0;
    while (lane < kBlock - 1 && acc + s_part[lane] < want) acc += s_part[lane++];
    int bin = lane * per;
    while (bin < kBins - 1 && acc + st->hist[bin] < want) acc += st->hist[bin++];
    const int hi_bits = pass * kRadixBits;
    const int shift = 64 - hi_bits - kRadixBits;
    const unsigned long long digit = (unsigned long long)bin;
    st->prefix |= shift >= 0 ? digit << shift : digit >> -shift;
    st->remaining = want - acc;  // still wanted among the keys of this digit
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kBins; i += kBlock) st->hist[i] = 0;
}

struct Cand {
  uint64_t key;
  uint64_t entry;
};

__global__ __launch_bounds__(kBlock) void k_topk_collect(const uint64_t* __restrict__ keys, int64_t n,
                                                          TopkState* st, Cand* __restrict__ cands,
                                                          unsigned long long k) {
  const uint64_t kth = st->prefix;            // all 64 bits are fixed now
  const unsigned long long ties_wanted = st->remaining;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
    const uint64_t u = keys[e];
    if (u == kEmptySortKey) continue;
    if (u < kth) {
      const unsigned long long i = atomicAdd(&st->n_below, 1ull);
      if (i < k) cands[i] = Cand{u, (uint64_t)e};
    } else if (u == kth) {
      const unsigned long long i = atomicAdd(&st->n_ties, 1ull);
      if (i < ties_wanted) cands[k - 1 - i] = Cand{u, (uint64_t)e};  // ties fill from the back
    }
  }
}

constexpr int kMaxTopk = 4096;

// one workgroup of 1024 lanes: bitonic sort of the candidates, then gather the rows
__global__ __launch_bounds__(1024) void k_topk_finish(DevPlan p, const int64_t* __restrict__ buf,
                                                       TopkState* st, const Cand* __restrict__ cands,
                                                       unsigned long long k, int64_t* __restrict__ out_rows,
                                                       int64_t* __restrict__ n_out) {
  __shared__ uint64_t s_key[kMaxTopk];
  __shared__ uint32_t s_ent[kMaxTopk];
  // number of real candidates: all below the k-th key + the ties taken (bounded by what exists)
  unsigned long long below = st->n_below;
  if (below > k) below = k;
  unsigned long long ties = st->n_ties < st->remaining ? st->n_ties : st->remaining;
  if (below + ties > k) ties = k - below;
  const unsigned int n = (unsigned int)(below + ties);
  for (unsigned int i = threadIdx.x; i < kMaxTopk; i += 1024) {
    Cand c{kEmptySortKey, 0};
    if (i < below) c = cands[i];
    else if (i < n) c = cands[k - 1 - (i - below)];
    s_key[i] = c.key;
    s_ent[i] = (uint32_t)c.entry;
  }
  __syncthreads();
  for (unsigned int size = 2; size <= kMaxTopk; size <<= 1) {
    for (unsigned int stride = size >> 1; stride > 0; stride >>= 1) {
      for (unsigned int i = threadIdx.x; i < kMaxTopk / 2; i += 1024) {
        const unsigned int lo = 2 * i - (i & (stride - 1));
        const unsigned int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const uint64_t a = s_key[lo], b = s_key[hi];
        const uint32_t ea = s_ent[lo], eb = s_ent[hi];
        // total order on (key, entry) so the result is deterministic for equal keys
        const bool gt = a > b || (a == b && ea > eb);
        if (gt == up) {
          s_key[lo] = b; s_key[hi] = a;
          s_ent[lo] = eb; s_ent[hi] = ea;
        }
      }
      __syncthreads();
    }
  }
  for (unsigned int i = threadIdx.x; i < n * (unsigned int)p.row_quad; i += 1024) {
    const unsigned int r = i / p.row_quad, j = i % p.row_quad;
    out_rows[(size_t)r * p.row_quad + j] = buf[(size_t)s_ent[r] * p.row_quad + j];
  }
  if (threadIdx.x == 0) *n_out = (int64_t)n;
}

// ---- full sort: ORDER BY several targets, any LIMIT / OFFSET (ResultSet::sort -> baselineSort /
// radixSortOnGpu / parallelTop, ResultSet.cpp:781-851; comparator semantics ResultSetComparator
// ::operator(), ResultSet.cpp:1310-1470: order entries compared in sequence, NULLs first or last per
// entry whatever the direction, DESC flips non-NULL values only, AVG compared as sum / count).
// A stable least-significant-key-first radix sort of the permutation of entries: for the LAST order
// entry first, the 64-bit order-preserving key of every entry in its current position is computed
// (k_sort_keys, the same key function as the top-k selection) and the (key, entry) pairs are sorted;
// after the first order entry's pass the permutation is ordered by all of them.  Empty entries carry
// the largest key in every pass and stay behind the live ones.
__global__ __launch_bounds__(kBlock) void k_sort_iota(uint32_t* __restrict__ perm, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) perm[e] = (uint32_t)e;
}

// ---- the pairs sort: a hand-written ONE-SWEEP least-significant-digit radix sort of (64-bit order key, 32-bit entry index)
// pairs (round 6; rounds 3-5 called rocprim::radix_sort_pairs here, the reference calls thrust::sort_by_key at the same
// spot, ResultSetSortImpl.cu:40-60, InPlaceSortImpl.cu:29-75).  Eight passes of 8-bit digits over two ping-pong buffers:
//   k_radix_hist   ONE pass over the keys: the 8 x 256 digit histograms (LDS atomics per workgroup, then global)
//   k_radix_scan   per digit position the exclusive prefix of its histogram = where each digit value's run starts; a
//                  position at which EVERY key holds the same digit is marked skipped (order keys of counts, small
//                  integers, dictionary ids leave their upper bytes constant: those passes move nothing) — the pass then
//                  returns at once and the buffers do not swap; which buffer holds the pairs is a device word, the host
//                  never looks
//   k_radix_pass   workgroups take tiles of 4 096 pairs off a ticket counter; a tile's keys are ranked stably — per wave by
//                  match-any ballots over the digit's bits (the lanes holding a digit value are a mask: rank = the digit's
//                  count so far + the lanes below), waves and items in order — the tile's digit counts go through a
//                  DECOUPLED LOOK-BACK (one lane per digit value, the same single-word relaxed agent-scope descriptors as
//                  the Projection family's tile_lookback: state in the top two bits), the pairs are ordered by digit in
//                  LDS and leave as runs (a digit value's pairs of one tile are contiguous in the output).
constexpr int kRsItems = 16;                    // pairs per lane and tile
constexpr int kRsTile = kBlock * kRsItems;      // 4 096
constexpr int kRsWaves = kBlock / 64;
constexpr uint32_t kRsShift = 30, kRsAggregate = 1u << kRsShift, kRsInclusive = 2u << kRsShift, kRsValue = (1u << kRsShift) - 1u;
struct RadixState {
  uint32_t hist[8][256];
  uint32_t base[8][256];   // exclusive prefix of hist[d]
  uint32_t skip[8];        // every key has the same digit d
  uint32_t parity[9];      // the buffer (0 / 1) the pairs are in BEFORE pass d; [8]: after the last
  uint32_t ticket[8];
  uint32_t cur;            // the buffer the pairs are in between sorts
  uint32_t pad_[6];
};

__global__ __launch_bounds__(kBlock) void k_sort_keys(DevPlan p, int idx_target_as_key, int target,
                                                       int64_t null_pattern, int fp_result, int desc,
                                                       int nulls_first, const int64_t* __restrict__ buf,
                                                       const uint32_t* perm0, const uint32_t* perm1, uint64_t* keys0, uint64_t* keys1,
                                                       const RadixState* st) {
  const uint32_t cur = st->cur;
  const uint32_t* __restrict__ perm = cur ? perm1 : perm0;
  uint64_t* __restrict__ keys = cur ? keys1 : keys0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < p.entry_count; i += stride) {
    keys[i] = order_key_of(p, p.targets[target], buf + (int64_t)perm[i] * p.row_quad, idx_target_as_key, null_pattern,
                           fp_result != 0, desc != 0, nulls_first != 0);
  }
}

__global__ __launch_bounds__(kBlock) void k_radix_hist(const uint64_t* keys0, const uint64_t* keys1, int64_t n, RadixState* st) {
  __shared__ uint32_t s_hist[8][256];
  const uint64_t* __restrict__ keys = st->cur ? keys1 : keys0;
  for (int i = threadIdx.x; i < 8 * 256; i += kBlock) (&s_hist[0][0])[i] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const uint64_t k = keys[i];
#pragma unroll
    for (int d = 0; d < 8; ++d) atomicAdd(&s_hist[d][(k >> (8 * d)) & 255u], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 256; i += kBlock) {
    const uint32_t v = (&s_hist[0][0])[i];
    if (v) atomicAdd(&st->hist[0][0] + i, v);
  }
}

// one workgroup: prefixes, skipped positions, buffer parities; the histograms and tickets are cleared for the next sort
__global__ __launch_bounds__(kBlock) void k_radix_scan(RadixState* st, int64_t n) {
  __shared__ uint32_t s_scan[256];
  __shared__ uint32_t s_skip[8];
  const int t = threadIdx.x;
  if (t < 8) s_skip[t] = 0;
  __syncthreads();
  for (int d = 0; d < 8; ++d) {
    const uint32_t v = st->hist[d][t];
    if (n > 0 && (int64_t)v == n) s_skip[d] = 1;
    s_scan[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const uint32_t o = t >= off ? s_scan[t - off] : 0u;
      __syncthreads();
      s_scan[t] += o;
      __syncthreads();
    }
    st->base[d][t] = s_scan[t] - v;
    st->hist[d][t] = 0;
    __syncthreads();
  }
  if (t == 0) {
    uint32_t par = st->cur;
    for (int d = 0; d < 8; ++d) {
      st->parity[d] = par;
      st->skip[d] = s_skip[d];
      st->ticket[d] = 0;
      if (!s_skip[d]) par ^= 1u;
    }
    st->parity[8] = par;
    st->cur = par;
  }
}

__global__ __launch_bounds__(kBlock) void k_radix_pass(int d, uint64_t* keys0, uint64_t* keys1, uint32_t* vals0, uint32_t* vals1, int64_t n,
                                                        RadixState* st, uint32_t* desc) {
  if (st->skip[d]) return;  // (uniform for the whole grid)
  __shared__ uint64_t s_keys[kRsTile];
  __shared__ uint32_t s_vals[kRsTile];
  __shared__ uint32_t s_wave_cnt[kRsWaves][256];  // per wave: pairs per digit value; then the wave's start inside the digit's run
  __shared__ uint32_t s_local[256];               // the digit value's first position in the tile (by digit order)
  __shared__ uint32_t s_goff[256];                // global position of the digit value's first pair of this tile
  __shared__ uint32_t s_scan[256];
  __shared__ uint32_t s_tile;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t par = st->parity[d];
  const uint64_t* __restrict__ kin = par ? keys1 : keys0;
  uint64_t* __restrict__ kout = par ? keys0 : keys1;
  const uint32_t* __restrict__ vin = par ? vals1 : vals0;
  uint32_t* __restrict__ vout = par ? vals0 : vals1;
  const int64_t n_tiles = (n + kRsTile - 1) / kRsTile;
  const int shift = 8 * d;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (;;) {
    if (t == 0) s_tile = atomicAdd(&st->ticket[d], 1u);
    __syncthreads();
    const int64_t tile = (int64_t)s_tile;
    if (tile >= n_tiles) break;
    const int64_t base = tile * kRsTile;
    // ---- load: item i of lane l of wave w is pair (w * kRsItems + i) * 64 + l of the tile (order = wave, item, lane)
    uint64_t key[kRsItems];
    uint32_t val[kRsItems], rank[kRsItems];
#pragma unroll
    for (int i = 0; i < kRsItems; ++i) {
      const int64_t g = base + (int64_t)(wave * kRsItems + i) * 64 + lane;
      key[i] = g < n ? kin[g] : ~0ull;
      val[i] = g < n ? vin[g] : 0u;
    }
    for (int i = t; i < kRsWaves * 256; i += kBlock) (&s_wave_cnt[0][0])[i] = 0;
    __syncthreads();
    // ---- stable ranks inside the wave
#pragma unroll
    for (int i = 0; i < kRsItems; ++i) {
      const int64_t g = base + (int64_t)(wave * kRsItems + i) * 64 + lane;
      const bool valid = g < n;
      const uint32_t dg = (uint32_t)(key[i] >> shift) & 255u;
      unsigned long long m = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot((dg >> b) & 1u);
        m &= ((dg >> b) & 1u) ? bal : ~bal;
      }
      uint32_t pre = 0;
      if (valid) pre = s_wave_cnt[wave][dg];
      rank[i] = pre + (uint32_t)__popcll(m & lt_mask);
      __builtin_amdgcn_wave_barrier();
      if (valid && (m & lt_mask) == 0) s_wave_cnt[wave][dg] = pre + (uint32_t)__popcll(m);  // (the lowest lane of the group)
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- digit value t: its count in the tile, the waves' starts inside its run, its place in the tile
    uint32_t cnt = 0;
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const uint32_t c = s_wave_cnt[w][t];
      s_wave_cnt[w][t] = cnt;
      cnt += c;
    }
    s_scan[t] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const uint32_t o = t >= off ? s_scan[t - off] : 0u;
      __syncthreads();
      s_scan[t] += o;
      __syncthreads();
    }
    s_local[t] = s_scan[t] - cnt;
    // ---- decoupled look-back for digit value t: pairs with this digit in the tiles before this one
    {
      uint32_t* my = desc + (size_t)tile * 256 + t;
      __hip_atomic_store(my, (tile == 0 ? kRsInclusive : kRsAggregate) | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t excl = 0;
      for (int64_t pt = tile - 1; pt >= 0; --pt) {
        const uint32_t* pd = desc + (size_t)pt * 256 + t;
        uint32_t v = __hip_atomic_load(pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((v >> kRsShift) == 0) {
          __builtin_amdgcn_s_sleep(4);
          v = __hip_atomic_load(pd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        excl += v & kRsValue;
        if ((v >> kRsShift) == 2) break;
      }
      if (tile > 0) __hip_atomic_store(my, kRsInclusive | (excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_goff[t] = st->base[d][t] + excl;
    }
    __syncthreads();
    // ---- the pairs in digit order in LDS ...
#pragma unroll
    for (int i = 0; i < kRsItems; ++i) {
      const int64_t g = base + (int64_t)(wave * kRsItems + i) * 64 + lane;
      if (g < n) {
        const uint32_t dg = (uint32_t)(key[i] >> shift) & 255u;
        const uint32_t lp = s_local[dg] + s_wave_cnt[wave][dg] + rank[i];
        s_keys[lp] = key[i];
        s_vals[lp] = val[i];
      }
    }
    __syncthreads();
    // ... and out: position j of the tile goes to the digit value's run
    const int64_t left = n - base;
    const int n_here = left < kRsTile ? (int)left : kRsTile;
#pragma unroll
    for (int i = 0; i < kRsItems; ++i) {
      const int j = i * kBlock + t;
      if (j < n_here) {
        const uint64_t k = s_keys[j];
        const uint32_t dg = (uint32_t)(k >> shift) & 255u;
        const uint32_t dst = s_goff[dg] + ((uint32_t)j - s_local[dg]);
        kout[dst] = k;
        vout[dst] = s_vals[j];
      }
    }
    __syncthreads();
  }
}

// rows [offset, offset + n_out) of the sorted permutation, as whole rows; n_live = non-empty entries
__global__ __launch_bounds__(kBlock) void k_sort_gather(DevPlan p, const int64_t* __restrict__ buf,
                                                         const uint32_t* perm0, const uint32_t* perm1, const RadixState* st, int64_t offset,
                                                         int64_t limit, const int64_t* __restrict__ n_live,
                                                         int64_t* __restrict__ out_rows, int64_t* __restrict__ n_out) {
  const uint32_t* __restrict__ perm = st->cur ? perm1 : perm0;
  int64_t n = *n_live - offset;
  if (n < 0) n = 0;
  if (limit > 0 && n > limit) n = limit;
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = n;
  const int64_t quads = n * p.row_quad;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < quads; i += stride) {
    const int64_t r = i / p.row_quad, j = i % p.row_quad;
    out_rows[i] = buf[(int64_t)perm[offset + r] * p.row_quad + j];
  }
}

inline int grid_for(int64_t work_items, int max_blocks = 2048) {
  int64_t b = (work_items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

}  // namespace

int64_t topk_scratch_bytes(int64_t entry_count) {
  return entry_count * 8 + (int64_t)sizeof(TopkState) + (int64_t)kMaxTopk * (int64_t)sizeof(Cand) + 512;
}
int topk_max_k() { return kMaxTopk; }

hipError_t launch_topk(const DevPlan& p, int idx_target_as_key, int target, int64_t null_pattern,
                       bool fp_result, bool desc, bool nulls_first, const int64_t* buf, int64_t k,
                       void* scratch, int64_t* out_rows, int64_t* d_n_out, hipStream_t s) {
  if (k < 1 || k > kMaxTopk) return hipErrorInvalidValue;
  uint64_t* keys = (uint64_t*)scratch;
  char* tail = (char*)scratch + (((size_t)p.entry_count * 8 + 255) & ~(size_t)255);
  TopkState* st = (TopkState*)tail;
  Cand* cands = (Cand*)(tail + ((sizeof(TopkState) + 255) & ~(size_t)255));
  const int grid = grid_for(p.entry_count);
  hipLaunchKernelGGL(k_topk_keys, dim3(grid), dim3(kBlock), 0, s, p, idx_target_as_key, target, null_pattern,
                     (int)fp_result, (int)desc, (int)nulls_first, buf, keys, st, (unsigned long long)k);
  for (int pass = 0; pass < kPasses; ++pass) {
    hipLaunchKernelGGL(k_topk_hist, dim3(grid), dim3(kBlock), 0, s, keys, p.entry_count, pass, st);
    hipLaunchKernelGGL(k_topk_pick, dim3(1), dim3(kBlock), 0, s, pass, st);
  }
  hipLaunchKernelGGL(k_topk_collect, dim3(grid), dim3(kBlock), 0, s, keys, p.entry_count, st, cands,
                     (unsigned long long)k);
  hipLaunchKernelGGL(k_topk_finish, dim3(1), dim3(1024), 0, s, p, buf, st, cands, (unsigned long long)k,
                     out_rows, d_n_out);
  return hipGetLastError();
}

// scratch layout of the full sort: keys[2][E] | perm[2][E] | n_live, n_out | RadixState | tile descriptors [tiles][256]
static size_t sort_state_bytes() { return (sizeof(RadixState) + 255) & ~(size_t)255; }
static size_t sort_desc_bytes(int64_t entry_count) {
  const size_t tiles = (size_t)((entry_count > 0 ? entry_count : 1) + kRsTile - 1) / kRsTile;
  return (tiles * 256 * 4 + 255) & ~(size_t)255;
}
int64_t sort_scratch_bytes(int64_t entry_count) {
  const size_t e = (size_t)(entry_count > 0 ? entry_count : 1);
  return (int64_t)(2 * ((e * 8 + 255) & ~(size_t)255) + 2 * ((e * 4 + 255) & ~(size_t)255) + 256 + sort_state_bytes() + sort_desc_bytes(entry_count));
}

hipError_t launch_sort(const DevPlan& p, int idx_target_as_key, const SortOrderEntry* order, int n_order,
                       const int64_t* buf, int64_t offset, int64_t limit, void* scratch, int64_t* out_rows,
                       int64_t* d_n_out, hipStream_t s) {
  // (the look-back descriptors keep a count in 30 bits)
  if (n_order < 1 || offset < 0 || limit < 0 || p.entry_count >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
  const size_t e = (size_t)(p.entry_count > 0 ? p.entry_count : 1);
  const size_t kb = (e * 8 + 255) & ~(size_t)255, pb = (e * 4 + 255) & ~(size_t)255;
  char* base = (char*)scratch;
  uint64_t* keys[2] = {(uint64_t*)base, (uint64_t*)(base + kb)};
  uint32_t* perm[2] = {(uint32_t*)(base + 2 * kb), (uint32_t*)(base + 2 * kb + pb)};
  int64_t* d_live = (int64_t*)(base + 2 * kb + 2 * pb);
  RadixState* st = (RadixState*)(base + 2 * kb + 2 * pb + 256);
  uint32_t* desc = (uint32_t*)(base + 2 * kb + 2 * pb + 256 + sort_state_bytes());
  const size_t desc_bytes = sort_desc_bytes(p.entry_count);
  const int grid = grid_for(p.entry_count);
  const int64_t n_tiles = (p.entry_count + kRsTile - 1) / kRsTile;
  const int pass_grid = (int)(n_tiles < 1 ? 1 : n_tiles > 1024 ? 1024 : n_tiles);
  hipError_t e0 = hipMemsetAsync(st, 0, sizeof(RadixState), s);
  if (e0 != hipSuccess) return e0;
  hipLaunchKernelGGL(k_sort_iota, dim3(grid), dim3(kBlock), 0, s, perm[0], p.entry_count);
  for (int o = n_order - 1; o >= 0; --o) {
    const SortOrderEntry& oe = order[o];
    hipLaunchKernelGGL(k_sort_keys, dim3(grid), dim3(kBlock), 0, s, p, idx_target_as_key, oe.target, oe.null_pattern,
                       (int)oe.fp_result, (int)oe.desc, (int)oe.nulls_first, buf, perm[0], perm[1], keys[0], keys[1], st);
    hipLaunchKernelGGL(k_radix_hist, dim3(grid_for(p.entry_count, 1024)), dim3(kBlock), 0, s, keys[0], keys[1], p.entry_count, st);
    hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(kBlock), 0, s, st, p.entry_count);
    for (int d = 0; d < 8; ++d) {
      e0 = hipMemsetAsync(desc, 0, desc_bytes, s);
      if (e0 != hipSuccess) return e0;
      hipLaunchKernelGGL(k_radix_pass, dim3(pass_grid), dim3(kBlock), 0, s, d, keys[0], keys[1], perm[0], perm[1], p.entry_count, st, desc);
    }
  }
  hipError_t e3 = hipMemsetAsync(d_live, 0, 16, s);
  if (e3 != hipSuccess) return e3;
  e3 = launch_count_nonempty(p, idx_target_as_key, buf, (unsigned long long*)d_live, s);
  if (e3 != hipSuccess) return e3;
  int64_t want = limit > 0 ? limit : p.entry_count;
  hipLaunchKernelGGL(k_sort_gather, dim3(grid_for(want * p.row_quad)), dim3(kBlock), 0, s, p, buf, perm[0], perm[1], st, offset, limit,
                     d_live, out_rows, d_n_out);
  return hipGetLastError();
}

}  // namespace mq