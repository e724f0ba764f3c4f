// kernels_lds.hip — GROUP BY with FEW groups: the whole table lives in every workgroup's LDS.
//
// The shapes of the reference's own synthetic benchmark that have up to a few thousand groups
// (Benchmarks/synthetic_benchmark/queries: PerfectHashSingleCol/PHS001-003, PerfectHashMultiCol/PHM001-002,
// BaselineHash/BH001-003 — GROUP BY cast(x AS DOUBLE) — and the MultiStep queries over x1k / x100 x y10): one to
// three key columns, one to three VALUE columns, every aggregate kind over columns declared nullable.  Before this
// kernel they ran in the row kernel: one global atomic per row and slot on a handful of addresses (0.2 - 2 % of the
// roofline, and a baseline table with ten groups serialised the whole device on ten cache lines).
//
//   layout   per workgroup (1024 lanes, one per CU) K REPLICAS of the table in LDS; lane l updates replica l % K,
//            so a table with 10 groups does not serialise 64 lanes on one LDS word (K = as many as fit, <= 64).
//            One replica = one array per accumulator: rows (u32) per entry; per value column, as needed by the
//            targets over it: non-NULL count (u32), sum (i64 / f64), min, max.  Perfect-hash layouts index the
//            arrays by the entry index  sum_i (key_i - min_i) * mul_i  (NULL keys translated; get_group_value_fast,
//            GroupByRuntime.cpp:208-223; perfect_key_hash, GroupByAndAggregate.cpp:1546-1598); baseline layouts
//            keep an open-addressing key array per replica and use the LDS slot as the index.
//   stream   the scan of k_scan_agg: every load of a step is issued before the first value is looked at.
//   flush    replicas are folded into replica 0 inside the workgroup, then every live entry is merged into the
//            output table with the reduce rule (reduce_target: the same code the reduce kernel runs, so NULL-aware
//            slots, AVG pairs and projected keys behave exactly as in ResultSetStorage::reduce); baseline entries
//            go through the reference's insert-or-find (get_group_value, GroupByRuntime.cpp:25-48).
//   give up  a baseline table with more groups than a replica holds sets a flag; the caller re-runs the step with
//            the partitioned family.
//   windows  a table that does not fit one LDS is cut into T <= 8 WINDOWS (perfect hash: T ranges of the entry index;
//            baseline: T classes of a key hash): workgroup b owns window b % T and row stripe b / T, reads every row of
//            its stripe and keeps the rows of its window.  The columns are read T times (by workgroups that run side by
//            side, so mostly out of the Infinity Cache) instead of once — against the partitioned family's exchange of
//            16 B written + 16 B read per row, and against one global atomic per row and slot on a few thousand
//            addresses: the reference benchmark's 10 K-group shapes (PHS004, PHM003, BH004, BH007) took 1.8 - 3.3 ns per
//            row that way (profiles/r03_refbench_128m_call4.jsonl).
#include <cstring>
#include <type_traits>

#include "fast_common.h"
#include "lds_args.h"

namespace mq {

using namespace fast;

namespace {

constexpr int kLdsBlock = 1024;
constexpr size_t kLdsBudget = 152 * 1024;

struct RawQ {
  v4i32 lo, hi;
};
// w: bytes per value — 8, 4, or 1 (a TINYINT / BOOLEAN / row-mask filter column: the quad's four bytes in lo.x)
MQ_D void load_rawq_w(const int8_t* base, int64_t quad, int w, RawQ& r) {
  if (w == 1) {
    r.lo.x = (int)__builtin_nontemporal_load((const MQ_GLOBAL uint32_t*)base + quad);
  } else if (w == 8) {
    r.lo = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2);
    r.hi = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2 + 1);
  } else {
    r.lo = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad);
  }
}
MQ_D void load_rawq(const int8_t* base, int64_t quad, bool w8, RawQ& r) {
  if (w8) {
    r.lo = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2);
    r.hi = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2 + 1);
  } else {
    r.lo = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad);
  }
}
MQ_D int64_t rawq_i64(const RawQ& r, int i) {
  const v4i32& h = i < 2 ? r.lo : r.hi;
  const int j = (i & 1) * 2;
  return (int64_t)(((uint64_t)(uint32_t)(j ? h.w : h.y) << 32) | (uint64_t)(uint32_t)(j ? h.z : h.x));
}
MQ_D int32_t rawq_i32(const RawQ& r, int i) { return i == 0 ? r.lo.x : i == 1 ? r.lo.y : i == 2 ? r.lo.z : r.lo.w; }
MQ_D int32_t rawq_i8(const RawQ& r, int i) { return (int32_t)(int8_t)((uint32_t)r.lo.x >> (8 * i)); }
MQ_D int64_t rawq_int(const RawQ& r, int type, int i) {
  return type == MI355Q_INT32 ? (int64_t)rawq_i32(r, i) : type == MI355Q_INT8 ? (int64_t)rawq_i8(r, i) : rawq_i64(r, i);
}
MQ_D int flt_width(int type) { return type == MI355Q_INT32 ? 4 : type == MI355Q_INT8 ? 1 : 8; }

MQ_D void lds_min_f64(int64_t* s, double v) {
  int64_t old = *(volatile int64_t*)s;
  for (;;) {
    const double o = bits_dbl(old);
    if (!(v < o)) return;
    const int64_t seen = (int64_t)atomicCAS((unsigned long long*)s, (unsigned long long)old, (unsigned long long)dbl_bits(v));
    if (seen == old) return;
    old = seen;
  }
}
MQ_D void lds_max_f64(int64_t* s, double v) {
  int64_t old = *(volatile int64_t*)s;
  for (;;) {
    const double o = bits_dbl(old);
    if (!(o < v)) return;
    const int64_t seen = (int64_t)atomicCAS((unsigned long long*)s, (unsigned long long)old, (unsigned long long)dbl_bits(v));
    if (seen == old) return;
    old = seen;
  }
}

// one value of value column c folded into entry e of replica `rep`
MQ_D void lds_update(char* rep, const LdsVal& v, uint32_t e, int64_t bits) {
  if (v.type == MI355Q_DOUBLE) {
    const double x = bits_dbl(bits);
    if (v.nullable && x == kNullDouble) return;
    if (v.off_cnt >= 0) atomicAdd((uint32_t*)(rep + v.off_cnt) + e, 1u);
    if (v.off_sum >= 0) atomicAdd((double*)(rep + v.off_sum) + e, x);
    if (v.off_min >= 0) lds_min_f64((int64_t*)(rep + v.off_min) + e, x);
    if (v.off_max >= 0) lds_max_f64((int64_t*)(rep + v.off_max) + e, x);
  } else {
    if (v.nullable && bits == (v.type == MI355Q_INT32 ? (int64_t)INT32_MIN : INT64_MIN)) return;
    if (v.off_cnt >= 0) atomicAdd((uint32_t*)(rep + v.off_cnt) + e, 1u);
    if (v.off_sum >= 0) atomicAdd((unsigned long long*)(rep + v.off_sum) + e, (unsigned long long)bits);
    if (v.off_min >= 0) atomicMin((long long*)(rep + v.off_min) + e, (long long)bits);
    if (v.off_max >= 0) atomicMax((long long*)(rep + v.off_max) + e, (long long)bits);
  }
}

// find-or-insert in one replica's key array (linear probing from a multiplicative hash); kNoSlot when full
constexpr uint32_t kNoSlot = 0xffffffffu;
// One 32-bit mix of the 8 key bytes serves both decisions: its TOP bits pick the window (mul-hi by T), its LOW bits the home
// slot inside the window's key array.  Both halves of the key feed it (a DOUBLE key that holds a small integer has its
// entropy in the HIGH word and 40 - 50 trailing zero bits: the first version hashed `key * odd >> 40` and sent every group
// of cast(x AS DOUBLE) to slot 0 — BH002 / BH003 ran 16 - 50 x slower than the same shape on an integer key,
// profiles/r03_refbench_1b_call4.jsonl; strided BIGINT keys have it in the low word), four 32-bit multiplies in all: the
// 64-bit multiplies of the second version cost 12 quarter-rate instructions per row VISIT, and a row is visited once
// per window (BH004, 8 windows: 12.8 ms per 128 M rows against 2.0 ms for the perfect-hash twin,
// profiles/r03_lds_retry_chain_kernel_trace_call7.csv).
MQ_D uint32_t lds_key_mix(int64_t key) {
  const uint32_t lo = (uint32_t)(uint64_t)key, hi = (uint32_t)((uint64_t)key >> 32);
  uint32_t h = (lo * 0x9E3779B1u) ^ (hi * 0x85EBCA77u);
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return h;
}
MQ_D uint32_t lds_mix_window(uint32_t T, uint32_t h) { return __umulhi(h, T); }
MQ_D uint32_t lds_key_slot(int64_t* keys, uint32_t H, int64_t key, uint32_t h) {
  uint32_t s = h & (H - 1);  // H is a power of two
  for (uint32_t trips = 0; trips < H; ++trips) {
    int64_t k = *(volatile int64_t*)&keys[s];
    if (k == kEmptyKey64) {
      // (a lost exchange hands back the key that took the slot: it is compared below and the walk moves on — the
      // re-look used to count as a trip without advancing, so a contended replica could report "full" early)
      k = (int64_t)atomicCAS((unsigned long long*)&keys[s], (unsigned long long)kEmptyKey64, (unsigned long long)key);
      if (k == kEmptyKey64) return s;
    }
    if (k == key) return s;
    s = (s + 1) & (H - 1);
  }
  return kNoSlot;
}

// Windows: which window and row stripe a workgroup takes.  The T workgroups that share a stripe read the same rows, so
// they belong on ONE XCD (one L2): consecutive block ids are dealt to the 8 XCDs round-robin, so with `xcd_aware`
// (grid = 8 x T x stripes-per-XCD) workgroup b sits on XCD b % 8 and is the (b / 8)-th there — stripe (b / 8) / T of that
// XCD, window (b / 8) % T.  Without it (grids too small to fill eight XCDs: tune_cus, the host simulation) stripe-mates
// are simply consecutive blocks.  Measured before: BH007 (8 windows, 12 B/row) 29.7 ms per 1 B rows = 3.2 TB/s of
// reads that mostly missed the L2 (profiles/r04_refbench_1b_call3.jsonl).
struct WinMap {
  uint32_t win, stripe, n_stripes;
};
MQ_D WinMap lds_window_map(uint32_t T, uint32_t xcd_aware) {
  WinMap m;
  if (T <= 1) {
    m.win = 0;
    m.stripe = blockIdx.x;
    m.n_stripes = gridDim.x;
  } else if (xcd_aware) {
    const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const uint32_t per_xcd = (gridDim.x >> 3) / T;  // stripes per XCD
    m.win = j % T;
    m.stripe = (j / T) * 8u + xcd;
    m.n_stripes = per_xcd * 8u;
  } else {
    m.win = blockIdx.x % T;
    m.stripe = blockIdx.x / T;
    m.n_stripes = gridDim.x / T;
  }
  return m;
}

// NF / NK / NV: quals, key columns and value columns this member holds registers for; UQ quads per column per step.
// Every index into the kernel-argument structs is a compile-time constant after unrolling: a dynamically indexed
// by-value kernel argument is lowered to a private-memory (scratch) copy, and every access to it would then be a
// scratch load in front of the column loads it feeds.
template <int NF, int NK, int NV, int UQ>
__global__ __launch_bounds__(kLdsBlock) void k_groupby_lds(const int8_t* const* __restrict__ cols,
                                                            const int64_t* __restrict__ num_rows, int n_frags, int n_cols,
                                                            LdsArgs a, DevPlan p, int64_t* __restrict__ out,
                                                            int32_t* __restrict__ d_err) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ BoolFilter s_bf;  // the compiled filter: atoms + truth table (a.bf_on)
  const int t = threadIdx.x;
  if (a.bf_on) bf_load(a.bf, &s_bf, t, kLdsBlock);  // (visible after the barrier behind the replicas' initialisation)
  const uint32_t K = 1u << a.copies_lg;
  const uint32_t ne = a.entries;
  // windows: this workgroup's window, its row stripe and the number of stripes (T = 1: the whole table, every workgroup a stripe)
  const uint32_t T = a.windows;
  const WinMap wm = lds_window_map(T, a.xcd_aware);
  const uint32_t win = wm.win, stripe = wm.stripe, n_stripes = wm.n_stripes;
  const uint32_t e_lo = a.baseline ? 0u : win * ne;  // perfect hash: first entry index of the window
  // ---- initialise every replica: counters 0, sums 0, min / max identities, keys empty
  for (uint32_t r = 0; r < K; ++r) {
    char* rep = smem + (size_t)r * a.copy_bytes;
    for (uint32_t e = t; e < ne; e += kLdsBlock) {
      ((uint32_t*)(rep + a.off_rows))[e] = 0;
      if (a.baseline) ((int64_t*)(rep + a.off_keys))[e] = kEmptyKey64;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        if (c >= a.n_vals) break;
        const LdsVal& v = a.v[c];
        const bool fp = v.type == MI355Q_DOUBLE;
        if (v.off_cnt >= 0) ((uint32_t*)(rep + v.off_cnt))[e] = 0;
        if (v.off_sum >= 0) ((int64_t*)(rep + v.off_sum))[e] = 0;  // 0 and +0.0 share the pattern
        // (doubles: +inf / -inf, not +-DBL_MAX — a group whose only values are infinite must come out infinite where the
        // reference's _skip_val aggregate takes the first value as it is; found by the division vectors, x / DBL_MIN = inf)
        if (v.off_min >= 0) ((int64_t*)(rep + v.off_min))[e] = fp ? 0x7ff0000000000000ll : INT64_MAX;
        if (v.off_max >= 0) ((int64_t*)(rep + v.off_max))[e] = fp ? (int64_t)0xfff0000000000000ull : INT64_MIN;
      }
    }
  }
  if (t == 0) *(uint32_t*)(smem + ((size_t)a.copy_bytes << a.copies_lg)) = 0u;
  __syncthreads();
  char* const my_rep = smem + (size_t)((uint32_t)t & (K - 1)) * a.copy_bytes;
  volatile uint32_t* const s_full = (volatile uint32_t*)(smem + ((size_t)a.copy_bytes << a.copies_lg));  // behind the replicas
  bool bad = false, full = false;

  // one row whose quals passed: kv = the key columns' values (DOUBLE keys as their bit pattern), vv = the value columns'
  auto one_row = [&](const int64_t (&kv)[NK], const int64_t (&vv)[NV]) {
    uint32_t e;
    if (a.baseline) {
      // a FLOAT key is the bit pattern of the double it widens to (castToTypeIn(group_key, 64), IRCodegen.cpp:1505-1507)
      const int64_t key = a.key_type[0] == MI355Q_FLOAT ? dbl_bits((double)bits_flt((int32_t)kv[0])) : kv[0];
      const uint32_t h = lds_key_mix(key);
      if (T > 1 && lds_mix_window(T, h) != win) return;
      // the attempt is lost as soon as ONE lane of the workgroup finds a replica full: everybody else learns it from an
      // LDS word instead of walking a full key array itself (one walk per lane and divergence group made a lost
      // attempt cost 10 - 12 ms whatever the input size, profiles/r03_lds_retry_chain_kernel_trace_call7.csv)
      if (full || *s_full) {
        full = true;
        return;
      }
      e = lds_key_slot((int64_t*)(my_rep + a.off_keys), ne, key, h);
      if (e == kNoSlot) {
        // published at once (the other workgroups stop when they see the flag), and ONCE per workgroup: 1 024 lanes x 256
        // workgroups exchanging the same global word are 262 K serialised atomics, ~ 10 ns each — most of what a lost
        // attempt used to cost (profiles/r03_lds_retry_chain_kernel_trace_call8.csv)
        if (atomicExch((uint32_t*)s_full, 1u) == 0u) atomicExch(d_err + 1, 1);
        full = true;
        return;
      }
    } else {
      int64_t idx = 0;
      bool in_range = true;
#pragma unroll
      for (int g = 0; g < NK; ++g) {
        if (g >= a.n_keys) break;
        int64_t k = kv[g];
        if (a.key_translate[g] && k == (a.key_type[g] == MI355Q_INT32 ? (int64_t)INT32_MIN : INT64_MIN)) k = a.key_null_key[g];
        const int64_t d = k - a.key_min[g];
        in_range = in_range && d >= 0 && d < a.key_card[g];
        idx += d * a.key_mul[g];
      }
      if (!in_range || (uint64_t)idx >= (uint64_t)p.entry_count) {
        bad = true;
        return;
      }
      e = (uint32_t)idx - e_lo;
      if (e >= ne) return;  // another window's row
    }
    atomicAdd((uint32_t*)(my_rep + a.off_rows) + e, 1u);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      if (c >= a.n_vals) break;
      lds_update(my_rep, a.v[c], e, vv[c]);
    }
  };

  const int64_t tile_q = (int64_t)kLdsBlock * UQ;
  const int64_t gtid = (int64_t)stripe * kLdsBlock + t;
  const int64_t gsize = (int64_t)n_stripes * kLdsBlock;
  for (int f = 0; f < n_frags; ++f) {
    if (*(volatile int32_t*)(d_err + 1)) break;  // some workgroup's replica overflowed: the step is re-run anyway
    const int8_t* const* fc = cols + (size_t)f * n_cols;
    const int64_t n = num_rows[f];
    const int64_t nq = n >> 2;
    const int64_t n_tiles = nq / tile_q;
    // the fragment's chunk of every stream, fetched once
    const int8_t *fb[NF > 0 ? NF : 1], *kb[NK], *vb[NV];
#pragma unroll
    for (int k = 0; k < NF; ++k) fb[k] = k < a.n_flt ? fc[a.flt[k].col] : nullptr;
#pragma unroll
    for (int g = 0; g < NK; ++g) kb[g] = g < a.n_keys ? fc[a.key_col[g]] : nullptr;
#pragma unroll
    for (int c = 0; c < NV; ++c) vb[c] = c < a.n_vals ? fc[a.v[c].col] : nullptr;
    auto do_step = [&](int64_t q0, int n_quads, int64_t stride) {
      RawQ fr[NF > 0 ? NF : 1][UQ], kr[NK][UQ], vr[NV][UQ];
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        if (u >= n_quads) break;
        const int64_t quad = q0 + (int64_t)u * stride;
#pragma unroll
        for (int k = 0; k < NF; ++k)
          if (k < a.n_flt) load_rawq_w(fb[k], quad, flt_width(a.flt_type[k]), fr[k][u]);
#pragma unroll
        for (int g = 0; g < NK; ++g)
          if (g < a.n_keys) load_rawq(kb[g], quad, a.key_type[g] != MI355Q_INT32 && a.key_type[g] != MI355Q_FLOAT, kr[g][u]);
#pragma unroll
        for (int c = 0; c < NV; ++c)
          if (c < a.n_vals) load_rawq(vb[c], quad, a.v[c].type != MI355Q_INT32, vr[c][u]);
      }
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        if (u >= n_quads) break;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bool pass = true;
          if (NF > 0 && a.bf_on) {  // atoms on the filter columns' values + one bit of the truth table
            int64_t fval[NF > 0 ? NF : 1];
#pragma unroll
            for (int k = 0; k < NF; ++k) fval[k] = k < a.n_flt ? rawq_int(fr[k][u], a.flt_type[k], i) : 0;
            pass = bf_row_passes<(NF > 0 ? NF : 1)>(s_bf, fval);
          } else {
#pragma unroll
            for (int k = 0; k < NF; ++k) {
              if (k >= a.n_flt) break;
              pass = pass && (a.flt_type[k] == MI355Q_INT32 ? filter_pass<int32_t>(a.flt[k], rawq_i32(fr[k][u], i))
                              : a.flt_type[k] == MI355Q_INT8 ? filter_pass<int32_t>(a.flt[k], rawq_i8(fr[k][u], i))
                                                             : filter_pass<int64_t>(a.flt[k], rawq_i64(fr[k][u], i)));
            }
          }
          if (!pass) continue;
          int64_t kv[NK], vv[NV];
#pragma unroll
          for (int g = 0; g < NK; ++g)
            kv[g] = g < a.n_keys ? rawq_int(kr[g][u], (a.key_type[g] == MI355Q_INT32 || a.key_type[g] == MI355Q_FLOAT) ? MI355Q_INT32 : MI355Q_INT64, i) : 0;
#pragma unroll
          for (int c = 0; c < NV; ++c) vv[c] = c < a.n_vals ? rawq_int(vr[c][u], a.v[c].type == MI355Q_INT32 ? MI355Q_INT32 : MI355Q_INT64, i) : 0;
          one_row(kv, vv);
        }
      }
    };
    uint32_t tiles_done = 0;
    for (int64_t tl = (stripe + (int64_t)f * 7) % n_stripes; tl < n_tiles; tl += n_stripes) {
      // (a baseline attempt that cannot hold the groups is abandoned by everybody soon after the first overflow)
      if (a.baseline && (full || ((tiles_done++ & 3u) == 3u && *(volatile int32_t*)(d_err + 1)))) break;
      do_step(tl * tile_q + t, UQ, kLdsBlock);
    }
    for (int64_t q = n_tiles * tile_q + gtid; q < nq; q += gsize) do_step(q, 1, 0);
    const int64_t tail = (nq << 2) + gtid;
    if (tail < n) {
      bool pass = true;
      if (NF > 0 && a.bf_on) {
        int64_t fval[NF > 0 ? NF : 1];
#pragma unroll
        for (int k = 0; k < NF; ++k)
          fval[k] = k >= a.n_flt ? 0 : a.flt_type[k] == MI355Q_INT32 ? (int64_t)load_one<int32_t>(fb[k], tail) : load_one<int64_t>(fb[k], tail);
        pass = bf_row_passes<(NF > 0 ? NF : 1)>(s_bf, fval);
      } else {
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          if (k >= a.n_flt) break;
          pass = pass && (a.flt_type[k] == MI355Q_INT32 ? filter_pass<int32_t>(a.flt[k], load_one<int32_t>(fb[k], tail))
                          : a.flt_type[k] == MI355Q_INT8 ? filter_pass<int32_t>(a.flt[k], (int32_t)load_one<int8_t>(fb[k], tail))
                                                         : filter_pass<int64_t>(a.flt[k], load_one<int64_t>(fb[k], tail)));
        }
      }
      if (pass) {
        int64_t kv[NK], vv[NV];
#pragma unroll
        for (int g = 0; g < NK; ++g)
          kv[g] = g >= a.n_keys ? 0 : (a.key_type[g] == MI355Q_INT32 || a.key_type[g] == MI355Q_FLOAT) ? (int64_t)load_one<int32_t>(kb[g], tail) : load_one<int64_t>(kb[g], tail);
#pragma unroll
        for (int c = 0; c < NV; ++c)
          vv[c] = c >= a.n_vals ? 0 : a.v[c].type == MI355Q_INT32 ? (int64_t)load_one<int32_t>(vb[c], tail) : load_one<int64_t>(vb[c], tail);
        one_row(kv, vv);
      }
    }
  }
  if (bad) atomicCAS(d_err, 0, MI355Q_ERR_OUT_OF_SLOTS);
  // a lost attempt (this workgroup's, or another's) is neither folded nor flushed: the caller takes another member and
  // initialises the table again
  if (t == 0 && a.baseline && *(volatile int32_t*)(d_err + 1)) *s_full = 1u;
  __syncthreads();
  if (*s_full) return;

  // ---- fold replicas 1 .. K-1 into replica 0
  char* const rep0 = smem;
  for (uint32_t r = 1; r < K; ++r) {
    char* rep = smem + (size_t)r * a.copy_bytes;
    for (uint32_t e = t; e < ne; e += kLdsBlock) {
      const uint32_t rows = ((uint32_t*)(rep + a.off_rows))[e];
      if (!rows) continue;
      uint32_t e0 = e;
      if (a.baseline) {
        const int64_t fk = ((int64_t*)(rep + a.off_keys))[e];
        e0 = lds_key_slot((int64_t*)(rep0 + a.off_keys), ne, fk, lds_key_mix(fk));
        if (e0 == kNoSlot) {  // the replicas together hold more groups than one does
          if (atomicExch((uint32_t*)s_full, 1u) == 0u) atomicExch(d_err + 1, 1);
          continue;
        }
      }
      atomicAdd((uint32_t*)(rep0 + a.off_rows) + e0, rows);
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        if (c >= a.n_vals) break;
        const LdsVal& v = a.v[c];
        const bool fp = v.type == MI355Q_DOUBLE;
        if (v.off_cnt >= 0) atomicAdd((uint32_t*)(rep0 + v.off_cnt) + e0, ((uint32_t*)(rep + v.off_cnt))[e]);
        if (v.off_sum >= 0) {
          if (fp) atomicAdd((double*)(rep0 + v.off_sum) + e0, ((double*)(rep + v.off_sum))[e]);
          else atomicAdd((unsigned long long*)(rep0 + v.off_sum) + e0, ((unsigned long long*)(rep + v.off_sum))[e]);
        }
        if (v.off_min >= 0) {
          if (fp) lds_min_f64((int64_t*)(rep0 + v.off_min) + e0, ((double*)(rep + v.off_min))[e]);
          else atomicMin((long long*)(rep0 + v.off_min) + e0, ((long long*)(rep + v.off_min))[e]);
        }
        if (v.off_max >= 0) {
          if (fp) lds_max_f64((int64_t*)(rep0 + v.off_max) + e0, ((double*)(rep + v.off_max))[e]);
          else atomicMax((long long*)(rep0 + v.off_max) + e0, ((long long*)(rep + v.off_max))[e]);
        }
      }
    }
    __syncthreads();  // (baseline: inserts into replica 0 of one round must be visible to the next)
    if (*s_full) return;
  }

  // ---- merge every live entry of replica 0 into the output table with the reduce rule
  for (uint32_t e = t; e < ne; e += kLdsBlock) {
    const uint32_t rows = ((const uint32_t*)(rep0 + a.off_rows))[e];
    if (!rows) continue;
    int64_t key0 = 0, key1 = 0, key2 = 0;  // the group columns' values as decoded (what a projection shows)
    int64_t* slots;
    if (a.baseline) {
      key0 = ((const int64_t*)(rep0 + a.off_keys))[e];
      slots = baseline_find_or_insert(out, (uint32_t)p.entry_count, p.row_quad, p.key_width, key0);
      if (!slots) {
        atomicCAS(d_err, 0, -1);  // out of group slots: the caller grows the table and retries
        continue;
      }
    } else {
      int64_t tk0 = 0, tk1 = 0, tk2 = 0;
      if ((uint64_t)e_lo + e >= (uint64_t)p.entry_count) continue;  // (padding of the last window)
      uint32_t rem = e_lo + e;
#pragma unroll
      for (int g = NK - 1; g >= 0; --g) {  // entry index -> key components (mul_g ascending with g)
        if (g >= a.n_keys) continue;
        const int64_t d = (int64_t)(rem / (uint32_t)a.key_mul[g]);
        rem -= (uint32_t)(d * a.key_mul[g]);
        const int64_t tk = d + a.key_min[g];
        const int64_t orig = (a.key_translate[g] && tk == a.key_null_key[g])
                                 ? (a.key_type[g] == MI355Q_INT32 ? (int64_t)INT32_MIN : INT64_MIN) : tk;
        if (g == 0) { tk0 = tk; key0 = orig; } else if (g == 1) { tk1 = tk; key1 = orig; } else { tk2 = tk; key2 = orig; }
      }
      int64_t* row = out + (size_t)(e_lo + e) * p.row_quad;
      if (!p.keyless) {
        if (MQ_LOAD64(row) == kEmptyKey64) {
          if (a.n_keys > 2) MQ_STORE64(row + 2, tk2);
          if (a.n_keys > 1) MQ_STORE64(row + 1, tk1);
          MQ_STORE64(row, tk0);
        }
        slots = row + a.n_keys;
      } else {
        slots = row;
      }
    }
    // the entry's partial row, slot by slot (static slot indices: no private-memory array), merged with the reduce rule
#pragma unroll
    for (int i = 0; i < MI355Q_MAX_TARGETS; ++i) {
      if (i >= p.n_targets) break;
      const DevTarget& tg = p.targets[i];
      if (tg.slot < 0) continue;
      int64_t v0 = p.init_vals[tg.slot], v1 = tg.agg == MI355Q_AVG ? p.init_vals[tg.slot + 1] : 0;
      if (tg.agg == MI355Q_PROJECT_KEY) {
        v0 = tg.key_idx == 0 ? key0 : tg.key_idx == 1 ? key1 : key2;
      } else {
        const int c = a.target_v[i];
        if (c < 0) {
          v0 = (int64_t)rows;
        } else {
          const LdsVal v = c == 0 ? a.v[0] : (NV > 1 && c == 1) ? a.v[NV > 1 ? 1 : 0] : a.v[NV > 2 ? 2 : 0];
          // rows with a value: the non-NULL counter where one is kept (nullable column), else every row
          const uint32_t cnt = v.off_cnt >= 0 ? ((const uint32_t*)(rep0 + v.off_cnt))[e] : rows;
          switch (tg.agg) {
            case MI355Q_COUNT: v0 = (int64_t)cnt; break;
            case MI355Q_AVG:
              v1 = (int64_t)cnt;
              [[fallthrough]];
            case MI355Q_SUM:
              if (cnt) v0 = ((const int64_t*)(rep0 + v.off_sum))[e];
              break;
            case MI355Q_MIN:
              if (cnt) v0 = ((const int64_t*)(rep0 + v.off_min))[e];
              break;
            default:
              if (cnt) v0 = ((const int64_t*)(rep0 + v.off_max))[e];
          }
        }
      }
      // reduce_target reads that_slots[t.slot (+ 1)]: hand it a two-quad window positioned at the target's slot
      int64_t win[2] = {v0, v1};
      DevTarget lt = tg;
      lt.slot = 0;
      reduce_target<true>(lt, p.init_vals + tg.slot, slots + tg.slot, win);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TYPED members (round 4).  k_groupby_lds above decides every role at run time — key / value types, which accumulators a
// column keeps, perfect hash or baseline — and its row loop is 5 941 static instructions per 8 rows, 3 899 of them uniform
// branches (profiles/r03_isa_k_groupby_lds_static.txt): ~130 dynamic instructions per row visit, 0.10 - 0.26 of the
// roofline on the reference benchmark's PerfectHashSingleCol / MultiCol / BaselineHash / MultiStep shapes.  Those shapes
// (Benchmarks/synthetic_benchmark/create_table.py:116-137: every column INT, the strided ones BIGINT; keys cast to DOUBLE
// or FLOAT for the baseline groups) are what the reference's shared-memory group-by serves (NativeCodegen.cpp:2750-2845),
// so they get members with the roles compiled in:
//   KK   0 = perfect hash over NK INT32 key columns (32-bit index arithmetic: the entry index fits 16 bits per component)
//        1 = baseline, one 8-byte key (BIGINT, or DOUBLE as its bit pattern)
//        2 = baseline, FLOAT key (the bit pattern of the double it widens to, IRCodegen.cpp:1505-1507)
//        3 = baseline, INT32 key (sign-extended)
//   NV   INT32 value columns (nullable or not: one uniform flag per column); per column a non-NULL count (u32), a sum
//        (i64) and — MM — min / max kept as INT32 (`ds_min_i32` instead of a 64-bit LDS atomic, 8 instead of 16 bytes)
//   UQ   quads per lane, column and step; the NEXT tile's loads are issued before this tile's rows are looked at
// A row of another window is rejected on its key alone, before any value is touched.  No quals (the benchmark's grouped
// queries have none); anything else — quals, wider values, DOUBLE arguments — stays with the generic member.
MQ_D int32_t v4_get(const v4i32& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// FM = 1: the step has a filter — up to kTypedFlt plain INT32 columns under range quals (make_range_filter; two quals on
// one column are one range), or under a filter compiled at plan time (boolfilter.h: atoms + truth table, a.bf_on)
constexpr int kTypedFlt = 3;
template <int KK, int NK, int NV, bool MM, int UQ, int FM>
__global__ __launch_bounds__(kLdsBlock) void k_groupby_lds_typed(const int8_t* const* __restrict__ cols,
                                                                  const int64_t* __restrict__ num_rows, int n_frags, int n_cols,
                                                                  LdsArgs a, DevPlan p, int64_t* __restrict__ out,
                                                                  int32_t* __restrict__ d_err) {
  static_assert(KK == 0 || NK == 1, "baseline members take one key column");
  constexpr bool kBase = KK != 0;
  constexpr int KW = KK == 1 ? 2 : 1;  // 16-byte loads per key quad
  constexpr int TF = FM ? kTypedFlt : 1;
  constexpr int NVA = NV ? NV : 1;  // (NV = 0: the COUNT(*)-only members — arrays of one unused element)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ BoolFilter s_bf;  // (FM, a.bf_on: visible after the barrier behind the replicas' initialisation)
  const int t = threadIdx.x;
  if constexpr (FM != 0) {
    if (a.bf_on) bf_load(a.bf, &s_bf, t, kLdsBlock);
  }
  const uint32_t K = 1u << a.copies_lg;
  const uint32_t E = a.entries;
  const uint32_t T = a.windows;
  const WinMap wm = lds_window_map(T, a.xcd_aware);
  const uint32_t win = wm.win, stripe = wm.stripe, n_stripes = wm.n_stripes;
  const uint32_t e_lo = kBase ? 0u : win * E;
  // ---- initialise every replica
  for (uint32_t r = 0; r < K; ++r) {
    char* rep = smem + (size_t)r * a.copy_bytes;
    for (uint32_t e = t; e < E; e += kLdsBlock) {
      if (kBase) ((int64_t*)(rep + a.t_off_keys))[e] = kEmptyKey64;
      ((uint32_t*)(rep + a.t_off_rows))[e] = 0;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        ((int64_t*)(rep + a.t_off_sum))[c * E + e] = 0;
        ((uint32_t*)(rep + a.t_off_cnt))[c * E + e] = 0;
        if (MM) {
          ((int32_t*)(rep + a.t_off_min))[c * E + e] = INT32_MAX;
          ((int32_t*)(rep + a.t_off_max))[c * E + e] = INT32_MIN;
        }
      }
    }
  }
  volatile uint32_t* const s_full = (volatile uint32_t*)(smem + ((size_t)a.copy_bytes << a.copies_lg));  // behind the replicas
  if (t == 0) *s_full = 0u;
  __syncthreads();
  char* const my_rep = smem + (size_t)((uint32_t)t & (K - 1)) * a.copy_bytes;
  int64_t* const my_keys = (int64_t*)(my_rep + a.t_off_keys);
  unsigned long long* const my_sum = (unsigned long long*)(my_rep + a.t_off_sum);
  uint32_t* const my_rows = (uint32_t*)(my_rep + a.t_off_rows);
  uint32_t* const my_cnt = (uint32_t*)(my_rep + a.t_off_cnt);
  int32_t* const my_min = (int32_t*)(my_rep + a.t_off_min);
  int32_t* const my_max = (int32_t*)(my_rep + a.t_off_max);
  bool bad = false, full = false;
  // uniform per-column constants (static indices after unrolling: scalar registers, never a scratch copy of `a`)
  uint32_t kmin[NK], kcard[NK], kmul[NK], knull[NK];
  bool ktr[NK], vnull[NVA];
#pragma unroll
  for (int g = 0; g < NK; ++g) {
    kmin[g] = (uint32_t)(int32_t)a.key_min[g];
    kcard[g] = (uint32_t)a.key_card[g];
    kmul[g] = (uint32_t)a.key_mul[g];
    knull[g] = (uint32_t)a.key_null_key[g];
    ktr[g] = a.key_translate[g] != 0;
  }
#pragma unroll
  for (int c = 0; c < NV; ++c) vnull[c] = a.v[c].nullable != 0;
  const uint32_t n_entries = (uint32_t)p.entry_count;
  // the range filters in the columns' own 32 bits (a bound beyond them is the type's; an empty range is lo > hi)
  const int nfl = FM ? a.n_flt : 0;
  int32_t flo[TF], fhi[TF], fnv[TF];
  bool fneg[TF], fnul[TF], f8[TF];  // f8: a 1-byte filter column (TINYINT / BOOLEAN / a compiled filter's row mask)
#pragma unroll
  for (int c = 0; c < TF; ++c) {
    f8[c] = c < nfl && a.flt_type[c] == MI355Q_INT8;
    flo[c] = 1;
    fhi[c] = 0;
    fnv[c] = INT32_MIN;
    fneg[c] = fnul[c] = false;
    if (c < nfl) {
      const int64_t lo = a.flt[c].lo, hi = a.flt[c].hi;
      if (lo <= hi && lo <= (int64_t)INT32_MAX && hi >= (int64_t)INT32_MIN) {
        flo[c] = lo < (int64_t)INT32_MIN ? INT32_MIN : (int32_t)lo;
        fhi[c] = hi > (int64_t)INT32_MAX ? INT32_MAX : (int32_t)hi;
      }
      fneg[c] = a.flt[c].negate != 0;
      fnul[c] = a.flt[c].nullable != 0;
      fnv[c] = (int32_t)a.flt[c].null_val;
    }
  }
  int32_t bf_err = 0;  // the error a program atom of the compiled filter raised for one of this lane's rows
  // a compiled filter's LEAN program atoms (boolfilter.h PairAtom: `a / b > 3`, `x + y > 100`, `a < b` over INT32 columns):
  // at most kLdsFusedProgs, read once into scalar registers; a row evaluates each with ONE 32-bit operation (pair_eval)
  int n_fprog = 0;
  uint32_t f_atoms_of = 0;
  PairAtom fpa[kLdsFusedProgs];
  int fca[kLdsFusedProgs], fcb[kLdsFusedProgs];
  bool fraise[kLdsFusedProgs];
#pragma unroll
  for (int k = 0; k < kLdsFusedProgs; ++k) {
    fpa[k] = PairAtom{};
    fca[k] = fcb[k] = 0;
    fraise[k] = false;
  }
  if constexpr (FM == 2) {  // (FM = 2: the member of filters with lean program atoms; FM = 1 carries none of this)
    if (a.bf_on) {
      n_fprog = MQ_WAVE_UNIFORM(s_bf.n_progs);
      if (n_fprog) {
#pragma unroll
        for (int c = 0; c < TF; ++c) f_atoms_of |= (uint32_t)MQ_WAVE_UNIFORM(c < s_bf.n_cols ? s_bf.atoms_of_col[c] : 0) << (8 * c);
#pragma unroll
        for (int k = 0; k < kLdsFusedProgs; ++k) {
          const PairAtom& src = s_bf.pair[k];
          fpa[k].op = MQ_WAVE_UNIFORM(src.op);
          fpa[k].ln = MQ_WAVE_UNIFORM(src.ln);
          fpa[k].rn = MQ_WAVE_UNIFORM(src.rn);
          fpa[k].b_is_lit = MQ_WAVE_UNIFORM(src.b_is_lit);
          fpa[k].b_lit = MQ_WAVE_UNIFORM(src.b_lit);
          fpa[k].lo = MQ_WAVE_UNIFORM(src.lo);
          fpa[k].hi = MQ_WAVE_UNIFORM(src.hi);
          fpa[k].negate = MQ_WAVE_UNIFORM(src.negate);
          fpa[k].op2 = MQ_WAVE_UNIFORM(src.op2);
          fpa[k].lit2 = MQ_WAVE_UNIFORM(src.lit2);
          fca[k] = MQ_WAVE_UNIFORM(s_bf.prog_op[k][0]);
          fcb[k] = MQ_WAVE_UNIFORM(s_bf.prog_op[k][1]);
          fraise[k] = MQ_WAVE_UNIFORM(s_bf.prog[k].can_raise) != 0;
        }
      }
    }
  }
  auto row_passes = [&](const int32_t (&fv)[TF]) -> bool {
    if constexpr (FM == 0) return true;
    if (FM == 2 && a.bf_on && n_fprog) {
      uint32_t idx = 0, mul = 1, ep = 0;
      int ai = 0;
#pragma unroll
      for (int c = 0; c < TF; ++c) {
        const int cnt = (int)((f_atoms_of >> (8 * c)) & 255u);
        for (int k = 0; k < cnt; ++k) {
          idx += bf_atom_state(s_bf.atom[ai], (int64_t)fv[c]) * mul;
          mul *= 3u;
          ++ai;
        }
      }
#pragma unroll
      for (int k = 0; k < kLdsFusedProgs; ++k) {
        if (k >= n_fprog) break;
        int32_t av = fv[0], bv = fv[0];
#pragma unroll
        for (int c = 1; c < TF; ++c) {
          if (fca[k] == c) av = fv[c];
          if (fcb[k] == c) bv = fv[c];
        }
        int32_t e = 0;
        idx += pair_eval(fpa[k], av, bv, e) * mul;
        ep |= ex_err_enc(e) << (2 * k);
        mul *= fraise[k] ? 4u : 3u;
      }
      bool bit = (s_bf.table[idx >> 5] >> (idx & 31u)) & 1u;
      if (ep) {  // rare: an atom of this row is in its ERROR state — is it the row's outcome?
        const uint32_t nib = (s_bf.etable[idx >> 3] >> ((idx & 7u) * 4u)) & 15u;
        if (nib) {
          if (!bf_err) bf_err = ex_err_dec((ep >> (2u * (nib - 1u))) & 3u);
          bit = false;
        }
      }
      return bit;
    }
    if (a.bf_on) {
      int64_t vals[TF];
#pragma unroll
      for (int c = 0; c < TF; ++c) vals[c] = (int64_t)fv[c];
      return bf_row_passes<TF>(s_bf, vals);
    }
    bool pass = true;
#pragma unroll
    for (int c = 0; c < TF; ++c) {
      if (c >= nfl) break;
      bool in = fv[c] >= flo[c] && fv[c] <= fhi[c];
      if (fneg[c]) in = !in;
      if (fnul[c] && fv[c] == fnv[c]) in = false;
      pass = pass && in;
    }
    return pass;
  };

  // entry e takes one row's values
  auto update = [&](uint32_t e, const int32_t (&vv)[NVA]) {
    atomicAdd(my_rows + e, 1u);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int32_t v = vv[c];
      if (vnull[c] && v == INT32_MIN) continue;
      atomicAdd(my_cnt + c * E + e, 1u);
      atomicAdd(my_sum + c * E + e, (unsigned long long)(int64_t)v);
      if (MM) {
        atomicMin(my_min + c * E + e, v);
        atomicMax(my_max + c * E + e, v);
      }
    }
  };
  // baseline: the 8-byte key a row groups by, from the low (or only) word klo and the high word khi of the column value
  auto key_of = [&](int32_t klo, int32_t khi) -> int64_t {
    if constexpr (KK == 1) return (int64_t)(((uint64_t)(uint32_t)khi << 32) | (uint64_t)(uint32_t)klo);
    else if constexpr (KK == 2) return dbl_bits((double)bits_flt(klo));
    else return (int64_t)klo;
  };
  // baseline, the first probe missed (a new group, or a collision): insert-or-find.  kNoSlot = the attempt is lost — as
  // soon as ONE lane of the workgroup finds a replica full everybody else learns it from an LDS word instead of walking
  // a full key array itself, and the global flag is raised once per workgroup (see k_groupby_lds)
  auto slow_locate = [&](int64_t key, uint32_t h) -> uint32_t {
    if (full || *s_full) {
      full = true;
      return kNoSlot;
    }
    const uint32_t e = lds_key_slot(my_keys, E, key, h);
    if (e == kNoSlot) {
      if (atomicExch((uint32_t*)s_full, 1u) == 0u) atomicExch(d_err + 1, 1);
      full = true;
    }
    return e;
  };
  // perfect hash: the entry of a row (32-bit throughout: make_lds_args admits |key_min| < 2^30 and cardinalities <= 65 536,
  // so `k - min` taken modulo 2^32 is either the true difference or >= 2^30 (out of range), and the entry index stays
  // below 2^32); kNoSlot = another window's row, or (bad) a key outside its declared range
  auto perfect_entry = [&](const int32_t (&klo)[NK]) -> uint32_t {
    uint32_t idx = 0;
    bool in_range = true;
#pragma unroll
    for (int g = 0; g < NK; ++g) {
      uint32_t ku = (uint32_t)klo[g];
      if (ktr[g] && klo[g] == INT32_MIN) ku = knull[g];
      const uint32_t d = ku - kmin[g];
      in_range = in_range && d < kcard[g];
      idx += d * kmul[g];
    }
    if (!in_range || idx >= n_entries) {
      bad = true;
      return kNoSlot;
    }
    const uint32_t e = idx - e_lo;
    return e < E ? e : kNoSlot;
  };
  // one row on its own (quad remainders, tail rows)
  auto one_row = [&](const int32_t (&klo)[NK], int32_t khi, const int32_t (&vv)[NVA]) {
    uint32_t e;
    if constexpr (kBase) {
      const int64_t key = key_of(klo[0], khi);
      const uint32_t h = lds_key_mix(key);
      if (T > 1 && lds_mix_window(T, h) != win) return;  // another window's row: nothing else of it is looked at
      e = slow_locate(key, h);
    } else {
      e = perfect_entry(klo);
    }
    if (e != kNoSlot) update(e, vv);
  };

  struct Tile {
    v4i32 k[NK][UQ][KW];
    v4i32 v[NVA][UQ];
    v4i32 f[TF][UQ];
  };
  const int64_t tile_q = (int64_t)kLdsBlock * UQ;
  const int64_t gtid = (int64_t)stripe * kLdsBlock + t;
  const int64_t gsize = (int64_t)n_stripes * kLdsBlock;
  for (int f = 0; f < n_frags; ++f) {
    if (kBase && *(volatile int32_t*)(d_err + 1)) break;  // some workgroup's replica overflowed: the step is re-run anyway
    const int8_t* const* fc = cols + (size_t)f * n_cols;
    const int64_t n = num_rows[f];
    const int64_t nq = n >> 2;
    const int64_t n_tiles = nq / tile_q;
    const int8_t *kb[NK], *vb[NVA], *fb[TF];
#pragma unroll
    for (int g = 0; g < NK; ++g) kb[g] = fc[a.key_col[g]];
#pragma unroll
    for (int c = 0; c < NV; ++c) vb[c] = fc[a.v[c].col];
#pragma unroll
    for (int c = 0; c < TF; ++c) fb[c] = c < nfl ? fc[a.flt[c].col] : nullptr;
    auto load_tile = [&](Tile& tl, int64_t q0, int64_t stride) {
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        const int64_t quad = q0 + (int64_t)u * stride;
        if constexpr (FM != 0) {
#pragma unroll
          for (int c = 0; c < TF; ++c)
            if (c < nfl) {
              if (f8[c]) tl.f[c][u].x = (int)__builtin_nontemporal_load((const MQ_GLOBAL uint32_t*)fb[c] + quad);  // four 1-byte rows
              else tl.f[c][u] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)fb[c] + quad);
            }
        }
#pragma unroll
        for (int g = 0; g < NK; ++g) {
#pragma unroll
          for (int w = 0; w < KW; ++w) tl.k[g][u][w] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)kb[g] + quad * KW + w);
        }
#pragma unroll
        for (int c = 0; c < NV; ++c) tl.v[c][u] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)vb[c] + quad);
      }
    };
    auto do_tile = [&](const Tile& tl, int n_quads) {
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        if (u >= n_quads) break;
        if constexpr (kBase) {
          // the four rows of a quad together: keys and hashes first, then the four first-probe reads of the key array
          // in one go (one LDS round trip instead of four dependent ones — a wave's step is a chain of latencies, and
          // with eight windows it is walked eight times per row), then the updates; a miss takes the insert-or-find walk
          int64_t key[4], k0[4];
          uint32_t hh[4];
          bool acc[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (KK == 1) key[i] = key_of(v4_get(tl.k[0][u][i >> 1], (i & 1) * 2), v4_get(tl.k[0][u][i >> 1], (i & 1) * 2 + 1));
            else key[i] = key_of(v4_get(tl.k[0][u][0], i), 0);
            hh[i] = lds_key_mix(key[i]);
            acc[i] = !(T > 1 && lds_mix_window(T, hh[i]) != win);
            if constexpr (FM != 0) {
              int32_t fv[TF];
#pragma unroll
              for (int c = 0; c < TF; ++c) fv[c] = c >= nfl ? 0 : f8[c] ? (int32_t)(int8_t)((uint32_t)tl.f[c][u].x >> (8 * i)) : v4_get(tl.f[c][u], i);
              acc[i] = acc[i] && row_passes(fv);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) k0[i] = acc[i] ? *(volatile int64_t*)&my_keys[hh[i] & (E - 1)] : kEmptyKey64;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (!acc[i]) continue;
            uint32_t e = hh[i] & (E - 1);
            if (k0[i] != key[i]) {
              e = slow_locate(key[i], hh[i]);
              if (e == kNoSlot) continue;
            }
            int32_t vv[NVA];
#pragma unroll
            for (int c = 0; c < NV; ++c) vv[c] = v4_get(tl.v[c][u], i);
            update(e, vv);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int32_t klo[NK], vv[NVA];
            if constexpr (FM != 0) {
              int32_t fv[TF];
#pragma unroll
              for (int c = 0; c < TF; ++c) fv[c] = c >= nfl ? 0 : f8[c] ? (int32_t)(int8_t)((uint32_t)tl.f[c][u].x >> (8 * i)) : v4_get(tl.f[c][u], i);
              if (!row_passes(fv)) continue;
            }
#pragma unroll
            for (int g = 0; g < NK; ++g) klo[g] = v4_get(tl.k[g][u][0], i);
            const uint32_t e = perfect_entry(klo);
            if (e == kNoSlot) continue;
#pragma unroll
            for (int c = 0; c < NV; ++c) vv[c] = v4_get(tl.v[c][u], i);
            update(e, vv);
          }
        }
      }
    };
    Tile cur, nxt;
    uint32_t tiles_done = 0;
    int64_t tl = (stripe + (int64_t)f * 7) % n_stripes;
    if (tl < n_tiles) load_tile(nxt, tl * tile_q + t, kLdsBlock);
    for (; tl < n_tiles; tl += n_stripes) {
      cur = nxt;
      const int64_t nx = tl + n_stripes;
      if (nx < n_tiles) load_tile(nxt, nx * tile_q + t, kLdsBlock);
      if (kBase && (full || ((tiles_done++ & 3u) == 3u && *(volatile int32_t*)(d_err + 1)))) break;
      do_tile(cur, UQ);
    }
    for (int64_t q = n_tiles * tile_q + gtid; q < nq; q += gsize) {
      load_tile(cur, q, 0);  // (UQ copies of one quad; only the first is used)
      do_tile(cur, 1);
    }
    const int64_t tail = (nq << 2) + gtid;
    if (tail < n) {
      int32_t klo[NK], vv[NVA];
      int32_t khi = 0;
#pragma unroll
      for (int g = 0; g < NK; ++g) {
        if constexpr (KK == 1) {
          const int64_t k8 = load_one<int64_t>(kb[g], tail);
          klo[g] = (int32_t)(uint32_t)(uint64_t)k8;
          khi = (int32_t)(uint32_t)((uint64_t)k8 >> 32);
        } else {
          klo[g] = load_one<int32_t>(kb[g], tail);
        }
      }
#pragma unroll
      for (int c = 0; c < NV; ++c) vv[c] = load_one<int32_t>(vb[c], tail);
      int32_t fv[TF];
#pragma unroll
      for (int c = 0; c < TF; ++c) fv[c] = c >= nfl ? 0 : f8[c] ? (int32_t)load_one<int8_t>(fb[c], tail) : load_one<int32_t>(fb[c], tail);
      if (row_passes(fv)) one_row(klo, khi, vv);
    }
  }
  if (bad) atomicCAS(d_err, 0, MI355Q_ERR_OUT_OF_SLOTS);
  if (FM == 2 && bf_err) atomicCAS(d_err, 0, bf_err);  // a program atom of the compiled filter raised (error 7 / error 1)
  if (t == 0 && kBase && *(volatile int32_t*)(d_err + 1)) *s_full = 1u;
  __syncthreads();
  if (*s_full) return;  // a lost attempt is neither folded nor flushed

  // ---- fold replicas 1 .. K-1 into replica 0
  char* const rep0 = smem;
  int64_t* const keys0 = (int64_t*)(rep0 + a.t_off_keys);
  unsigned long long* const sum0 = (unsigned long long*)(rep0 + a.t_off_sum);
  uint32_t* const rows0 = (uint32_t*)(rep0 + a.t_off_rows);
  uint32_t* const cnt0 = (uint32_t*)(rep0 + a.t_off_cnt);
  int32_t* const min0 = (int32_t*)(rep0 + a.t_off_min);
  int32_t* const max0 = (int32_t*)(rep0 + a.t_off_max);
  for (uint32_t r = 1; r < K; ++r) {
    const char* rep = smem + (size_t)r * a.copy_bytes;
    for (uint32_t e = t; e < E; e += kLdsBlock) {
      const uint32_t rows = ((const uint32_t*)(rep + a.t_off_rows))[e];
      if (!rows) continue;
      uint32_t e0 = e;
      if (kBase) {
        const int64_t fk = ((const int64_t*)(rep + a.t_off_keys))[e];
        e0 = lds_key_slot(keys0, E, fk, lds_key_mix(fk));
        if (e0 == kNoSlot) {  // the replicas together hold more groups than one does
          if (atomicExch((uint32_t*)s_full, 1u) == 0u) atomicExch(d_err + 1, 1);
          continue;
        }
      }
      atomicAdd(rows0 + e0, rows);
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const uint32_t cn = ((const uint32_t*)(rep + a.t_off_cnt))[c * E + e];
        if (!cn) continue;
        atomicAdd(cnt0 + c * E + e0, cn);
        atomicAdd(sum0 + c * E + e0, ((const unsigned long long*)(rep + a.t_off_sum))[c * E + e]);
        if (MM) {
          atomicMin(min0 + c * E + e0, ((const int32_t*)(rep + a.t_off_min))[c * E + e]);
          atomicMax(max0 + c * E + e0, ((const int32_t*)(rep + a.t_off_max))[c * E + e]);
        }
      }
    }
    __syncthreads();  // (baseline: inserts into replica 0 of one round must be visible to the next)
    if (*s_full) return;
  }

  // ---- merge every live entry of replica 0 into the output table with the reduce rule (as k_groupby_lds does)
  for (uint32_t e = t; e < E; e += kLdsBlock) {
    const uint32_t rows = rows0[e];
    if (!rows) continue;
    int64_t key0 = 0, key1 = 0, key2 = 0;
    int64_t* slots;
    if (kBase) {
      key0 = keys0[e];
      slots = baseline_find_or_insert(out, n_entries, p.row_quad, p.key_width, key0);
      if (!slots) {
        atomicCAS(d_err, 0, -1);  // out of group slots: the caller grows the table and retries
        continue;
      }
    } else {
      int64_t tk0 = 0, tk1 = 0, tk2 = 0;
      if (e_lo + e >= n_entries) continue;  // (padding of the last window)
      uint32_t rem = e_lo + e;
#pragma unroll
      for (int g = NK - 1; g >= 0; --g) {  // entry index -> key components (mul_g ascending with g)
        const uint32_t d = rem / kmul[g];
        rem -= d * kmul[g];
        const int64_t tk = (int64_t)d + a.key_min[g];
        const int64_t orig = (ktr[g] && tk == a.key_null_key[g]) ? (int64_t)INT32_MIN : tk;
        if (g == 0) { tk0 = tk; key0 = orig; } else if (g == 1) { tk1 = tk; key1 = orig; } else { tk2 = tk; key2 = orig; }
      }
      int64_t* row = out + (size_t)(e_lo + e) * p.row_quad;
      if (!p.keyless) {
        if (MQ_LOAD64(row) == kEmptyKey64) {
          if (NK > 2) MQ_STORE64(row + 2, tk2);
          if (NK > 1) MQ_STORE64(row + 1, tk1);
          MQ_STORE64(row, tk0);
        }
        slots = row + NK;
      } else {
        slots = row;
      }
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
        const int c = a.target_v[i];
        if (c < 0) {
          v0 = (int64_t)rows;
        } else {
          const uint32_t cnt = cnt0[(uint32_t)c * E + e];
          switch (tg.agg) {
            case MI355Q_COUNT: v0 = (int64_t)cnt; break;
            case MI355Q_AVG:
              v1 = (int64_t)cnt;
              [[fallthrough]];
            case MI355Q_SUM:
              if (cnt) v0 = (int64_t)sum0[(uint32_t)c * E + e];
              break;
            case MI355Q_MIN:
              if (cnt && MM) v0 = (int64_t)min0[(uint32_t)c * E + e];
              break;
            default:
              if (cnt && MM) v0 = (int64_t)max0[(uint32_t)c * E + e];
          }
        }
      }
      int64_t win2[2] = {v0, v1};
      DevTarget lt = tg;
      lt.slot = 0;
      reduce_target<true>(lt, p.init_vals + tg.slot, slots + tg.slot, win2);
    }
  }
}

bool make_lds_args(const DevPlan& p, const FragView& fv, int n_cus, LdsArgs* out) {
  LdsArgs& a = *out;
  bool need[kLdsVals][4] = {};
  if (!lds_describe(p, fv, 65536, tune_knobs().flags, &a, need, true)) return false;
  // one replica: 8-byte arrays first, then the 4-byte counters
  auto lay_out = [&](uint32_t entries) -> uint32_t {
    uint32_t off = 0;
    a.entries = entries;
    a.off_keys = -1;
    if (a.baseline) {
      a.off_keys = (int32_t)off;
      off += entries * 8;
    }
    for (int c = 0; c < a.n_vals; ++c)
      for (int k = 1; k < 4; ++k) {
        int32_t& o = k == 1 ? a.v[c].off_sum : k == 2 ? a.v[c].off_min : a.v[c].off_max;
        o = -1;
        if (need[c][k]) {
          o = (int32_t)off;
          off += entries * 8;
        }
      }
    a.off_rows = (int32_t)off;
    off += entries * 4;
    for (int c = 0; c < a.n_vals; ++c) {
      a.v[c].off_cnt = -1;
      if (need[c][0]) {
        a.v[c].off_cnt = (int32_t)off;
        off += entries * 4;
      }
    }
    return (off + 15u) & ~15u;
  };
  // typed member (k_groupby_lds_typed): up to three plain INT32 filter columns, 1 - 3 plain INT32 value columns; perfect hash over INT32 keys whose
  // ranges keep the index arithmetic in 32 bits, or a baseline table over one BIGINT / DOUBLE / FLOAT / INT key
  // (no value column at all — COUNT(*) alone — is typed only under a filter: the unfiltered count shapes have members of
  // their own, k_perfect_lds's count program and the index family's count-only member)
  a.typed = a.n_flt <= kTypedFlt && (a.n_vals >= 1 || a.n_flt >= 1) && !(tune_knobs().flags & MI355Q_OPT_LDS_GENERIC_MEMBER);
  for (int k = 0; k < a.n_flt; ++k) a.typed = a.typed && (a.flt_type[k] == MI355Q_INT32 || (a.flt_type[k] == MI355Q_INT8 && !a.bf_on));
  a.mm = 0;
  for (int c = 0; c < a.n_vals; ++c) {
    a.typed = a.typed && a.v[c].type == MI355Q_INT32;
    a.mm |= (need[c][2] || need[c][3]) ? 1 : 0;
  }
  if (!a.baseline)
    for (int g = 0; g < a.n_keys; ++g)
      a.typed = a.typed && a.key_type[g] == MI355Q_INT32 && a.key_min[g] > -(1ll << 30) && a.key_min[g] < (1ll << 30) &&
                a.key_card[g] >= 1 && a.key_card[g] <= 65536 && a.key_mul[g] >= 1 && a.key_mul[g] <= 65536;
  auto lay_out_typed = [&](uint32_t entries) -> uint32_t {
    const uint32_t nv = (uint32_t)a.n_vals;
    uint32_t off = 0;
    a.entries = entries;
    a.t_off_keys = 0;
    if (a.baseline) off += entries * 8;
    a.t_off_sum = off;
    off += nv * entries * 8;
    a.t_off_rows = off;
    off += entries * 4;
    a.t_off_cnt = off;
    off += nv * entries * 4;
    a.t_off_min = a.t_off_max = 0;
    if (a.mm) {
      a.t_off_min = off;
      off += nv * entries * 4;
      a.t_off_max = off;
      off += nv * entries * 4;
    }
    return (off + 15u) & ~15u;
  };
  auto lay = [&](uint32_t entries) -> uint32_t { return a.typed ? lay_out_typed(entries) : lay_out(entries); };
  a.copy_bytes = lay(a.entries);
  // (second baseline attempt: the largest power-of-two replica the accumulators leave room for)
  while (a.baseline && a.copy_bytes > kLdsBudget && a.entries > kLdsHashSmall) a.copy_bytes = lay(a.entries / 2);

  // a perfect-hash table larger than the LDS: the fewest windows whose share fits (one replica each)
  if (!a.baseline && a.copy_bytes > kLdsBudget) {
    const uint32_t total = (uint32_t)p.entry_count;
    for (uint32_t T = 2; T <= kLdsMaxWindows; ++T) {
      const uint32_t share = (total + T - 1) / T;
      if (lay(share) <= kLdsBudget) {
        a.windows = T;
        a.copy_bytes = lay(share);
        break;
      }
    }
    if (a.windows == 1) a.copy_bytes = lay(total);  // (does not fit: refused below)
  }
  if (a.copy_bytes > kLdsBudget) return false;
  a.copies_lg = 0;
  while (a.copies_lg < 6 && ((size_t)a.copy_bytes << (a.copies_lg + 1)) <= kLdsBudget) ++a.copies_lg;
  // the per-entry row / non-NULL counters are 32 bits wide per workgroup: a stripe must stay below 2^32 rows (ADVICE r03;
  // only reachable with a handful of workgroups — tune_cus, or eight windows on a small device — and > 4 G rows)
  {
    int64_t stripes = (n_cus > 0 ? n_cus : 1) / (int64_t)a.windows;
    if (stripes < 1) stripes = 1;
    if (fv.total_rows / stripes >= 0xfff00000ll) return false;
  }
  // the run-time-role member's widest instantiation filters on NF = 4 columns (launch_lds_groupby); a step with more
  // range filters than that (e.g. x <> 1 AND … AND x <> 5: negated quals do not merge) goes to another family instead of
  // running with the quals past the fourth dropped
  if (!a.typed && a.n_flt > kLdsGenericFlt) return false;
  if (!a.typed && a.bf_on && step_bool_filter() && step_bool_filter()->n_progs != 0) return false;  // (program atoms: the typed member or the pre-pass)
  return a.n_flt + a.n_keys + a.n_vals <= 8;
}

}  // namespace

bool lds_groupby_eligible(const DevPlan& p, const FragView& fv, int n_cus) {
  LdsArgs a;
  return make_lds_args(p, fv, n_cus, &a);
}
bool lds_groupby_typed_eligible(const DevPlan& p, const FragView& fv, int n_cus) {
  LdsArgs a;
  return make_lds_args(p, fv, n_cus, &a) && a.typed;
}

namespace {
struct LdsLaunch {
  const FragView& fv;
  const LdsArgs& a;
  const DevPlan& p;
  int64_t* out;
  int32_t* d_err;
  int grid;
  size_t lds;
  hipStream_t s;
};
template <int KK, int NK, int NV, bool MM>
void launch_typed_member(const LdsLaunch& l) {
  // two quads per lane and column while the tile (this one and the next in flight) stays within ~64 registers
  constexpr int UQ = (NK * (KK == 1 ? 2 : 1) + NV) <= 3 ? 2 : 1;
  if (l.a.n_flt > 0 && l.a.bf_on && step_bool_filter() && step_bool_filter()->n_progs != 0) {  // ... with lean program atoms
    auto k = k_groupby_lds_typed<KK, NK, NV, MM, 1, 2>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds);
    hipLaunchKernelGGL(k, dim3(l.grid), dim3(kLdsBlock), l.lds, l.s, l.fv.d_cols, l.fv.d_num_rows, l.fv.n_frags, l.fv.n_cols,
                       l.a, l.p, l.out, l.d_err);
    return;
  }
  if (l.a.n_flt > 0) {  // the filtered member: one quad per lane and column (up to three more columns in the tile)
    auto k = k_groupby_lds_typed<KK, NK, NV, MM, 1, 1>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds);
    hipLaunchKernelGGL(k, dim3(l.grid), dim3(kLdsBlock), l.lds, l.s, l.fv.d_cols, l.fv.d_num_rows, l.fv.n_frags, l.fv.n_cols,
                       l.a, l.p, l.out, l.d_err);
    return;
  }
  auto k = k_groupby_lds_typed<KK, NK, NV, MM, UQ, 0>;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.lds);
  hipLaunchKernelGGL(k, dim3(l.grid), dim3(kLdsBlock), l.lds, l.s, l.fv.d_cols, l.fv.d_num_rows, l.fv.n_frags, l.fv.n_cols,
                     l.a, l.p, l.out, l.d_err);
}
template <int KK, int NK>
void launch_typed_nv(const LdsLaunch& l) {
  const bool mm = l.a.mm != 0;
  switch (l.a.n_vals) {
    case 0: launch_typed_member<KK, NK, 0, false>(l); break;  // COUNT(*) alone: the entries' row counters are the result
    case 1: mm ? launch_typed_member<KK, NK, 1, true>(l) : launch_typed_member<KK, NK, 1, false>(l); break;
    case 2: mm ? launch_typed_member<KK, NK, 2, true>(l) : launch_typed_member<KK, NK, 2, false>(l); break;
    default: mm ? launch_typed_member<KK, NK, 3, true>(l) : launch_typed_member<KK, NK, 3, false>(l);
  }
}
void launch_typed(const LdsLaunch& l) {
  if (!l.a.baseline) {
    if (l.a.n_keys == 1) launch_typed_nv<0, 1>(l);
    else if (l.a.n_keys == 2) launch_typed_nv<0, 2>(l);
    else launch_typed_nv<0, 3>(l);
  } else {
    const int kt = l.a.key_type[0];
    if (kt == MI355Q_INT64 || kt == MI355Q_DOUBLE) launch_typed_nv<1, 1>(l);
    else if (kt == MI355Q_FLOAT) launch_typed_nv<2, 1>(l);
    else launch_typed_nv<3, 1>(l);
  }
}
}  // namespace

hipError_t launch_lds_groupby(const DevPlan& p, const FragView& fv, int64_t* out, int32_t* d_err, int n_cus,
                              hipStream_t s, LaunchStats* st) {
  LdsArgs a;
  if (!make_lds_args(p, fv, n_cus, &a)) return hipErrorInvalidValue;
  const size_t lds = ((size_t)a.copy_bytes << a.copies_lg) + 16;  // + the workgroup's "a replica is full" word
  int64_t want = (fv.total_rows / 4 + kLdsBlock - 1) / kLdsBlock;
  if (want < 1) want = 1;
  // windows: T workgroups (one per window) share a row stripe
  const int T = (int)a.windows;
  int64_t stripes = n_cus / T;
  if (stripes > want) stripes = want;
  if (stripes < 1) stripes = 1;
  int grid = (int)stripes * T;
  // windows on a full device: the T workgroups of a stripe on one XCD, i.e. per XCD a whole number of stripes
  a.xcd_aware = 0;
  if (T > 1 && n_cus >= 64 && (n_cus / 8) / T >= 1 && stripes * T >= n_cus - T) {
    const int per_xcd = (n_cus / 8) / T;
    grid = 8 * T * per_xcd;
    a.xcd_aware = 1;
  }
  st->kernel_name = "k_groupby_lds";
  st->n_launches = 1;
  st->variant = a.typed ? 5 : 4;  // 5: a typed member (k_groupby_lds_typed)
  rec(st->k_start, s);
  const int streams = a.n_flt + a.n_keys + a.n_vals;
#define MQ_LDS_LAUNCH(NF, NK, NV, UQ)                                                                                \
  do {                                                                                                               \
    auto k = k_groupby_lds<NF, NK, NV, UQ>;                                                                          \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    hipLaunchKernelGGL(k, dim3(grid), dim3(kLdsBlock), lds, s, fv.d_cols, fv.d_num_rows, fv.n_frags, fv.n_cols, a, p, \
                       out, d_err);                                                                                  \
  } while (0)
  (void)streams;
  if (a.typed) launch_typed(LdsLaunch{fv, a, p, out, d_err, grid, lds, s});
  else if (a.n_flt <= 1 && a.n_keys == 1 && a.n_vals <= 1) MQ_LDS_LAUNCH(1, 1, 1, 2);       // PHS / BH shapes
  else if (a.n_flt <= 1 && a.n_vals <= 1) MQ_LDS_LAUNCH(1, 3, 1, 1);                     // PHM shapes
  else if (a.n_flt <= 1 && a.n_keys == 1) MQ_LDS_LAUNCH(1, 1, 3, 1);                     // MultiStep, one key
  else MQ_LDS_LAUNCH(kLdsGenericFlt, 3, 3, 1);
#undef MQ_LDS_LAUNCH
  rec(st->k_stop, s);
  return hipGetLastError();
}

}  // namespace mq
