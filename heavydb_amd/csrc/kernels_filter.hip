// kernels_filter.hip — the ROW MASK of a filter compiled at plan time whose atoms include programs (round 6).
//
// The reference compiles a WHERE clause into the row function (Executor::compileBody, NativeCodegen.cpp:3455; arithmetic
// leaves: codegenArith / codegenDiv, ArithmeticIR.cpp:39-560; the short-circuit forms behind prioritizeQuals,
// LogicalIR.cpp:158-297).  Here a filter of comparisons with literals is evaluated INSIDE the consuming kernel (atoms +
// truth table, boolfilter.h).  A filter with PROGRAM atoms — `b <> 0 AND a / b > 3`, `x + y > 100`, `a < b`, DOUBLE
// leaves: two-register programs of typed steps, regprog.h — is evaluated by this pre-pass instead: it streams the
// filter's columns ONCE (16-byte loads, two quads per lane in flight), runs atoms, programs and the truth table on the four
// rows of a quad together and leaves ONE BYTE per row (1 = the row passes).  The step proper then runs without the
// filter's columns and with the qual `mask = 1` on that 1-byte column — which every typed family loads as one 4-byte word
// per quad.  Bytes: the filter's columns are read once either way; the mask adds 1 B/row written + 1 B/row read (the
// interpreter pass of rounds 3-5 wrote and re-read a 4-byte column per expression and ran ~450 wave instructions per 64
// rows).  An error a row raises (error 7 / error 1) ends the step exactly as the row function's would: every row evaluates
// the filter's expressions, whatever the plain quals say of it.
#include "boolfilter.h"
#include "fast_common.h"
#include "kernels.h"

namespace mq {

using namespace fast;

namespace {

constexpr int kFmBlock = 256;

struct FilterMaskArgs {
  int32_t n_flt, n_frags, n_cols_table, pad_;
  int32_t col[kBfMaxCols], width[kBfMaxCols];  // the filter's columns: index in the column table, bytes per value (4 / 8)
  const BoolFilter* bf;                         // DEVICE memory
  const int8_t* const* cols;
  const int64_t* num_rows;
  int8_t* const* mask;                          // per fragment: ceil(n / 4) * 4 bytes, 16-byte aligned
  int32_t* d_err;
};

template <int NF>
MQ_D void fm_load_quad(const FilterMaskArgs& a, const int8_t* const* fc, int64_t quad, v4i32 (&lo)[NF], v4i32 (&hi)[NF]) {
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    if (k >= a.n_flt) break;
    const int8_t* base = fc[a.col[k]];
    if (a.width[k] == 8) {
      lo[k] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2);
      hi[k] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad * 2 + 1);
    } else {
      lo[k] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base + quad);
    }
  }
}
MQ_D int32_t fm_v4(const v4i32& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
template <int NF>
MQ_D void fm_values(const FilterMaskArgs& a, const v4i32 (&lo)[NF], const v4i32 (&hi)[NF], int64_t (&vals)[4][NF]) {
#pragma unroll
  for (int k = 0; k < NF; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (k >= a.n_flt) {
        vals[i][k] = 0;
      } else if (a.width[k] == 8) {
        const v4i32& h = i < 2 ? lo[k] : hi[k];
        const int j = (i & 1) * 2;
        vals[i][k] = (int64_t)(((uint64_t)(uint32_t)fm_v4(h, j + 1) << 32) | (uint64_t)(uint32_t)fm_v4(h, j));
      } else {
        vals[i][k] = (int64_t)fm_v4(lo[k], i);
      }
    }
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
#define FM_UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#else
#define FM_UNIFORM(x) ((int)(x))
#endif
// what bf_quad_pass (boolfilter.h) reads of the filter per quad, read ONCE per workgroup into scalar registers: every one
// of those reads is an LDS round trip the next one waits for, and a quad's evaluation is a chain of them (the first
// version of this kernel ran 5.3 - 8.0 ms per 1 B rows with 3 waves per SIMD: latency, not bandwidth or arithmetic)
struct FmMeta {
  int n_progs;
  uint32_t atoms_of;   // range atoms of filter column c: byte c
  uint64_t prog_meta;  // program k: 16 bits — operand columns (4 bits each), can_raise (bit 8)
};
template <int NF>
MQ_D uint32_t fm_quad_pass(const BoolFilter& bf, const FmMeta& mt, const int64_t (&vals)[4][NF], uint32_t valid, int32_t* err) {
  uint32_t idx[4] = {0, 0, 0, 0}, mul = 1;
  int ai = 0;
#pragma unroll
  for (int c = 0; c < NF; ++c) {
    const int cnt = (int)((mt.atoms_of >> (8 * c)) & 255u);
    for (int k = 0; k < cnt; ++k) {
      const BoolAtom at = bf.atom[ai];
#pragma unroll
      for (int j = 0; j < 4; ++j) idx[j] += bf_atom_state(at, vals[j][c]) * mul;
      mul *= 3u;
      ++ai;
    }
  }
  uint32_t epack[4] = {0, 0, 0, 0};  // two bits per program atom: the error it raised (ex_err_enc)
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int k = 0; k < mt.n_progs; ++k) {
    const uint32_t pm = (uint32_t)(mt.prog_meta >> (16 * k)) & 0xffffu;
    const int ca = (int)(pm & 15u), cb = (int)((pm >> 4) & 15u);
    int64_t ops[4][2], outv[4];
    int32_t e4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ops[j][0] = vals[j][0];
      ops[j][1] = vals[j][0];
#pragma unroll
      for (int c = 1; c < NF; ++c) {
        if (ca == c) ops[j][0] = vals[j][c];
        if (cb == c) ops[j][1] = vals[j][c];
      }
    }
    rp_eval<4, 2>(bf.prog[k], ops, outv, e4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t st = e4[j] ? 3u : outv[j] == 1 ? 1u : outv[j] == 0 ? 0u : 2u;  // (anything else is the INT8 NULL)
      idx[j] += st * mul;
      epack[j] |= ex_err_enc(e4[j]) << (2 * k);
    }
    mul *= ((pm >> 8) & 1u) ? 4u : 3u;
  }
  uint32_t w4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = bf.table[idx[j] >> 5];  // (four independent reads: one round trip)
  uint32_t pass = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t bit = (w4[j] >> (idx[j] & 31u)) & 1u;
    if (epack[j] && ((valid >> j) & 1u)) {  // rare: an atom of this row is in its ERROR state — is it the row's outcome?
      const uint32_t nib = (bf.etable[idx[j] >> 3] >> ((idx[j] & 7u) * 4u)) & 15u;
      if (nib) {
        if (!*err) *err = ex_err_dec((epack[j] >> (2u * (nib - 1u))) & 3u);
        bit = 0;
      }
    }
    pass |= bit << j;
  }
  return pass & valid;
}

// NF: filter columns this member holds registers for
template <int NF>
__global__ __launch_bounds__(kFmBlock) void k_filter_mask(FilterMaskArgs a) {
  __shared__ BoolFilter s_bf;
  bf_load(a.bf, &s_bf, threadIdx.x, kFmBlock);
  __syncthreads();
  FmMeta mt;
  mt.n_progs = FM_UNIFORM(s_bf.n_progs);
  mt.atoms_of = 0;
  mt.prog_meta = 0;
#pragma unroll
  for (int c = 0; c < kBfMaxCols; ++c) mt.atoms_of |= (uint32_t)FM_UNIFORM(c < s_bf.n_cols ? s_bf.atoms_of_col[c] : 0) << (8 * c);
#pragma unroll
  for (int k = 0; k < kBfMaxProgs; ++k)
    mt.prog_meta |= (uint64_t)(uint32_t)FM_UNIFORM((s_bf.prog_op[k][0] & 15) | ((s_bf.prog_op[k][1] & 15) << 4) | (s_bf.prog[k].can_raise ? 256 : 0)) << (16 * k);
  int32_t err = 0;
  const int64_t gtid = (int64_t)blockIdx.x * kFmBlock + threadIdx.x;
  const int64_t gsize = (int64_t)gridDim.x * kFmBlock;
  for (int f = 0; f < a.n_frags; ++f) {
    const int8_t* const* fc = a.cols + (size_t)f * a.n_cols_table;
    const int64_t n = a.num_rows[f];
    const int64_t nq = n >> 2;
    uint32_t* const out = (uint32_t*)a.mask[f];
    // ONE copy of the evaluator in the kernel (the programs' typed members are inlined four rows wide: ~5 K instructions).
    // Every lane walks the fragment's quads with the loads of its NEXT quad in flight while this one's programs run; the
    // fragment's last, partial quad takes the same path with its rows loaded one by one (a row past the end repeats the
    // last one and is not valid).
    const int64_t nq_all = (n + 3) >> 2;
    auto load_vals = [&](int64_t quad, v4i32 (&lo)[NF], v4i32 (&hi)[NF]) {
      if (quad < nq) {
        fm_load_quad<NF>(a, fc, quad, lo, hi);
      } else if (quad < nq_all) {  // (one lane per fragment)
        const int left = (int)(n & 3);
#pragma unroll
        for (int k = 0; k < NF; ++k) {
          if (k >= a.n_flt) break;
          int64_t v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t pos = (nq << 2) + (i < left ? i : left - 1);
            v[i] = a.width[k] == 8 ? load_one<int64_t>(fc[a.col[k]], pos) : (int64_t)load_one<int32_t>(fc[a.col[k]], pos);
          }
          if (a.width[k] == 8) {
            lo[k].x = (int)(uint32_t)v[0]; lo[k].y = (int)(uint32_t)((uint64_t)v[0] >> 32); lo[k].z = (int)(uint32_t)v[1]; lo[k].w = (int)(uint32_t)((uint64_t)v[1] >> 32);
            hi[k].x = (int)(uint32_t)v[2]; hi[k].y = (int)(uint32_t)((uint64_t)v[2] >> 32); hi[k].z = (int)(uint32_t)v[3]; hi[k].w = (int)(uint32_t)((uint64_t)v[3] >> 32);
          } else {
            lo[k].x = (int)v[0]; lo[k].y = (int)v[1]; lo[k].z = (int)v[2]; lo[k].w = (int)v[3];
          }
        }
      }
    };
    v4i32 lo_c[NF], hi_c[NF], lo_n[NF], hi_n[NF];
    int64_t q = gtid;
    load_vals(q, lo_c, hi_c);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (; q < nq_all; q += gsize) {
      load_vals(q + gsize, lo_n, hi_n);
      int64_t vals[4][NF];
      fm_values<NF>(a, lo_c, hi_c, vals);
      const uint32_t valid = q < nq ? 15u : (1u << (int)(n & 3)) - 1u;
      const uint32_t m = fm_quad_pass<NF>(s_bf, mt, vals, valid, &err);
      // one byte per row: 1 = the row passes
      const uint32_t w = (m & 1u) | ((m & 2u) << 7) | ((m & 4u) << 14) | ((m & 8u) << 21);
      __builtin_nontemporal_store(w, (MQ_GLOBAL uint32_t*)out + q);
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        lo_c[k] = lo_n[k];
        hi_c[k] = hi_n[k];
      }
    }
  }
  if (err) atomicCAS(a.d_err, 0, err);
}

// ---- the LEAN member: every filter column a plain INT32, every program a PairAtom (boolfilter.h): `a <cmp> b`,
// `(a <op> b) <cmp> literal` at INT32.  Values stay 32 bits wide, a program is ONE typed operation per row behind a few
// scalar branches (no steps, no register selects per step), the tile loop is uniform (a workgroup walks tiles of kFmBlock
// quads, the next tile's loads in flight while this one's rows are looked at).
constexpr int kFmHoisted = 2;  // pair atoms kept in scalar registers for the whole kernel (the rest are read per quad)
struct FmPairs {
  PairAtom pa[kFmHoisted];
};
template <int NF>
MQ_D uint32_t fm_quad_pass_i32(const BoolFilter& bf, const FmMeta& mt, const FmPairs& hp, const v4i32 (&col)[NF], uint32_t valid, int32_t* err) {
  uint32_t idx[4] = {0, 0, 0, 0}, mul = 1;
  int ai = 0;
#pragma unroll
  for (int c = 0; c < NF; ++c) {
    const int cnt = (int)((mt.atoms_of >> (8 * c)) & 255u);
    for (int k = 0; k < cnt; ++k) {
      const BoolAtom at = bf.atom[ai];
#pragma unroll
      for (int j = 0; j < 4; ++j) idx[j] += bf_atom_state(at, (int64_t)fm_v4(col[c], j)) * mul;
      mul *= 3u;
      ++ai;
    }
  }
  uint32_t epack[4] = {0, 0, 0, 0};
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int k = 0; k < mt.n_progs; ++k) {
    const uint32_t pm = (uint32_t)(mt.prog_meta >> (16 * k)) & 0xffffu;
    const int ca = (int)(pm & 15u), cb = (int)((pm >> 4) & 15u);
    PairAtom pa;
    if (k == 0) {
      pa = hp.pa[0];
    } else if (k == 1) {
      pa = hp.pa[1];
    } else {
      pa = bf.pair[k];
      pa.op = FM_UNIFORM(pa.op);
      pa.ln = FM_UNIFORM(pa.ln);
      pa.rn = FM_UNIFORM(pa.rn);
      pa.b_is_lit = FM_UNIFORM(pa.b_is_lit);
      pa.b_lit = FM_UNIFORM(pa.b_lit);
      pa.lo = FM_UNIFORM(pa.lo);
      pa.hi = FM_UNIFORM(pa.hi);
      pa.negate = FM_UNIFORM(pa.negate);
      pa.op2 = FM_UNIFORM(pa.op2);
      pa.lit2 = FM_UNIFORM(pa.lit2);
    }
    v4i32 av = col[0], bv = col[0];
#pragma unroll
    for (int c = 1; c < NF; ++c) {
      if (ca == c) av = col[c];
      if (cb == c) bv = col[c];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int32_t e = 0;
      const uint32_t st = pair_eval(pa, fm_v4(av, j), fm_v4(bv, j), e);
      idx[j] += st * mul;
      epack[j] |= ex_err_enc(e) << (2 * k);
    }
    mul *= ((pm >> 8) & 1u) ? 4u : 3u;
  }
  uint32_t w4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = bf.table[idx[j] >> 5];  // (four independent reads: one round trip)
  uint32_t pass = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t bit = (w4[j] >> (idx[j] & 31u)) & 1u;
    if (epack[j] && ((valid >> j) & 1u)) {  // rare: an atom of this row is in its ERROR state — is it the row's outcome?
      const uint32_t nib = (bf.etable[idx[j] >> 3] >> ((idx[j] & 7u) * 4u)) & 15u;
      if (nib) {
        if (!*err) *err = ex_err_dec((epack[j] >> (2u * (nib - 1u))) & 3u);
        bit = 0;
      }
    }
    pass |= bit << j;
  }
  return pass & valid;
}

template <int NF>
__global__ __launch_bounds__(kFmBlock) void k_filter_mask_i32(FilterMaskArgs a) {
  __shared__ BoolFilter s_bf;
  bf_load(a.bf, &s_bf, threadIdx.x, kFmBlock);
  __syncthreads();
  FmMeta mt;
  mt.n_progs = FM_UNIFORM(s_bf.n_progs);
  mt.atoms_of = 0;
  mt.prog_meta = 0;
#pragma unroll
  for (int c = 0; c < kBfMaxCols; ++c) mt.atoms_of |= (uint32_t)FM_UNIFORM(c < s_bf.n_cols ? s_bf.atoms_of_col[c] : 0) << (8 * c);
#pragma unroll
  for (int k = 0; k < kBfMaxProgs; ++k)
    mt.prog_meta |= (uint64_t)(uint32_t)FM_UNIFORM((s_bf.prog_op[k][0] & 15) | ((s_bf.prog_op[k][1] & 15) << 4) | (s_bf.prog[k].can_raise ? 256 : 0)) << (16 * k);
  FmPairs hp;
#pragma unroll
  for (int k = 0; k < kFmHoisted; ++k) {
    const PairAtom& src = s_bf.pair[k];
    hp.pa[k] = PairAtom{};
    hp.pa[k].op = FM_UNIFORM(src.op);
    hp.pa[k].ln = FM_UNIFORM(src.ln);
    hp.pa[k].rn = FM_UNIFORM(src.rn);
    hp.pa[k].b_is_lit = FM_UNIFORM(src.b_is_lit);
    hp.pa[k].b_lit = FM_UNIFORM(src.b_lit);
    hp.pa[k].lo = FM_UNIFORM(src.lo);
    hp.pa[k].hi = FM_UNIFORM(src.hi);
    hp.pa[k].negate = FM_UNIFORM(src.negate);
    hp.pa[k].op2 = FM_UNIFORM(src.op2);
    hp.pa[k].lit2 = FM_UNIFORM(src.lit2);
  }
  int32_t err = 0;
  const int tid = threadIdx.x;
  for (int f = 0; f < a.n_frags; ++f) {
    const int8_t* const* fc = a.cols + (size_t)f * a.n_cols_table;
    const int64_t n = a.num_rows[f];
    const int64_t nq = n >> 2;
    uint32_t* const out = (uint32_t*)a.mask[f];
    const int8_t* base[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) base[k] = k < a.n_flt ? fc[a.col[k]] : nullptr;
    const int64_t n_tiles = (nq + kFmBlock - 1) / kFmBlock;
    auto load_tile = [&](int64_t t, v4i32 (&col)[NF]) {
      const int64_t q = t * kFmBlock + tid;
      if (q < nq) {
#pragma unroll
        for (int k = 0; k < NF; ++k)
          if (k < a.n_flt) col[k] = __builtin_nontemporal_load((const MQ_GLOBAL v4i32*)base[k] + q);
      }
    };
    v4i32 cur[NF], nxt[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) cur[k] = nxt[k] = v4i32{0, 0, 0, 0};
    int64_t t = (blockIdx.x + (int64_t)f * 7) % gridDim.x;
    if (t < n_tiles) load_tile(t, nxt);
    for (; t < n_tiles; t += gridDim.x) {  // (uniform)
#pragma unroll
      for (int k = 0; k < NF; ++k) cur[k] = nxt[k];
      if (t + gridDim.x < n_tiles) load_tile(t + gridDim.x, nxt);
      const int64_t q = t * kFmBlock + tid;
      const uint32_t m = fm_quad_pass_i32<NF>(s_bf, mt, hp, cur, q < nq ? 15u : 0u, &err);
      const uint32_t w = (m & 1u) | ((m & 2u) << 7) | ((m & 4u) << 14) | ((m & 8u) << 21);
      if (q < nq) __builtin_nontemporal_store(w, (MQ_GLOBAL uint32_t*)out + q);
    }
    // the fragment's last, partial quad: its rows one by one (a row past the end repeats the last one and is not valid)
    if ((n & 3) && blockIdx.x == (unsigned)(f % (int)gridDim.x) && tid == 0) {
      const int left = (int)(n & 3);
      v4i32 col[NF];
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        col[k] = v4i32{0, 0, 0, 0};
        if (k < a.n_flt) {
          const int64_t p0 = nq << 2;
          col[k].x = load_one<int32_t>(base[k], p0);
          col[k].y = load_one<int32_t>(base[k], p0 + (left > 1 ? 1 : 0));
          col[k].z = load_one<int32_t>(base[k], p0 + (left > 2 ? 2 : left - 1));
          col[k].w = col[k].z;
        }
      }
      const uint32_t m = fm_quad_pass_i32<NF>(s_bf, mt, hp, col, (1u << left) - 1u, &err);
      const uint32_t w = (m & 1u) | ((m & 2u) << 7) | ((m & 4u) << 14) | ((m & 8u) << 21);
      out[nq] = w;
    }
  }
  if (err) atomicCAS(a.d_err, 0, err);
}

}  // namespace

int64_t filter_mask_chunk_bytes(int64_t n_rows) { return ((((n_rows + 3) >> 2) << 2) + 15) & ~(int64_t)15; }

bool filter_mask_eligible(const BoolFilter& bf, const FragView& fv) {
  if (bf.n_cols < 1 || bf.n_cols > kBfMaxCols) return false;
  for (int k = 0; k < bf.n_cols; ++k) {
    const int t = bf.col_type[k];
    if (t != MI355Q_INT32 && t != MI355Q_INT64 && t != MI355Q_DOUBLE) return false;
    if (!all_aligned16(fv, bf.col[k])) return false;
  }
  return true;
}

hipError_t launch_filter_mask(const BoolFilter& bf, const BoolFilter* d_bf, const FragView& fv, int8_t* const* d_mask, int32_t* d_err,
                              int n_cus, hipStream_t s) {
  FilterMaskArgs a{};
  a.n_flt = bf.n_cols;
  a.n_frags = fv.n_frags;
  a.n_cols_table = fv.n_cols;
  for (int k = 0; k < bf.n_cols; ++k) {
    a.col[k] = bf.col[k];
    a.width[k] = bf.col_type[k] == MI355Q_INT32 ? 4 : 8;
  }
  a.bf = d_bf;
  a.cols = fv.d_cols;
  a.num_rows = fv.d_num_rows;
  a.mask = d_mask;
  a.d_err = d_err;
  int64_t want = (fv.max_frag_rows / 4 + kFmBlock - 1) / kFmBlock;
  int64_t grid = (int64_t)n_cus * 8;
  if (grid > want) grid = want;
  if (grid < 1) grid = 1;
  const bool lean = bf.all_lean && bf.all_i32 && !(tune_knobs().flags & MI355Q_OPT_LDS_GENERIC_MEMBER);
  if (lean && bf.n_cols <= 2) hipLaunchKernelGGL(k_filter_mask_i32<2>, dim3((unsigned)grid), dim3(kFmBlock), 0, s, a);
  else if (lean) hipLaunchKernelGGL(k_filter_mask_i32<kBfMaxCols>, dim3((unsigned)grid), dim3(kFmBlock), 0, s, a);
  else if (bf.n_cols <= 2) hipLaunchKernelGGL(k_filter_mask<2>, dim3((unsigned)grid), dim3(kFmBlock), 0, s, a);
  else hipLaunchKernelGGL(k_filter_mask<kBfMaxCols>, dim3((unsigned)grid), dim3(kFmBlock), 0, s, a);
  return hipGetLastError();
}

}  // namespace mq
