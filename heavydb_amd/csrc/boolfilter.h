// boolfilter.h — BOOLEAN filters compiled at plan time for the typed kernel families: no temporary column, no
// interpreter pass.
//
// The reference compiles a WHERE clause into the row function (Executor::compileBody, NativeCodegen.cpp:3455: filters
// first, then the body; codegenLogical / codegenCmp / codegenIsNull, LogicalIR.cpp:299-432, CompareIR.cpp:230-330).
// A filter whose leaves are comparisons of a column with a literal (`x < 5`, `5 <= y`, `z IS NULL`; casts to a wider
// integer are transparent) under any nest of AND / OR / NOT — plain or short-circuit — cannot raise an error, so all
// that matters of a row is, per leaf ("atom"), one of three states: FALSE, TRUE, NULL.  At plan time the filter is
// therefore reduced to
//   * up to kBfMaxAtoms atoms over up to kBfMaxCols INT32 / INT64 physical columns, each normalised to
//     lo <= v <= hi (+ negation, + the column's NULL pattern): what make_range_filter does for a plain qual;
//   * a truth table of 3^n bits, indexed by the atoms' states in base 3, filled by running the filter's own micro-op
//     program (expr.h eval_expr — the evaluator the interpreter pass uses, so the three-valued and short-circuit rules
//     are the same code) once per state vector with the atoms replaced by literals.
// A kernel evaluates the atoms on the column values it already holds in registers and looks one bit up in LDS.
//
// Round 6 — PROGRAM atoms.  A leaf that is not `column <op> literal` — `a / b > 3`, `x + y > 100`, `a < b`, `f * 2.0 < g`,
// `CAST(x AS BIGINT) * 1000 >= k`, `d < 0.5` on a DOUBLE column — is an atom too when its operand subtree never holds more
// than two live values: it is compiled into a two-register program of typed steps (regprog.h: one instantiation of the
// expr.h function per operation and type, a wave-uniform switch per step and FOUR rows) that leaves the comparison's
// BOOLEAN.  Such an atom can RAISE (error 7 / error 1), so it has a fourth state, ERROR; whether the error of an atom in
// that state is the row's outcome depends on the states of the others (`b <> 0 AND a / b > 3` in the short-circuit form
// never evaluates the division where b = 0 — the reference's deferred quals, LogicalIR.cpp:158-297), which the plan-time
// run over all state vectors records next to the truth table: a 4-bit "which atom raises" table, filled by the same
// ex_logic / ex_not / ex_is_null (expr.h) that propagate errors in the interpreter.
// Filters with CASE, leaves of more than two live values, INT8 / INT16 / FLOAT arithmetic or encoded columns are not
// taken (the step then runs through the projection pass, kernels_generic.hip k_project).
#pragma once

#include <cstddef>

#include "dev_common.h"
#include "fast_common.h"
#include "regprog.h"

namespace mq {

constexpr int kBfMaxAtoms = 8;
constexpr int kBfMaxCols = 4;
constexpr int kBfTableWords = 206;  // ceil(3^8 / 32)
constexpr int kBfMaxProgs = 4;      // program atoms (they count towards kBfMaxAtoms)
constexpr int kLdsFusedProgs = 2;   // lean program atoms the typed few-groups member (kernels_lds.hip) evaluates itself
constexpr int kBfErrStates = 2048;  // state vectors of a filter whose atoms can raise (a nibble each)

struct BoolAtom {
  int64_t lo, hi, null_val;
  int32_t negate, nullable;  // nullable: the column can hold null_val, and the atom is then NULL (IS NULL atoms: 0)
};
// A program atom in its LEAN form (round 6): the program is `a <cmp> b` over two INT32 columns, or `(a <op> b) <cmp> literal`
// with a an INT32 column, b an INT32 column or literal and <op> one of + - * / % at INT32 — the shapes of `a / b > 3`,
// `x + y > 100`, `a < b`.  pair_eval below states the same semantics as the program's steps (ex_arith / ex_divmod / ex_cmp,
// expr.h) on 32-bit values instead of 64-bit patterns; tests/test_expr.py holds the two against each other on every edge.
struct PairAtom {
  int32_t lean;          // 1: this form states the program
  int32_t op;            // 0: compare the two operands; else MI355Q_EX_ADD .. MI355Q_EX_MOD
  int32_t ln, rn;        // the operands can be NULL (INT32_MIN)
  int32_t b_is_lit, b_lit;
  int32_t lo, hi, negate;  // the comparison as a range of the value (op 0: of sign(a - b))
  int32_t op2, lit2;     // a CHAIN: ((a <op> b) <op2> lit2) <cmp> literal — `a * 3 - 7 <= k`; 0: none
  int32_t pad_;
};
// one INT32 operation of a pair atom: -> 0 (the value is in v), 2 NULL, 3 ERROR (err set).  ex_arith / ex_divmod (expr.h) on
// 32-bit values: NULL operands give NULL before any check except MOD's zero test; DIV skips its zero test behind a NULL
// pattern as soon as one operand may be NULL
MQ_HD uint32_t pair_arith(int op, int32_t a, int32_t b, bool ln, bool rn, int32_t& v, int32_t& err) {
  const int32_t nul = INT32_MIN;
  const bool is_null = (ln && a == nul) || (rn && b == nul);
  if (op == MI355Q_EX_DIV || op == MI355Q_EX_MOD) {
    const bool skip = op == MI355Q_EX_DIV && (ln || rn) && (a == nul || b == nul);
    if (!skip && b == 0) {
      err = MI355Q_ERR_DIV_BY_ZERO;
      return 3u;
    }
    if (is_null) return 2u;
    if (b == 0) v = nul;
    else if (b == -1) v = op == MI355Q_EX_DIV ? (int32_t)(0u - (uint32_t)a) : 0;
    else v = op == MI355Q_EX_DIV ? a / b : a % b;
  } else {
    if (is_null) return 2u;
    const int64_t r = op == MI355Q_EX_ADD ? (int64_t)a + b : op == MI355Q_EX_SUB ? (int64_t)a - b : (int64_t)a * b;
    if (r > (int64_t)INT32_MAX || r < (int64_t)INT32_MIN) {
      err = MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
      return 3u;
    }
    v = (int32_t)r;
  }
  return 0u;
}
MQ_HD uint32_t pair_eval(const PairAtom& pa, int32_t a, int32_t bc, int32_t& err) {  // -> 0 FALSE, 1 TRUE, 2 NULL, 3 ERROR
  const int32_t nul = INT32_MIN;
  const int32_t b = pa.b_is_lit ? pa.b_lit : bc;
  const bool ln = pa.ln != 0, rn = pa.rn != 0;
  int32_t v;
  if (pa.op == 0) {
    if ((ln && a == nul) || (rn && b == nul)) return 2u;
    v = a < b ? -1 : a > b ? 1 : 0;
  } else {
    const uint32_t st = pair_arith(pa.op, a, b, ln, rn, v, err);
    if (st) return st;
    const bool vn = ln || rn;  // (a value that is the NULL pattern, behind a nullable operand: NULL for whatever reads it)
    if (pa.op2) {
      int32_t w;
      const uint32_t st2 = pair_arith(pa.op2, v, pa.lit2, vn, false, w, err);
      if (st2) return st2;
      v = w;
    }
    if (vn && v == nul) return 2u;
  }
  bool in = v >= pa.lo && v <= pa.hi;
  if (pa.negate) in = !in;
  return in ? 1u : 0u;
}
// a compiled program -> its lean form (pa->lean = 0: not one of the shapes); col_type[slot] = the filter columns' types
inline void pair_atom_of(const RegProg& p, const int32_t (&op_col_type)[2], PairAtom* pa) {
  *pa = PairAtom{};
  const RpStep* s = p.step;
  if (p.n_steps < 3 || s[0].kind != RP_LDX_COL || op_col_type[0] != MI355Q_INT32 || s[0].arg != 0) return;
  const bool b_col = s[1].kind == RP_LDY_COL, b_lit = s[1].kind == RP_LDY_LIT;
  if (!b_col && !b_lit) return;
  if (b_col && (op_col_type[1] != MI355Q_INT32 || s[1].arg != 1)) return;
  if (b_lit && (s[1].type != MI355Q_INT32 || s[1].lit > INT32_MAX || s[1].lit < INT32_MIN)) return;
  const RpStep* cmp;
  int64_t c;
  if (p.n_steps == 3 && b_col && s[2].kind == RP_BIN && s[2].op >= MI355Q_EX_EQ && s[2].op <= MI355Q_EX_GE && s[2].arg == MI355Q_INT32) {
    cmp = &s[2];
    pa->op = 0;
    pa->ln = (s[2].flags & EXF_LHS_NULLABLE) != 0;
    pa->rn = (s[2].flags & EXF_RHS_NULLABLE) != 0;
    c = 0;
  } else if (p.n_steps == 5 && s[2].kind == RP_BIN && s[2].op >= MI355Q_EX_ADD && s[2].op <= MI355Q_EX_MOD && s[2].type == MI355Q_INT32 &&
             s[3].kind == RP_LDY_LIT && s[3].type == MI355Q_INT32 && s[4].kind == RP_BIN && s[4].op >= MI355Q_EX_EQ &&
             s[4].op <= MI355Q_EX_GE && s[4].arg == MI355Q_INT32) {
    cmp = &s[4];
    pa->op = s[2].op;
    pa->ln = (s[2].flags & EXF_LHS_NULLABLE) != 0;
    pa->rn = (s[2].flags & EXF_RHS_NULLABLE) != 0;
    // (the comparison's left side is NULL exactly where the arithmetic says so: its flag is the arithmetic's result flag)
    if (((s[4].flags & EXF_LHS_NULLABLE) != 0) != (pa->ln || pa->rn) || (s[4].flags & EXF_RHS_NULLABLE)) return;
    c = s[3].lit;
  } else if (p.n_steps == 7 && s[2].kind == RP_BIN && s[2].op >= MI355Q_EX_ADD && s[2].op <= MI355Q_EX_MOD && s[2].type == MI355Q_INT32 &&
             s[3].kind == RP_LDY_LIT && s[3].type == MI355Q_INT32 && s[3].lit <= INT32_MAX && s[3].lit >= INT32_MIN &&
             s[4].kind == RP_BIN && s[4].op >= MI355Q_EX_ADD && s[4].op <= MI355Q_EX_MOD && s[4].type == MI355Q_INT32 &&
             s[5].kind == RP_LDY_LIT && s[5].type == MI355Q_INT32 && s[6].kind == RP_BIN && s[6].op >= MI355Q_EX_EQ &&
             s[6].op <= MI355Q_EX_GE && s[6].arg == MI355Q_INT32) {
    // a chain: ((a <op> b) <op2> literal) <cmp> literal
    cmp = &s[6];
    pa->op = s[2].op;
    pa->ln = (s[2].flags & EXF_LHS_NULLABLE) != 0;
    pa->rn = (s[2].flags & EXF_RHS_NULLABLE) != 0;
    const bool vn = pa->ln || pa->rn;
    if (((s[4].flags & EXF_LHS_NULLABLE) != 0) != vn || (s[4].flags & EXF_RHS_NULLABLE)) return;
    if (((s[6].flags & EXF_LHS_NULLABLE) != 0) != vn || (s[6].flags & EXF_RHS_NULLABLE)) return;
    pa->op2 = s[4].op;
    pa->lit2 = (int32_t)s[3].lit;
    c = s[5].lit;
  } else {
    return;
  }
  pa->b_is_lit = b_lit;
  pa->b_lit = b_lit ? (int32_t)s[1].lit : 0;
  // the comparison as a range of the value: what make_range_filter states for `value <op> c` at INT32
  DevQual dq{};
  dq.op = cmp->op == MI355Q_EX_EQ ? MI355Q_EQ : cmp->op == MI355Q_EX_NE ? MI355Q_NE : cmp->op == MI355Q_EX_LT ? MI355Q_LT
          : cmp->op == MI355Q_EX_LE ? MI355Q_LE : cmp->op == MI355Q_EX_GT ? MI355Q_GT : MI355Q_GE;
  dq.type = MI355Q_INT32;
  dq.ival = c;
  fast::RangeFilter f;
  if (!fast::make_range_filter(dq, &f)) return;
  pa->lo = (int32_t)f.lo;
  pa->hi = (int32_t)f.hi;
  pa->negate = f.negate;
  pa->lean = 1;
}

struct BoolFilter {
  int32_t n_cols, n_atoms;                         // n_atoms: the RANGE atoms (the program atoms follow them in the state index)
  int32_t col[kBfMaxCols], col_type[kBfMaxCols];  // physical column; MI355Q_INT32 / MI355Q_INT64 / MI355Q_DOUBLE (program operands only)
  int32_t atoms_of_col[kBfMaxCols];               // range atoms are stored grouped by column, in column order
  BoolAtom atom[kBfMaxAtoms];
  uint32_t table[kBfTableWords];                   // one bit per state vector (mixed radix: 3 per range atom, then 3 or 4 per program)
  // ---- program atoms (round 6); everything from here on is only copied into LDS when n_progs != 0
  int32_t n_progs, any_raise;
  int32_t prog_op[kBfMaxProgs][2];                 // the filter column behind operand slot 0 / 1 of each program
  int32_t all_lean, all_i32;                       // every program has a lean form (PairAtom); every filter column is INT32
  PairAtom pair[kBfMaxProgs];
  RegProg prog[kBfMaxProgs];
  uint32_t etable[kBfErrStates / 8];               // any_raise: per state vector 0, or 1 + the program atom whose error is the row's
};
// A kernel receives a POINTER to the filter in device memory and copies it into LDS (a by-value kernel argument indexed
// with a run-time atom number would be lowered to a scratch copy of the whole argument block).

#if defined(__HIPCC__) || defined(HOSTSIM_DEVICE_CODE)
// state of one atom for a value: 0 FALSE, 1 TRUE, 2 NULL
MQ_D uint32_t bf_atom_state(const BoolAtom& a, int64_t v) {
  if (a.nullable && v == a.null_val) return 2u;
  bool in = v >= a.lo && v <= a.hi;
  if (a.negate) in = !in;
  return in ? 1u : 0u;
}
// the filter for one row: vals[c] = the row's value of filter column c (sign-extended); s_table = the table in LDS
template <int NF>
MQ_D bool bf_row_passes(const BoolFilter& bf, const int64_t (&vals)[NF]) {
  uint32_t idx = 0, mul = 1;
  int ai = 0;
#pragma unroll
  for (int c = 0; c < NF; ++c) {
    if (c >= bf.n_cols) break;
    for (int j = 0; j < bf.atoms_of_col[c]; ++j) {
      idx += bf_atom_state(bf.atom[ai], vals[c]) * mul;
      mul *= 3u;
      ++ai;
    }
  }
  return (bf.table[idx >> 5] >> (idx & 31u)) & 1u;
}
// copies the filter from device memory into LDS (call by every thread of the block, then synchronise)
MQ_D void bf_load(const BoolFilter* src, BoolFilter* s_dst, int tid, int block) {
  const uint32_t* a = (const uint32_t*)src;
  uint32_t* b = (uint32_t*)s_dst;
  const uint32_t head = (uint32_t)(offsetof(BoolFilter, n_progs) / 4) + 2u;
  const uint32_t words = src->n_progs ? (uint32_t)(sizeof(BoolFilter) / 4) : head;
  for (uint32_t w = tid; w < words; w += block) b[w] = a[w];
}
// The filter for the FOUR rows of a quad: vals[j][c] = row j's value of filter column c (integers sign-extended, DOUBLE
// as its bits); `valid` = the rows that exist (bit j).  Returns the rows that pass (bit j); *err receives the error a
// VALID row raises (program atoms only), if it holds none yet.
// The programs' typed steps are INLINED here (four rows of every member, 64-bit divisions among them): this is for a kernel
// that holds nothing but the filter's columns — the row-mask pre-pass (kernels_filter.hip).  Measured in round 6: inlined
// into the typed few-groups member it took the member from 52 to 128 registers + scratch (2.66 -> 9.8 ms per 1 B rows on
// the PLAIN shape); behind a non-inlined call the values live across the call site did the same (13.2 ms).
template <int NF>
MQ_D uint32_t bf_quad_pass(const BoolFilter& bf, const int64_t (&vals)[4][NF], uint32_t valid, int32_t* err) {
  uint32_t idx[4] = {0, 0, 0, 0}, mul = 1;
  int ai = 0;
#pragma unroll
  for (int c = 0; c < NF; ++c) {
    if (c >= bf.n_cols) break;
    for (int k = 0; k < bf.atoms_of_col[c]; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j) idx[j] += bf_atom_state(bf.atom[ai], vals[j][c]) * mul;
      mul *= 3u;
      ++ai;
    }
  }
  uint32_t epack[4] = {0, 0, 0, 0};  // two bits per program atom: the error it raised (ex_err_enc)
  const int np = bf.n_progs;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int k = 0; k < np; ++k) {
    // the program's (at most two) operand columns, picked with wave-uniform selects (a run-time index into a register
    // array would be laid out in scratch)
    const int ca = bf.prog_op[k][0], cb = bf.prog_op[k][1];
    int64_t av[4], bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      av[j] = vals[j][0];
      bv[j] = vals[j][0];
#pragma unroll
      for (int c = 1; c < NF; ++c) {
        if (ca == c) av[j] = vals[j][c];
        if (cb == c) bv[j] = vals[j][c];
      }
    }
    int64_t ops[4][2], outv[4];
    int32_t e4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ops[j][0] = av[j];
      ops[j][1] = bv[j];
    }
    rp_eval<4, 2>(bf.prog[k], ops, outv, e4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t st = e4[j] ? 3u : outv[j] == 1 ? 1u : outv[j] == 0 ? 0u : 2u;  // (anything else is the INT8 NULL)
      idx[j] += st * mul;
      epack[j] |= ex_err_enc(e4[j]) << (2 * k);
    }
    mul *= bf.prog[k].can_raise ? 4u : 3u;
  }
  uint32_t pass = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t bit = (bf.table[idx[j] >> 5] >> (idx[j] & 31u)) & 1u;
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
// one row on its own (fragment tails): through the quad form when the filter has program atoms
template <int NF>
MQ_D bool bf_one_row_passes(const BoolFilter& bf, const int64_t (&vals)[NF], int32_t* err) {
  if (bf.n_progs == 0) return bf_row_passes<NF>(bf, vals);
  int64_t qv[4][NF];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < NF; ++c) qv[j][c] = vals[c];
  return bf_quad_pass<NF>(bf, qv, 1u, err) & 1u;
}
#endif

// ---- host side (plan.cpp)
struct BoolFilterHost {
  BoolFilter bf;
  int32_t table_words;
};
// Compiles the whole filter of a plan — its plain quals and its quals `BOOLEAN expression = 1` — into one BoolFilter.
// false: the filter has a shape this form does not state (see above), or needs more atoms / columns than it holds.
// On success *rest receives the plan WITHOUT quals and without the expressions the quals read (group keys / targets
// must not read expressions either: those steps take the projection pass).
bool compile_bool_filter(const mi355q_plan& plan, BoolFilterHost* out, mi355q_plan* rest);

}  // namespace mq
