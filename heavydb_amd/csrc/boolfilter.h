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
// Filters with arithmetic, comparisons of two columns, CASE, narrowing casts or floating-point leaves are not
// taken (the step then runs through the projection pass, kernels_generic.hip k_project).
#pragma once

#include "dev_common.h"
#include "fast_common.h"

namespace mq {

constexpr int kBfMaxAtoms = 8;
constexpr int kBfMaxCols = 4;
constexpr int kBfTableWords = 206;  // ceil(3^8 / 32)

struct BoolAtom {
  int64_t lo, hi, null_val;
  int32_t negate, nullable;  // nullable: the column can hold null_val, and the atom is then NULL (IS NULL atoms: 0)
};
struct BoolFilter {
  int32_t n_cols, n_atoms;
  int32_t col[kBfMaxCols], col_type[kBfMaxCols];  // physical column; MI355Q_INT32 / MI355Q_INT64
  int32_t atoms_of_col[kBfMaxCols];               // atoms are stored grouped by column, in column order
  BoolAtom atom[kBfMaxAtoms];
  uint32_t table[kBfTableWords];                   // 3^n_atoms bits
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
  for (uint32_t w = tid; w < sizeof(BoolFilter) / 4; w += block) b[w] = a[w];
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
