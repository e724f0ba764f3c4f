// regprog.h — expressions compiled at plan time into TWO-REGISTER programs of typed steps (round 6).
//
// The reference compiles the expressions of a step into its row function (Executor::compileBody, NativeCodegen.cpp:3455;
// codegenArith / codegenAdd / codegenMul / codegenDiv, ArithmeticIR.cpp:39-431; codegenCmp, CompareIR.cpp:230-330;
// codegenCast, CastIR.cpp).  This library has no run-time compiler (north_star: "No LLVM/NVVM JIT ... a fixed family of
// parametric HIP kernels selected at plan time"); what it has instead, for the expression shapes real filters and targets
// are made of — `a / b > 3`, `x + y > 100`, `a < b`, `CAST(x AS BIGINT) * 1000`, `f * 2.0` — is this form:
//   * a postfix program whose evaluation stack never holds more than TWO values (every left-deep chain of arithmetic,
//     casts and one comparison over columns and literals) is a sequence of at most kRpMaxSteps steps over two registers
//     x (the value below) and y (the top): load a column / a literal into x or y, a unary step on either, a binary step
//     x = x <op> y;
//   * every step is one of a FIXED set of typed members — one instantiation of the same ex_* function of expr.h that the
//     interpreter and the oracle-checked host evaluator run, with the operation and the type compiled in (the node's
//     nullability flags stay wave-uniform run-time values) — picked by a wave-uniform switch ONCE per step and J rows of a
//     lane: no stack in LDS or scratch, no node decode per row, the semantics (NULL rules, error 7 / error 1, the
//     evaluation order of the checks) stated once, in expr.h.
// Consumers: the atoms of a filter compiled at plan time (boolfilter.h: an atom is then the BOOLEAN such a program leaves;
// evaluated by the row-mask pre-pass, kernels_filter.hip), the typed expression targets of the Projection family
// (kernels_proj.hip k_proj_fast).
// Programs that do not fit (CASE, more than two live values, INT8 / INT16 / FLOAT arithmetic, encoded columns, values of
// earlier expressions) keep the interpreter (expr.h eval_expr_rows).
#pragma once

#include "dev_common.h"
#include "expr.h"

namespace mq {

constexpr int kRpMaxSteps = 8;
enum : int32_t {
  RP_END = 0,
  RP_LDX_COL = 1,  // x = the consumer's operand `arg` (a column's value as ex_col gives it: integers sign-extended, DOUBLE as its bits)
  RP_LDY_COL = 2,
  RP_LDX_LIT = 3,  // x = lit (ex_lit's pattern)
  RP_LDY_LIT = 4,
  RP_UNX = 5,      // x = <op>(x): EX_CAST / EX_UMINUS / EX_IS_NULL / EX_NOT
  RP_UNY = 6,
  RP_BIN = 7       // x = x <op> y: EX_ADD .. EX_MOD, EX_EQ .. EX_GE; y is consumed
};
struct RpStep {
  int32_t kind;      // RP_*
  int32_t op;        // mi355q_expr_op of a unary / binary step
  int32_t type;      // the node's result type
  int32_t arg;       // RP_LD?_COL: operand slot; casts / comparisons / IS NULL: the operand's type (DevExprNode::arg)
  int32_t flags;     // EXF_* of the node
  uint32_t packed;   // kind | op << 4 | type << 9 | arg << 12 | flags << 15: what the device reads (one word, one readfirstlane)
  int64_t lit;       // RP_LD?_LIT
};
struct RegProg {
  int32_t n_steps;
  int32_t type;       // plain type of the value x holds at the end
  int32_t nullable;
  int32_t can_raise;  // some step can raise error 7 / 1 (arithmetic, narrowing or floating-point casts, unary minus)
  int32_t has_divmod, n_ops;  // a division / modulo step; operand slots the program reads (1 or 2: slots 0 and 1)
  RpStep step[kRpMaxSteps];
};

MQ_HD constexpr bool rp_type_ok(int t) { return t == MI355Q_INT32 || t == MI355Q_INT64 || t == MI355Q_DOUBLE; }

// ---- host: nodes [first, last] of a lowered expression -> a program.  `slot_of(column)` maps a physical column to the
// consumer's operand slot (< 0: the consumer has no room for it).  false: the shape is not one this form states.
template <typename SlotOf>
inline bool rp_compile(const DevExpr& e, int first, int last, int n_phys_cols, SlotOf&& slot_of, RegProg* out) {
  RegProg& p = *out;
  p = RegProg{};
  int depth = 0;
  int types[2] = {0, 0};
  bool nulls[2] = {false, false};
  auto push_step = [&](const RpStep& s) -> bool {
    if (p.n_steps >= kRpMaxSteps) return false;
    p.step[p.n_steps++] = s;
    return true;
  };
  for (int i = first; i <= last; ++i) {
    const DevExprNode& n = e.nodes[i];
    RpStep s{};
    s.op = n.op;
    s.type = n.type;
    s.arg = n.arg;
    s.flags = n.flags & (EXF_NULLABLE | EXF_LHS_NULLABLE | EXF_RHS_NULLABLE);
    switch (n.op) {
      case MI355Q_EX_COL: {
        if (depth >= 2 || n.arg < 0 || n.arg >= n_phys_cols) return false;  // (the value of an earlier expression: the interpreter)
        const int code = (int)n.ilit;
        if (!rp_type_ok(code) || n.type != code) return false;  // plain 4- / 8-byte integer or DOUBLE chunks
        const int slot = slot_of(n.arg);
        if (slot < 0) return false;
        s.kind = depth == 0 ? RP_LDX_COL : RP_LDY_COL;
        s.arg = slot;
        types[depth] = n.type;
        nulls[depth] = (n.flags & EXF_NULLABLE) != 0;
        ++depth;
        break;
      }
      case MI355Q_EX_LIT:
        if (depth >= 2 || !(rp_type_ok(n.type) || n.type == MI355Q_INT8 || n.type == MI355Q_INT16)) return false;
        s.kind = depth == 0 ? RP_LDX_LIT : RP_LDY_LIT;
        s.lit = ex_lit(n);
        types[depth] = n.type;
        nulls[depth] = n.arg != 0;
        ++depth;
        break;
      case MI355Q_EX_CAST:
        if (depth < 1 || !rp_type_ok(n.type)) return false;
        // (a literal of a narrow type cast up is fine; a narrow COLUMN never gets here)
        if (!(rp_type_ok(n.arg) || n.arg == MI355Q_INT8 || n.arg == MI355Q_INT16)) return false;
        s.kind = depth == 1 ? RP_UNX : RP_UNY;
        if ((ex_is_int(n.arg) && ex_is_int(n.type) && plain_width(n.type) < plain_width(n.arg)) || (!ex_is_int(n.arg) && ex_is_int(n.type)))
          p.can_raise = 1;  // a narrowing cast raises error 7; (floating point to integer: undefined beyond the type, never an error)
        types[depth - 1] = n.type;
        break;
      case MI355Q_EX_UMINUS:
        if (depth < 1 || !rp_type_ok(n.type)) return false;
        s.kind = depth == 1 ? RP_UNX : RP_UNY;
        if (ex_is_int(n.type)) p.can_raise = 1;
        break;
      case MI355Q_EX_IS_NULL:
        if (depth < 1 || !rp_type_ok(n.arg)) return false;
        s.kind = depth == 1 ? RP_UNX : RP_UNY;
        types[depth - 1] = MI355Q_INT8;
        nulls[depth - 1] = false;
        break;
      case MI355Q_EX_NOT:
        if (depth < 1 || types[depth - 1] != MI355Q_INT8) return false;
        s.kind = depth == 1 ? RP_UNX : RP_UNY;
        break;
      case MI355Q_EX_ADD: case MI355Q_EX_SUB: case MI355Q_EX_MUL: case MI355Q_EX_DIV: case MI355Q_EX_MOD:
        if (depth != 2 || !rp_type_ok(n.type) || (n.op == MI355Q_EX_MOD && n.type == MI355Q_DOUBLE)) return false;
        s.kind = RP_BIN;
        if (ex_is_int(n.type) || n.op == MI355Q_EX_DIV) p.can_raise = 1;
        if (n.op == MI355Q_EX_DIV || n.op == MI355Q_EX_MOD) p.has_divmod = 1;
        depth = 1;
        types[0] = n.type;
        nulls[0] = (n.flags & EXF_NULLABLE) != 0;
        break;
      case MI355Q_EX_EQ: case MI355Q_EX_NE: case MI355Q_EX_LT: case MI355Q_EX_LE: case MI355Q_EX_GT: case MI355Q_EX_GE:
        if (depth != 2 || !rp_type_ok(n.arg)) return false;
        s.kind = RP_BIN;
        depth = 1;
        types[0] = MI355Q_INT8;
        nulls[0] = (n.flags & EXF_NULLABLE) != 0;
        break;
      default:
        return false;  // CASE, AND / OR (a filter's logic lives in its truth table)
    }
    if (!push_step(s)) return false;
  }
  if (depth != 1 || p.n_steps < 1) return false;
  for (int i = 0; i < p.n_steps; ++i) {
    RpStep& s = p.step[i];
    s.packed = (uint32_t)s.kind | ((uint32_t)s.op << 4) | ((uint32_t)s.type << 9) | ((uint32_t)s.arg << 12) | ((uint32_t)s.flags << 15);
  }
  p.type = types[0];
  p.nullable = nulls[0] ? 1 : 0;
  return true;
}

// ---- the typed members.  OP / T are compile-time constants: the compiler folds every decision ex_* takes on them.
template <int OP, int T>
MQ_HD int64_t rp_binary(int flags, int64_t a, int64_t b, int32_t& ev) {
  DevExprNode n{};
  n.op = OP;
  n.flags = flags;
  if constexpr (OP >= MI355Q_EX_EQ && OP <= MI355Q_EX_GE) {
    n.type = MI355Q_INT8;
    n.arg = T;
    return ex_cmp(n, a, b);
  } else if constexpr (OP == MI355Q_EX_DIV || OP == MI355Q_EX_MOD) {
    n.type = T;
    return ex_divmod(n, a, b, ev);
  } else {
    n.type = T;
    return ex_arith(n, a, b, ev);
  }
}
template <int FROM, int TO>
MQ_HD int64_t rp_cast(int flags, int64_t v, int32_t& ev) {
  DevExprNode n{};
  n.op = MI355Q_EX_CAST;
  n.arg = FROM;
  n.type = TO;
  n.flags = flags;
  return ex_cast(n, v, ev);
}

// one step's operation on J rows; `s` is wave-uniform (an LDS or scalar copy of the step)
// (the rows of one step run one AFTER the other: left to itself the scheduler interleaves the J copies of a member — four
// 64-bit divisions at once — and the kernel around it pays with its registers: k_filter_mask 172 VGPRs, two waves per SIMD)
#if defined(__HIP_DEVICE_COMPILE__)
#define RP_ROW_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define RP_ROW_FENCE ((void)0)
#endif
#define RP_ROWS(expr)                        \
  _Pragma("unroll") for (int j = 0; j < J; ++j) { expr; RP_ROW_FENCE; }
template <int J>
MQ_HD void rp_apply_unary(const RpStep& s, int64_t (&v)[J], int32_t (&ev)[J]) {
  const int fl = s.flags;
  switch (s.op) {
    case MI355Q_EX_CAST: {
      const int from = s.arg, to = s.type;
#define RP_CAST_CASE(F, T)                                   \
  if (from == F && to == T) {                                \
    RP_ROWS(v[j] = (rp_cast<F, T>(fl, v[j], ev[j])))         \
    return;                                                  \
  }
      RP_CAST_CASE(MI355Q_INT32, MI355Q_INT64)
      RP_CAST_CASE(MI355Q_INT32, MI355Q_DOUBLE)
      RP_CAST_CASE(MI355Q_INT64, MI355Q_DOUBLE)
      RP_CAST_CASE(MI355Q_INT64, MI355Q_INT32)
      RP_CAST_CASE(MI355Q_DOUBLE, MI355Q_INT32)
      RP_CAST_CASE(MI355Q_DOUBLE, MI355Q_INT64)
      RP_CAST_CASE(MI355Q_INT32, MI355Q_INT32)
      RP_CAST_CASE(MI355Q_INT64, MI355Q_INT64)
      RP_CAST_CASE(MI355Q_DOUBLE, MI355Q_DOUBLE)
#undef RP_CAST_CASE
      {  // a narrow literal cast up (INT8 / INT16 -> ...): the flat code on the step itself
        DevExprNode n{};
        n.op = MI355Q_EX_CAST;
        n.arg = from;
        n.type = to;
        n.flags = fl;
        RP_ROWS(v[j] = ex_cast(n, v[j], ev[j]))
      }
      return;
    }
    case MI355Q_EX_UMINUS: {
      DevExprNode n{};
      n.op = MI355Q_EX_UMINUS;
      n.flags = fl;
      if (s.type == MI355Q_INT32) { n.type = MI355Q_INT32; RP_ROWS(v[j] = ex_uminus(n, v[j], ev[j])) }
      else if (s.type == MI355Q_INT64) { n.type = MI355Q_INT64; RP_ROWS(v[j] = ex_uminus(n, v[j], ev[j])) }
      else { n.type = MI355Q_DOUBLE; RP_ROWS(v[j] = ex_uminus(n, v[j], ev[j])) }
      return;
    }
    case MI355Q_EX_IS_NULL: {
      DevExprNode n{};
      n.op = MI355Q_EX_IS_NULL;
      n.flags = fl;
      if (s.arg == MI355Q_INT32) { n.arg = MI355Q_INT32; RP_ROWS(v[j] = ex_is_null(n, v[j], ev[j])) }
      else if (s.arg == MI355Q_INT64) { n.arg = MI355Q_INT64; RP_ROWS(v[j] = ex_is_null(n, v[j], ev[j])) }
      else { n.arg = MI355Q_DOUBLE; RP_ROWS(v[j] = ex_is_null(n, v[j], ev[j])) }
      return;
    }
    default: {  // MI355Q_EX_NOT
      DevExprNode n{};
      n.op = MI355Q_EX_NOT;
      n.flags = fl;
      RP_ROWS(v[j] = ex_not(n, v[j]))
    }
  }
}
// DIVMOD = false: a consumer whose programs never divide (the division members are the large ones: four rows of a 64-bit
// division inlined into a streaming kernel cost it its registers)
template <int J, bool DIVMOD>
MQ_HD void rp_apply_binary(const RpStep& s, int64_t (&x)[J], const int64_t (&y)[J], int32_t (&ex)[J], const int32_t (&ey)[J]) {
  const int fl = s.flags;
  RP_ROWS(if (!ex[j]) ex[j] = ey[j])  // (lhs first, then rhs, then this operation)
  const int t = s.op >= MI355Q_EX_EQ && s.op <= MI355Q_EX_GE ? s.arg : s.type;
#define RP_BIN_T(OP)                                                                                   \
  case OP:                                                                                             \
    if (t == MI355Q_INT32) { RP_ROWS(x[j] = (rp_binary<OP, MI355Q_INT32>(fl, x[j], y[j], ex[j]))) }    \
    else if (t == MI355Q_INT64) { RP_ROWS(x[j] = (rp_binary<OP, MI355Q_INT64>(fl, x[j], y[j], ex[j]))) } \
    else { RP_ROWS(x[j] = (rp_binary<OP, MI355Q_DOUBLE>(fl, x[j], y[j], ex[j]))) }                     \
    break;
  switch (s.op) {
    RP_BIN_T(MI355Q_EX_ADD)
    RP_BIN_T(MI355Q_EX_SUB)
    RP_BIN_T(MI355Q_EX_MUL)
    RP_BIN_T(MI355Q_EX_EQ)
    RP_BIN_T(MI355Q_EX_NE)
    RP_BIN_T(MI355Q_EX_LT)
    RP_BIN_T(MI355Q_EX_LE)
    RP_BIN_T(MI355Q_EX_GT)
    RP_BIN_T(MI355Q_EX_GE)
    default:
      if constexpr (DIVMOD) {
        if (s.op == MI355Q_EX_DIV) {
          if (t == MI355Q_INT32) { RP_ROWS(x[j] = (rp_binary<MI355Q_EX_DIV, MI355Q_INT32>(fl, x[j], y[j], ex[j]))) }
          else if (t == MI355Q_INT64) { RP_ROWS(x[j] = (rp_binary<MI355Q_EX_DIV, MI355Q_INT64>(fl, x[j], y[j], ex[j]))) }
          else { RP_ROWS(x[j] = (rp_binary<MI355Q_EX_DIV, MI355Q_DOUBLE>(fl, x[j], y[j], ex[j]))) }
        } else {  // MI355Q_EX_MOD: integers only
          if (t == MI355Q_INT32) { RP_ROWS(x[j] = (rp_binary<MI355Q_EX_MOD, MI355Q_INT32>(fl, x[j], y[j], ex[j]))) }
          else { RP_ROWS(x[j] = (rp_binary<MI355Q_EX_MOD, MI355Q_INT64>(fl, x[j], y[j], ex[j]))) }
        }
      }
  }
#undef RP_BIN_T
}

// The program for J rows of a lane.  vals[j][c] = operand slot c of row j; out[j] = the value (ex_wrap_int'ed to the
// program's type), err[j] = 0 or the first error row j met in evaluation order (its value is then unspecified).
// `p` should live in LDS (or constant memory): every field read is wave-uniform.
template <int J, int NC, bool DIVMOD = true>
MQ_HD void rp_eval(const RegProg& p, const int64_t (&vals)[J][NC], int64_t (&out)[J], int32_t (&err)[J]) {
  int64_t x[J], y[J];
  int32_t ex[J], ey[J];
  RP_ROWS(x[j] = 0; y[j] = 0; ex[j] = 0; ey[j] = 0)
#if defined(__HIP_DEVICE_COMPILE__)
  const int ns = __builtin_amdgcn_readfirstlane(p.n_steps);
#else
  const int ns = p.n_steps;
#endif
  // a step is read ONE AHEAD of its use (an LDS round trip per step otherwise parks the wave: the steps of a program are a
  // dependent chain), as one packed word + the literal
  uint32_t pk_next = p.step[0].packed;
  int64_t lit_next = p.step[0].lit;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int i = 0; i < ns; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t pk = (uint32_t)__builtin_amdgcn_readfirstlane((int)pk_next);
#else
    const uint32_t pk = pk_next;
#endif
    RpStep s;
    s.lit = lit_next;
    if (i + 1 < ns) {
      pk_next = p.step[i + 1].packed;
      lit_next = p.step[i + 1].lit;
    }
    s.kind = (int32_t)(pk & 15u);
    s.op = (int32_t)((pk >> 4) & 31u);
    s.type = (int32_t)((pk >> 9) & 7u);
    s.arg = (int32_t)((pk >> 12) & 7u);
    s.flags = (int32_t)((pk >> 15) & 15u);
    switch (s.kind) {
      case RP_LDX_COL:
      case RP_LDY_COL: {
        int64_t v[J];
        // (a run-time index into a register array would be laid out in scratch: a chain of uniform selects instead)
        RP_ROWS(v[j] = vals[j][0])
#pragma unroll
        for (int c = 1; c < NC; ++c)
          if (s.arg == c) { RP_ROWS(v[j] = vals[j][c]) }
        if (s.kind == RP_LDX_COL) { RP_ROWS(x[j] = v[j]; ex[j] = 0) }
        else { RP_ROWS(y[j] = v[j]; ey[j] = 0) }
        break;
      }
      case RP_LDX_LIT: RP_ROWS(x[j] = s.lit; ex[j] = 0) break;
      case RP_LDY_LIT: RP_ROWS(y[j] = s.lit; ey[j] = 0) break;
      case RP_UNX:
      case RP_UNY: {  // ONE copy of the unary members: a step on y swaps the registers around it
        const bool on_y = s.kind == RP_UNY;
        if (on_y) {
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const int64_t tv = x[j]; x[j] = y[j]; y[j] = tv;
            const int32_t te = ex[j]; ex[j] = ey[j]; ey[j] = te;
          }
        }
        rp_apply_unary<J>(s, x, ex);
        if (on_y) {
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const int64_t tv = x[j]; x[j] = y[j]; y[j] = tv;
            const int32_t te = ex[j]; ex[j] = ey[j]; ey[j] = te;
          }
        }
        break;
      }
      default: rp_apply_binary<J, DIVMOD>(s, x, y, ex, ey);
    }
  }
  RP_ROWS(out[j] = ex_wrap_int(p.type, x[j]); err[j] = ex[j])
}
#undef RP_ROWS
#undef RP_ROW_FENCE

}  // namespace mq
