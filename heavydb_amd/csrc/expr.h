// expr.h — evaluator of the projected-expression micro-ops (mi355q_expr lowered to DevExpr by plan.cpp).
//
// One function, used by the projection kernel (kernels_generic.hip k_project) on the device and by the
// host emulation of tests/emu.  Values travel as 64-bit patterns: integers sign-extended (SQL NULL = the
// inline sentinel of the node's type, Shared/InlineNullValues.h), DOUBLE as its bits, FLOAT as its bits in
// the low word.  Reference semantics restated (heavyai/heavydb):
//   casts        CastIR.cpp:424-495 codegenCastBetweenIntTypes (+ :497-553 the narrowing check: error when
//                v > max(to) or v <= min(to), NULL exempt), :555-594 codegenCastToFp, :596-653
//                codegenCastFromFp; cast_<a>_to_<b>_nullable and DEF_ROUND_NULLABLE,
//                RuntimeFunctions.cpp:262-330
//   + - *        ArithmeticIR.cpp:39-75 codegenArith; integers: :861-909 codegenBinOpWithOverflowForCPU
//                (sadd/ssub/smul.with.overflow at the operand type's width, NULL operands skip the check and
//                give NULL: codegenSkipOverflowCheckForNull); floating point: add_/sub_/mul_<type>_nullable,
//                RuntimeFunctions.cpp:46-53
//   / %          ArithmeticIR.cpp:431-560 codegenDiv (zero check skipped behind a NULL operand), :731-760 codegenMod (zero
//                check first); div_/mod_<type>_nullable[_lhs|_rhs], RuntimeFunctions.cpp:46-71
//   comparisons  CompareIR.cpp:230-330 codegenCmp: icmp / fcmp, or <op>_<type>_nullable[_lhs|_rhs] (RuntimeFunctions.cpp:73-107):
//                BOOLEAN 1 / 0, the INT8 NULL when a nullable operand is NULL
//   CASE         CaseIR.cpp:67-140 codegenCase: the THEN value where the condition is TRUE (toBool: NULL is not), else the ELSE
//                value; each branch is its own basic block, so a check that fires in the branch NOT taken does not exist —
//                every value on the stack carries the error its computation met, and CASE keeps the taken branch's only
//   NOT AND OR   LogicalIR.cpp:299-379 codegenLogical: toBool operands (NOT NULL) or logical_not / logical_and / logical_or
//                (RuntimeFunctions.cpp:331-358); :197-297 codegenLogicalShortCircuit where an operand holds an unsafe division:
//                the first operand decides where it can and the second one's basic block is then never entered
//   IS NULL      LogicalIR.cpp:381-432 codegenIsNull: constant false for a NOT NULL operand (which is not evaluated), else the
//                comparison with the type's inline NULL (FCMP_OEQ for doubles / floats)
//   unary minus  ArithmeticIR.cpp:787-838 codegenUMinus: error 7 for the type's minimum as a value, NULL stays NULL
//                (uminus_<type>_nullable, RuntimeFunctions.cpp:247-258)
//   error        ErrorCode::OVERFLOW_OR_UNDERFLOW = 7, DIV_BY_ZERO = 1 (QueryEngine/enums.h:30-51)
#pragma once

#include "dev_common.h"

namespace mq {

MQ_HD bool ex_is_int(int t) { return t >= MI355Q_INT8 && t <= MI355Q_INT64; }
MQ_HD int64_t ex_int_max(int t) {
  return t == MI355Q_INT8 ? (int64_t)INT8_MAX : t == MI355Q_INT16 ? (int64_t)INT16_MAX
         : t == MI355Q_INT32 ? (int64_t)INT32_MAX : INT64_MAX;
}
MQ_HD int64_t ex_int_min(int t) { return plain_int_null(t); }  // the type's minimum IS its NULL sentinel
MQ_HD int64_t ex_flt_pattern(float f) { return (int64_t)(uint32_t)flt_bits(f); }
MQ_HD float ex_flt_of(int64_t v) { return bits_flt((int32_t)(uint32_t)v); }

// ---- one function per micro-op: pure on 64-bit patterns (+ the error a value carries).  Both evaluators below are
// flat loops over these, so the row-at-a-time form (host, emulation, rare paths) and the rows-in-LDS form of the kernels
// state the reference's semantics ONCE.
MQ_HD int64_t ex_col(const DevExprNode& n, const int8_t* const* cols, int64_t pos) {
  const int8_t* c = cols[n.arg];
  if (n.type == MI355Q_DOUBLE) return *(const int64_t*)(c + pos * 8);
  if (n.type == MI355Q_FLOAT) return (int64_t)*(const uint32_t*)(c + pos * 4);
  return decode_int(c, (int)n.ilit, pos);
}
MQ_HD int64_t ex_lit(const DevExprNode& n) {
  return n.arg ? n.ilit  // (the NULL literal: its pattern was laid down by the lowering)
         : n.type == MI355Q_DOUBLE ? dbl_bits(n.flit)
         : n.type == MI355Q_FLOAT ? ex_flt_pattern((float)n.flit) : n.ilit;
}
// The VALUE of an expression, as anything outside it sees it (a target, a qual, a later expression), is a value of its
// type: where a floating-point operand was cast to a narrower integer it does not fit (undefined in the reference:
// fptosi), the type's low bits, sign-extended, remain — what the dense temporary column of the projection pass stores
MQ_HD int64_t ex_wrap_int(int t, int64_t r) {
  return t == MI355Q_INT8 ? (int64_t)(int8_t)r : t == MI355Q_INT16 ? (int64_t)(int16_t)r : t == MI355Q_INT32 ? (int64_t)(int32_t)r : r;
}
MQ_HD int64_t ex_cast(const DevExprNode& n, int64_t v, int32_t& ev) {
  const int from = n.arg, to = n.type;
  const bool nullable = (n.flags & EXF_LHS_NULLABLE) != 0;
  int64_t r = v;
  if (ex_is_int(from)) {
    const bool is_null = nullable && v == plain_int_null(from);
    if (ex_is_int(to)) {
      if (is_null) {
        r = plain_int_null(to);
      } else if (plain_width(to) < plain_width(from) && (v > ex_int_max(to) || v <= ex_int_min(to))) {
        if (!ev) ev = MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
      }
    } else if (to == MI355Q_DOUBLE) {
      r = is_null ? kNullDoubleBits : dbl_bits((double)v);
    } else {
      r = is_null ? (int64_t)(uint32_t)kNullFloatBits : ex_flt_pattern((float)v);
    }
  } else if (from == MI355Q_DOUBLE) {
    const double d = bits_dbl(v);
    const bool is_null = nullable && d == kNullDouble;
    if (to == MI355Q_FLOAT) r = is_null ? (int64_t)(uint32_t)kNullFloatBits : ex_flt_pattern((float)d);
    else if (ex_is_int(to)) r = is_null ? plain_int_null(to) : (int64_t)(d + (d < 0.0 ? -0.5 : 0.5));
  } else {  // FLOAT
    const float f = ex_flt_of(v);
    const bool is_null = nullable && f == kNullFloat;
    if (to == MI355Q_DOUBLE) r = is_null ? kNullDoubleBits : dbl_bits((double)f);
    else if (ex_is_int(to)) r = is_null ? plain_int_null(to) : (int64_t)(f + (f < 0.0f ? -0.5f : 0.5f));
  }
  return r;
}
// ev: the error the operands carry (lhs first, then rhs) on entry; this operation's check is added behind them
MQ_HD int64_t ex_divmod(const DevExprNode& n, int64_t a, int64_t b, int32_t& ev) {
  const int t = n.type;
  const bool ln = (n.flags & EXF_LHS_NULLABLE) != 0, rn = (n.flags & EXF_RHS_NULLABLE) != 0;
  int64_t r;
  if (ex_is_int(t)) {
    const int64_t nul = plain_int_null(t);
    // DIV: a NULL pattern in EITHER operand skips the zero check as soon as one of them may be NULL; MOD tests first
    const bool skip = n.op == MI355Q_EX_DIV && (ln || rn) && (a == nul || b == nul);
    if (!skip && b == 0) {
      if (!ev) ev = MI355Q_ERR_DIV_BY_ZERO;
      r = nul;
    } else if ((ln && a == nul) || (rn && b == nul)) {
      r = nul;
    } else if (b == 0) {
      r = nul;                        // (INT_MIN / 0 behind the skip: undefined in the reference, never trapped here)
    } else if (b == -1) {
      r = n.op == MI355Q_EX_DIV ? (int64_t)(0 - (uint64_t)a) : 0;  // INT_MIN / -1 wraps instead of trapping
    } else if (t != MI355Q_INT64) {   // operands of a narrower type are sign-extended 32-bit values: a 32-bit division
      const int32_t a32 = (int32_t)a, b32 = (int32_t)b;
      r = n.op == MI355Q_EX_DIV ? (int64_t)(a32 / b32) : (int64_t)(a32 % b32);
    } else {
      r = n.op == MI355Q_EX_DIV ? a / b : a % b;
    }
    if (t != MI355Q_INT64) r = t == MI355Q_INT8 ? (int64_t)(int8_t)r : t == MI355Q_INT16 ? (int64_t)(int16_t)r : (int64_t)(int32_t)r;
  } else if (t == MI355Q_DOUBLE) {
    const double x = bits_dbl(a), y = bits_dbl(b);
    const bool skip = (ln || rn) && (x == kNullDouble || y == kNullDouble);
    if (!skip && !(y < 0.0 || y > 0.0) && !ev) ev = MI355Q_ERR_DIV_BY_ZERO;
    r = ((ln && x == kNullDouble) || (rn && y == kNullDouble)) ? kNullDoubleBits : dbl_bits(x / y);
  } else {
    const float x = ex_flt_of(a), y = ex_flt_of(b);
    const bool skip = (ln || rn) && (x == kNullFloat || y == kNullFloat);
    if (!skip && !(y < 0.0f || y > 0.0f) && !ev) ev = MI355Q_ERR_DIV_BY_ZERO;
    r = ((ln && x == kNullFloat) || (rn && y == kNullFloat)) ? (int64_t)(uint32_t)kNullFloatBits : ex_flt_pattern(x / y);
  }
  return r;
}
MQ_HD int64_t ex_cmp(const DevExprNode& n, int64_t a, int64_t b) {
  const int t = n.arg;  // the operands' type
  const bool ln = (n.flags & EXF_LHS_NULLABLE) != 0, rn = (n.flags & EXF_RHS_NULLABLE) != 0;
  // lt / eq / gt as the C operators of the runtime functions give them (a NaN operand: all three false, so that
  // <, <=, >, >=, = are false and <> is true)
  bool is_null, lt, eq, gt;
  if (ex_is_int(t)) {
    const int64_t nul = plain_int_null(t);
    is_null = (ln && a == nul) || (rn && b == nul);
    lt = a < b;
    eq = a == b;
    gt = a > b;
  } else if (t == MI355Q_DOUBLE) {
    const double x = bits_dbl(a), y = bits_dbl(b);
    is_null = (ln && x == kNullDouble) || (rn && y == kNullDouble);
    lt = x < y;
    eq = x == y;
    gt = x > y;
  } else {
    const float x = ex_flt_of(a), y = ex_flt_of(b);
    is_null = (ln && x == kNullFloat) || (rn && y == kNullFloat);
    lt = x < y;
    eq = x == y;
    gt = x > y;
  }
  const bool v = n.op == MI355Q_EX_EQ ? eq : n.op == MI355Q_EX_NE ? !eq : n.op == MI355Q_EX_LT ? lt
                 : n.op == MI355Q_EX_LE ? (lt || eq) : n.op == MI355Q_EX_GT ? gt : (gt || eq);
  return is_null ? plain_int_null(MI355Q_INT8) : (v ? 1 : 0);
}
MQ_HD int64_t ex_not(const DevExprNode& n, int64_t v) {
  if ((n.flags & EXF_LHS_NULLABLE) && v == plain_int_null(MI355Q_INT8)) return v;  // logical_not: NULL stays NULL
  return (n.flags & EXF_LHS_NULLABLE) ? (v ? 0 : 1) : (v > 0 ? 0 : 1);             // (toBool on the NOT NULL side)
}
// ea: the first operand's error on entry, the result's on return; eb: the second operand's
MQ_HD int64_t ex_logic(const DevExprNode& n, int64_t a, int64_t b, int32_t& ea, int32_t eb) {
  const int64_t nul = plain_int_null(MI355Q_INT8);
  const bool is_or = n.op == MI355Q_EX_OR, nullable = (n.flags & EXF_NULLABLE) != 0;
  int64_t r;
  if (n.flags & EXF_SHORT_CIRCUIT) {
    // the first operand alone where it decides — the second one's checks then do not exist
    if (nullable && a == nul) r = nul;
    else if (a == (is_or ? 1 : 0)) r = a;
    else {
      if (!ea) ea = eb;
      r = b;
    }
  } else {
    if (!ea) ea = eb;
    if (!nullable) r = is_or ? (a > 0 || b > 0) : (a > 0 && b > 0);
    else if (a == nul) r = is_or ? (b == 0 ? nul : b) : (b == 0 ? b : nul);  // logical_or / logical_and
    else if (b == nul) r = is_or ? (a == 0 ? nul : a) : (a == 0 ? a : nul);
    else r = is_or ? (a || b) : (a && b);
  }
  return r;
}
MQ_HD int64_t ex_is_null(const DevExprNode& n, int64_t v, int32_t& ev) {
  const int t = n.arg;  // the operand's type
  if (!(n.flags & EXF_LHS_NULLABLE)) {  // a NOT NULL operand is never evaluated: constant false, no check of it exists
    ev = 0;
    return 0;
  }
  return ex_is_int(t) ? v == plain_int_null(t) : t == MI355Q_DOUBLE ? bits_dbl(v) == kNullDouble : ex_flt_of(v) == kNullFloat;
}
MQ_HD int64_t ex_uminus(const DevExprNode& n, int64_t v, int32_t& ev) {
  const int t = n.type;
  const bool nullable = (n.flags & EXF_LHS_NULLABLE) != 0;
  if (ex_is_int(t)) {
    if (v == plain_int_null(t)) {  // the type's minimum: NULL stays NULL, a value cannot be negated
      if (!nullable && !ev) ev = MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
      return v;
    }
    return -v;
  }
  if (t == MI355Q_DOUBLE) {
    const double x = bits_dbl(v);
    return (nullable && x == kNullDouble) ? v : dbl_bits(-x);
  }
  const float x = ex_flt_of(v);
  return (nullable && x == kNullFloat) ? v : ex_flt_pattern(-x);
}
// MI355Q_EX_ADD / _SUB / _MUL; ev as in ex_divmod
MQ_HD int64_t ex_arith(const DevExprNode& n, int64_t a, int64_t b, int32_t& ev) {
  const int t = n.type;
  int64_t r;
  if (ex_is_int(t)) {
    const int64_t nul = plain_int_null(t);
    if (((n.flags & EXF_LHS_NULLABLE) && a == nul) || ((n.flags & EXF_RHS_NULLABLE) && b == nul)) {
      r = nul;
    } else {
      bool ovf;
      long long w;
      if (n.op == MI355Q_EX_ADD) ovf = __builtin_add_overflow((long long)a, (long long)b, &w);
      else if (n.op == MI355Q_EX_SUB) ovf = __builtin_sub_overflow((long long)a, (long long)b, &w);
      else ovf = __builtin_mul_overflow((long long)a, (long long)b, &w);
      r = (int64_t)w;
      if (t != MI355Q_INT64) {  // narrower operands cannot wrap 64 bits (|a|, |b| <= 2^31)
        ovf = r > ex_int_max(t) || r < ex_int_min(t);
        r = t == MI355Q_INT8 ? (int64_t)(int8_t)r : t == MI355Q_INT16 ? (int64_t)(int16_t)r : (int64_t)(int32_t)r;
      }
      if (ovf && !ev) ev = MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
    }
  } else if (t == MI355Q_DOUBLE) {
    const double x = bits_dbl(a), y = bits_dbl(b);
    if (((n.flags & EXF_LHS_NULLABLE) && x == kNullDouble) || ((n.flags & EXF_RHS_NULLABLE) && y == kNullDouble))
      r = kNullDoubleBits;
    else
      r = dbl_bits(n.op == MI355Q_EX_ADD ? x + y : n.op == MI355Q_EX_SUB ? x - y : x * y);
  } else {
    const float x = ex_flt_of(a), y = ex_flt_of(b);
    if (((n.flags & EXF_LHS_NULLABLE) && x == kNullFloat) || ((n.flags & EXF_RHS_NULLABLE) && y == kNullFloat))
      r = (int64_t)(uint32_t)kNullFloatBits;
    else
      r = ex_flt_pattern(n.op == MI355Q_EX_ADD ? x + y : n.op == MI355Q_EX_SUB ? x - y : x * y);
  }
  return r;
}

// *err receives MI355Q_ERR_OVERFLOW_OR_UNDERFLOW / MI355Q_ERR_DIV_BY_ZERO when a check fires on the way to the RESULT (the
// value returned is then unspecified): the first one in evaluation order — operands left to right, then the operation; a
// CASE's condition, then the branch it takes.  es[] = the error each stack value carries.
// xv (optional): the values of the plan's EARLIER expressions for this row, computed by the caller in order — a kernel that
// evaluates expressions in registers instead of reading the dense temporary columns a projection pass left (n_phys = the
// number of physical columns: a column node `arg` >= n_phys reads xv[arg - n_phys]).
MQ_HD int64_t eval_expr(const DevExpr& e, const int8_t* const* cols, int64_t pos, int32_t* err, const int64_t* xv = nullptr,
                        int n_phys = 0) {
  int64_t st[MI355Q_MAX_EXPR_STACK] = {};
  int32_t es[MI355Q_MAX_EXPR_STACK] = {};
  int sp = 0;
  for (int i = 0; i < e.n_nodes; ++i) {
    const DevExprNode& n = e.nodes[i];
    switch (n.op) {
      case MI355Q_EX_COL:
        // (xv: the value of an earlier expression, already in the caller's registers)
        st[sp] = (xv && n.arg >= n_phys) ? xv[n.arg - n_phys] : ex_col(n, cols, pos);
        es[sp++] = 0;
        break;
      case MI355Q_EX_LIT:
        st[sp] = ex_lit(n);
        es[sp++] = 0;
        break;
      case MI355Q_EX_CAST: st[sp - 1] = ex_cast(n, st[sp - 1], es[sp - 1]); break;
      case MI355Q_EX_DIV:
      case MI355Q_EX_MOD:
        --sp;
        if (!es[sp - 1]) es[sp - 1] = es[sp];  // (lhs first, then rhs, then this operation)
        st[sp - 1] = ex_divmod(n, st[sp - 1], st[sp], es[sp - 1]);
        break;
      case MI355Q_EX_EQ:
      case MI355Q_EX_NE:
      case MI355Q_EX_LT:
      case MI355Q_EX_LE:
      case MI355Q_EX_GT:
      case MI355Q_EX_GE:
        --sp;
        if (!es[sp - 1]) es[sp - 1] = es[sp];
        st[sp - 1] = ex_cmp(n, st[sp - 1], st[sp]);
        break;
      case MI355Q_EX_CASE: {  // stack: ELSE, THEN, cond
        const int64_t c = st[sp - 1];
        const int32_t ce = es[sp - 1];
        sp -= 2;
        const bool take = c == 1;  // TRUE; 0 and NULL are not
        st[sp - 1] = take ? st[sp] : st[sp - 1];
        es[sp - 1] = ce ? ce : take ? es[sp] : es[sp - 1];
        break;
      }
      case MI355Q_EX_NOT: st[sp - 1] = ex_not(n, st[sp - 1]); break;
      case MI355Q_EX_AND:
      case MI355Q_EX_OR:
        --sp;
        st[sp - 1] = ex_logic(n, st[sp - 1], st[sp], es[sp - 1], es[sp]);
        break;
      case MI355Q_EX_IS_NULL: st[sp - 1] = ex_is_null(n, st[sp - 1], es[sp - 1]); break;
      case MI355Q_EX_UMINUS: st[sp - 1] = ex_uminus(n, st[sp - 1], es[sp - 1]); break;
      default:  // MI355Q_EX_ADD / _SUB / _MUL
        --sp;
        if (!es[sp - 1]) es[sp - 1] = es[sp];
        st[sp - 1] = ex_arith(n, st[sp - 1], st[sp], es[sp - 1]);
    }
  }
  if (es[0] && !*err) *err = es[0];
  return ex_wrap_int(e.type, st[0]);
}

// ---- typed handlers.  The flat loops above decide, per node and per row, what `type`, the nullability flags and the
// operation say; all of that is known when the step is planned.  k_project's upload therefore labels every node with a
// HANDLER: one instantiation of the same ex_* function with the node's constant fields compiled in (the compiler folds the
// decisions away), picked by one wave-uniform switch per node and J x 64 rows.  A node whose combination has no handler
// (FLOAT, INT8 / INT16 arithmetic, ...) keeps handler 0 and runs through the flat code.
enum : int32_t {
  XH_GENERIC = 0,
  XH_COLPRE = 1,               // + slot (4): the value was loaded with the tile (eval_expr_rows `raw`)
  XH_LIT = XH_COLPRE + 4,      // the pattern is in ilit
  XH_CMP = XH_LIT + 1,         // + ((op - EQ) * 2 + t2) * 4 + nf           t2: 0 INT32, 1 INT64; nf: 1 lhs nullable, 2 rhs nullable
  XH_ARITH = XH_CMP + 48,      // + ((op - ADD) * 3 + t3) * 4 + nf          t3: 0 INT32, 1 INT64, 2 DOUBLE
  XH_DIVMOD = XH_ARITH + 36,   // + ((op - DIV) * 3 + t3) * 4 + nf
  XH_LOGIC = XH_DIVMOD + 24,   // + ((op - AND) * 2 + short_circuit) * 2 + nullable
  XH_NOT = XH_LOGIC + 8,       // + lhs nullable
  XH_ISNULL = XH_NOT + 2,      // + t3 * 2 + lhs nullable
  XH_CAST = XH_ISNULL + 6,     // + (from3 * 3 + to3) * 2 + lhs nullable
  XH_UMINUS = XH_CAST + 18,    // + t3 * 2 + lhs nullable
  XH_CASE = XH_UMINUS + 6,
  XH_GCOL = XH_CASE + 1,       // no typed handler: the flat code on the lowered node itself (encoded / narrow columns and the
  XH_GUN = XH_GCOL + 1,        // values of earlier expressions; FLOAT, INT8 / INT16 operations)
  XH_GBIN = XH_GUN + 1,
  XH_END = XH_GBIN + 1
};
// a node as the kernel reads it from LDS: the handler, the literal's pattern
struct XNode {
  int32_t h, pad_;
  int64_t lit;
};
constexpr int kExHandlerShift = 8;  // the handler sits above the EXF_* bits of DevExprNode::flags (device copy only)
MQ_HD constexpr int xh_t3(int t) { return t == MI355Q_INT32 ? 0 : t == MI355Q_INT64 ? 1 : t == MI355Q_DOUBLE ? 2 : -1; }
MQ_HD constexpr int xh_type_of_t3(int i) { return i == 0 ? MI355Q_INT32 : i == 1 ? MI355Q_INT64 : MI355Q_DOUBLE; }
MQ_HD constexpr int xh_nf_flags(int nf) { return ((nf & 1) ? EXF_LHS_NULLABLE : 0) | ((nf & 2) ? EXF_RHS_NULLABLE : 0); }
// the handler of a lowered node (column nodes: the caller, which knows what it loads up front)
inline int xh_of(const DevExprNode& n) {
  const int nf = ((n.flags & EXF_LHS_NULLABLE) ? 1 : 0) | ((n.flags & EXF_RHS_NULLABLE) ? 2 : 0);
  const int lhs = (n.flags & EXF_LHS_NULLABLE) ? 1 : 0;
  switch (n.op) {
    case MI355Q_EX_LIT: return XH_LIT;
    case MI355Q_EX_EQ: case MI355Q_EX_NE: case MI355Q_EX_LT: case MI355Q_EX_LE: case MI355Q_EX_GT: case MI355Q_EX_GE: {
      const int t = xh_t3(n.arg);
      return t < 0 || t > 1 ? XH_GBIN : XH_CMP + ((n.op - MI355Q_EX_EQ) * 2 + t) * 4 + nf;
    }
    case MI355Q_EX_ADD: case MI355Q_EX_SUB: case MI355Q_EX_MUL: {
      const int t = xh_t3(n.type);
      return t < 0 ? XH_GBIN : XH_ARITH + ((n.op - MI355Q_EX_ADD) * 3 + t) * 4 + nf;
    }
    case MI355Q_EX_DIV: case MI355Q_EX_MOD: {
      const int t = xh_t3(n.type);
      return t < 0 || (n.op == MI355Q_EX_MOD && t == 2) ? XH_GBIN : XH_DIVMOD + ((n.op - MI355Q_EX_DIV) * 3 + t) * 4 + nf;
    }
    case MI355Q_EX_AND: case MI355Q_EX_OR:
      return XH_LOGIC + ((n.op - MI355Q_EX_AND) * 2 + ((n.flags & EXF_SHORT_CIRCUIT) ? 1 : 0)) * 2 + ((n.flags & EXF_NULLABLE) ? 1 : 0);
    case MI355Q_EX_NOT: return XH_NOT + lhs;
    case MI355Q_EX_IS_NULL: {
      const int t = xh_t3(n.arg);
      return t < 0 ? XH_GUN : XH_ISNULL + t * 2 + lhs;
    }
    case MI355Q_EX_CAST: {
      const int f = xh_t3(n.arg), t = xh_t3(n.type);
      return f < 0 || t < 0 ? XH_GUN : XH_CAST + (f * 3 + t) * 2 + lhs;
    }
    case MI355Q_EX_UMINUS: {
      const int t = xh_t3(n.type);
      return t < 0 ? XH_GUN : XH_UMINUS + t * 2 + lhs;
    }
    case MI355Q_EX_CASE: return XH_CASE;
    default: return XH_GCOL;
  }
}

// The device copy of a plan's programs: every node labelled with its handler (above the EXF_* bits of `flags`), literals
// laid down as patterns, the error-free programs marked, and — use_pre — the first kExPre plain physical columns the
// programs read listed for the caller's per-tile batch of loads (their column nodes become XH_COLPRE + slot).
// Returns the deepest evaluation stack of the programs.
inline int xh_label_programs(DevExprSet* up, bool use_pre) {
  int deepest = 1;
  up->n_pre = 0;
  up->noerr_mask = 0;
  for (int k = 0; k < up->n; ++k) {
    bool noerr = true;
    int sp = 0;
    for (int i = 0; i < up->e[k].n_nodes; ++i) {
      DevExprNode& n = up->e[k].nodes[i];
      const int op = n.op;
      noerr = noerr && (op == MI355Q_EX_COL || op == MI355Q_EX_LIT || (op >= MI355Q_EX_EQ && op <= MI355Q_EX_GE) || op == MI355Q_EX_CASE ||
                        op == MI355Q_EX_NOT || op == MI355Q_EX_AND || op == MI355Q_EX_OR || op == MI355Q_EX_IS_NULL);
      if (op == MI355Q_EX_COL || op == MI355Q_EX_LIT) ++sp;
      else if (op == MI355Q_EX_CASE) sp -= 2;
      else if (op != MI355Q_EX_CAST && op != MI355Q_EX_NOT && op != MI355Q_EX_IS_NULL && op != MI355Q_EX_UMINUS) --sp;
      if (sp > deepest) deepest = sp;
      int h = XH_GCOL;
      if (op == MI355Q_EX_COL) {
        const int code = (int)n.ilit;
        if (use_pre && n.arg < up->n_cols && (code == MI355Q_INT32 || code == MI355Q_INT64 || code == MI355Q_DOUBLE)) {
          int slot = -1;
          for (int c = 0; c < up->n_pre; ++c)
            if (up->pre_col[c] == n.arg) slot = c;
          if (slot < 0 && up->n_pre < 4) {
            slot = up->n_pre++;
            up->pre_col[slot] = n.arg;
            up->pre_type[slot] = n.type;
            up->pre_code[slot] = code;
          }
          if (slot >= 0) h = XH_COLPRE + slot;
        }
      } else {
        h = xh_of(n);
        if (op == MI355Q_EX_LIT) {  // (ex_lit's pattern, laid down once)
          n.ilit = ex_lit(n);
          n.arg = 1;
        }
      }
      n.flags = (n.flags & ((1 << kExHandlerShift) - 1)) | (h << kExHandlerShift);
    }
    if (noerr) up->noerr_mask |= 1 << k;
  }
  return deepest;
}

#if defined(__HIPCC__) || defined(HOSTSIM_DEVICE_CODE)
// a value every lane of the wave holds alike, moved to a scalar register (the host simulation has no such distinction)
#if defined(__HIPCC__)
#define MQ_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define MQ_WAVE_UNIFORM(x) (x)
#endif
// the node a handler stands for: exactly the fields the ex_* function of its family reads
template <int H>
MQ_D int64_t xh_binary(int64_t a, int64_t b, int32_t& ev) {  // XH_CMP .. XH_LOGIC - 1
  DevExprNode n{};
  if constexpr (H < XH_ARITH) {
    constexpr int k = H - XH_CMP;
    n.op = MI355Q_EX_EQ + k / 8;
    n.arg = xh_type_of_t3((k / 4) % 2);
    n.flags = xh_nf_flags(k % 4);
    return ex_cmp(n, a, b);
  } else if constexpr (H < XH_DIVMOD) {
    constexpr int k = H - XH_ARITH;
    n.op = MI355Q_EX_ADD + k / 12;
    n.type = xh_type_of_t3((k / 4) % 3);
    n.flags = xh_nf_flags(k % 4);
    return ex_arith(n, a, b, ev);
  } else {
    constexpr int k = H - XH_DIVMOD;
    n.op = MI355Q_EX_DIV + k / 12;
    n.type = xh_type_of_t3((k / 4) % 3);
    n.flags = xh_nf_flags(k % 4);
    return ex_divmod(n, a, b, ev);
  }
}
template <int H>
MQ_D int64_t xh_logic(int64_t a, int64_t b, int32_t& ea, int32_t eb) {  // XH_LOGIC .. XH_NOT - 1
  constexpr int k = H - XH_LOGIC;
  DevExprNode n{};
  n.op = MI355Q_EX_AND + k / 4;
  n.flags = (((k / 2) % 2) ? EXF_SHORT_CIRCUIT : 0) | ((k % 2) ? EXF_NULLABLE : 0);
  return ex_logic(n, a, b, ea, eb);
}
template <int H>
MQ_D int64_t xh_unary(int64_t v, int32_t& ev) {  // XH_NOT .. XH_CASE - 1
  DevExprNode n{};
  if constexpr (H < XH_ISNULL) {
    n.flags = (H - XH_NOT) ? EXF_LHS_NULLABLE : 0;
    return ex_not(n, v);
  } else if constexpr (H < XH_CAST) {
    constexpr int k = H - XH_ISNULL;
    n.arg = xh_type_of_t3(k / 2);
    n.flags = (k % 2) ? EXF_LHS_NULLABLE : 0;
    return ex_is_null(n, v, ev);
  } else if constexpr (H < XH_UMINUS) {
    constexpr int k = H - XH_CAST;
    n.arg = xh_type_of_t3(k / 6);
    n.type = xh_type_of_t3((k / 2) % 3);
    n.flags = (k % 2) ? EXF_LHS_NULLABLE : 0;
    return ex_cast(n, v, ev);
  } else {
    constexpr int k = H - XH_UMINUS;
    n.type = xh_type_of_t3(k / 2);
    n.flags = (k % 2) ? EXF_LHS_NULLABLE : 0;
    return ex_uminus(n, v, ev);
  }
}
#define XH_REP2(m, b) m((b)) m((b) + 1)
#define XH_REP4(m, b) XH_REP2(m, (b)) XH_REP2(m, (b) + 2)
#define XH_REP8(m, b) XH_REP4(m, (b)) XH_REP4(m, (b) + 4)
#define XH_REP16(m, b) XH_REP8(m, (b)) XH_REP8(m, (b) + 8)
#define XH_REP32(m, b) XH_REP16(m, (b)) XH_REP16(m, (b) + 16)
#define XH_REP64(m, b) XH_REP32(m, (b)) XH_REP32(m, (b) + 32)
static_assert(XH_LOGIC - XH_CMP == 108 && XH_NOT - XH_LOGIC == 8 && XH_CASE - XH_NOT == 32, "the case lists below");

// ---- the kernels' form: J rows of one lane at a time, so that a node is decoded once per J x 64 rows, and NO private
// array: a stack indexed by a run-time depth is laid out in scratch memory by the compiler (k_project of rounds 3-4:
// 80 bytes of scratch per lane, three trips through it per node — the pass ran at 1.4 TB/s, profiles/README.md).  Here
// the top of the stack stays in registers (tv / te), the values below it live in LDS, one 8-byte column per (depth, j)
// and lane — conflict-free, addressed with the wave-uniform depth — and the errors they carry in two bits each of one
// register per row.  The program is read from LDS too (XNode, one node ahead of its use): a scalar load per node would
// park the wave on its latency each time.
//   s_st: (depth - 1) * J * n_threads 8-byte words
struct ExLdsStack {
  int64_t* st;
  int tid;
};
MQ_HD uint32_t ex_err_enc(int32_t c) { return (uint32_t)((c & 1) + ((c >> 2) & 1)); }  // 0 / 1 / 7 -> 0 / 1 / 2
MQ_HD int32_t ex_err_dec(uint32_t x) { return (int32_t)(x + (x >> 1) * 5u); }
// raw: the values of up to kExPre plain physical columns, loaded by the caller for ALL J rows before the first node is
// interpreted (one batch of independent loads per tile instead of one exposed memory latency per column node)
constexpr int kExPre = 4;
// NT: threads of the workgroup; ERR = false: a program that cannot raise an error (comparisons, AND / OR / NOT, IS NULL,
// CASE over columns and literals) carries none
// xv (optional, LDS): the values of the plan's EARLIER expressions for these rows — value of expression k, row j at
// xv[(k * J + j) * NT + tid] — for a kernel that keeps no temporary columns (a column node `arg` >= n_phys reads them)
template <int J, int NT, bool ERR>
MQ_D void eval_expr_rows(const DevExpr& e, const XNode* prog, const int8_t* const* cols, const int64_t (&pos)[J],
                         const int64_t (&raw)[kExPre][J], const ExLdsStack& s, int64_t (&out)[J], int32_t (&err)[J],
                         const int64_t* xv = nullptr, int n_phys = 0) {
  int64_t tv[J];
  int32_t te[J];
  uint32_t be[J];  // the errors of the values below the top, two bits each, the nearest in bits 0-1
#pragma unroll
  for (int j = 0; j < J; ++j) {
    tv[j] = 0;
    te[j] = 0;
    be[j] = 0;
  }
  int sp = 0;  // values on the stack: tv = value sp - 1, LDS slot d = value d (d < sp - 1)
  auto slot = [&](int d, int j) { return (size_t)(d * J + j) * NT + s.tid; };
  const int nn = e.n_nodes;
  XNode nd = prog[0];
#pragma unroll 1
  for (int i = 0; i < nn; ++i) {
    const int h = MQ_WAVE_UNIFORM(nd.h);
    const int64_t lit = (int64_t)(((uint64_t)(uint32_t)MQ_WAVE_UNIFORM((int)(nd.lit >> 32)) << 32) | (uint32_t)MQ_WAVE_UNIFORM((int)nd.lit));
    if (i + 1 < nn) nd = prog[i + 1];
    if (h <= XH_LIT || h == XH_GCOL) {  // push
      if (sp > 0) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          s.st[slot(sp - 1, j)] = tv[j];
          if (ERR) be[j] = (be[j] << 2) | ex_err_enc(te[j]);
        }
      }
      if (h == XH_LIT) {
#pragma unroll
        for (int j = 0; j < J; ++j) tv[j] = lit;
      } else if (h == XH_COLPRE) {
#pragma unroll
        for (int j = 0; j < J; ++j) tv[j] = raw[0][j];
      } else if (h == XH_COLPRE + 1) {
#pragma unroll
        for (int j = 0; j < J; ++j) tv[j] = raw[1][j];
      } else if (h == XH_COLPRE + 2) {
#pragma unroll
        for (int j = 0; j < J; ++j) tv[j] = raw[2][j];
      } else if (h == XH_COLPRE + 3) {
#pragma unroll
        for (int j = 0; j < J; ++j) tv[j] = raw[3][j];
      } else {
        const DevExprNode& n = e.nodes[i];
        if (xv && n.arg >= n_phys) {
#pragma unroll
          for (int j = 0; j < J; ++j) tv[j] = xv[(size_t)((n.arg - n_phys) * J + j) * NT + s.tid];
        } else {
#pragma unroll
          for (int j = 0; j < J; ++j) tv[j] = ex_col(n, cols, pos[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j) te[j] = 0;
      ++sp;
      continue;
    }
    if (h >= XH_NOT && (h < XH_CASE || h == XH_GUN)) {  // unary
      switch (h) {
#define XH_UN_CASE(H)                                               \
  case H:                                                           \
    _Pragma("unroll") for (int j = 0; j < J; ++j) tv[j] = xh_unary<H>(tv[j], te[j]); \
    break;
        XH_REP32(XH_UN_CASE, XH_NOT)
#undef XH_UN_CASE
        default: {
          const DevExprNode& n = e.nodes[i];
          const int op = n.op;
#pragma unroll
          for (int j = 0; j < J; ++j)
            tv[j] = op == MI355Q_EX_CAST ? ex_cast(n, tv[j], te[j]) : op == MI355Q_EX_NOT ? ex_not(n, tv[j])
                    : op == MI355Q_EX_IS_NULL ? ex_is_null(n, tv[j], te[j]) : ex_uminus(n, tv[j], te[j]);
        }
      }
      continue;
    }
    if (h == XH_CASE) {  // stack: ELSE (sp - 3), THEN (sp - 2), cond (tv)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const bool take = tv[j] == 1;  // TRUE; 0 and NULL are not
        tv[j] = s.st[slot(take ? sp - 2 : sp - 3, j)];
        if (ERR) {
          if (!te[j]) te[j] = ex_err_dec(take ? (be[j] & 3u) : ((be[j] >> 2) & 3u));
          be[j] >>= 4;
        }
      }
      sp -= 2;
      continue;
    }
    // binary: lhs = LDS slot sp - 2, rhs = tv
    int64_t a[J];
    int32_t ea[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      a[j] = s.st[slot(sp - 2, j)];
      ea[j] = 0;
      if (ERR) {
        ea[j] = ex_err_dec(be[j] & 3u);
        be[j] >>= 2;
      }
    }
    --sp;
    if (h >= XH_LOGIC && h < XH_NOT) {
      switch (h) {
#define XH_LG_CASE(H)                                                 \
  case H:                                                             \
    _Pragma("unroll") for (int j = 0; j < J; ++j) {                   \
      tv[j] = xh_logic<H>(a[j], tv[j], ea[j], te[j]);                 \
      te[j] = ea[j];                                                  \
    }                                                                 \
    break;
        XH_REP8(XH_LG_CASE, XH_LOGIC)
#undef XH_LG_CASE
      }
      continue;
    }
    if (ERR) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (!ea[j]) ea[j] = te[j];  // (lhs first, then rhs, then this operation)
        te[j] = ea[j];
      }
    }
    switch (h) {
#define XH_BIN_CASE(H)                                                \
  case H:                                                             \
    _Pragma("unroll") for (int j = 0; j < J; ++j) tv[j] = xh_binary<H>(a[j], tv[j], te[j]); \
    break;
      XH_REP64(XH_BIN_CASE, XH_CMP)
      XH_REP32(XH_BIN_CASE, XH_CMP + 64)
      XH_REP8(XH_BIN_CASE, XH_CMP + 96)
      XH_REP4(XH_BIN_CASE, XH_CMP + 104)
#undef XH_BIN_CASE
      default: {
        const DevExprNode& n = e.nodes[i];
        const int op = n.op;
        if (op == MI355Q_EX_DIV || op == MI355Q_EX_MOD) {
#pragma unroll
          for (int j = 0; j < J; ++j) tv[j] = ex_divmod(n, a[j], tv[j], te[j]);
        } else if (op >= MI355Q_EX_EQ && op <= MI355Q_EX_GE) {
#pragma unroll
          for (int j = 0; j < J; ++j) tv[j] = ex_cmp(n, a[j], tv[j]);
        } else {
#pragma unroll
          for (int j = 0; j < J; ++j) tv[j] = ex_arith(n, a[j], tv[j], te[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    out[j] = ex_wrap_int(e.type, tv[j]);
    err[j] = ERR ? te[j] : 0;
  }
}
#endif

// the value as the dense temporary column stores it: a plain column of the expression's type — or of its wider store_type
// (DevExpr), the NULL of the one becoming the NULL of the other
MQ_HD void store_expr_value(int8_t* col, const DevExpr& e, int64_t pos, int64_t v) {
  if (e.store_type != e.type && v == plain_int_null(e.type)) v = plain_int_null(e.store_type);
  switch (e.store_type) {
    case MI355Q_INT8: *(int8_t*)(col + pos) = (int8_t)v; break;
    case MI355Q_INT16: *(int16_t*)(col + pos * 2) = (int16_t)v; break;
    case MI355Q_INT32:
    case MI355Q_FLOAT: *(int32_t*)(col + pos * 4) = (int32_t)(uint32_t)v; break;
    default: *(int64_t*)(col + pos * 8) = v;
  }
}

}  // namespace mq
