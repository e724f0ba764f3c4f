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
      case MI355Q_EX_COL: {
        if (xv && n.arg >= n_phys) {  // the value of an earlier expression, already in the caller's registers
          st[sp++] = xv[n.arg - n_phys];
          es[sp - 1] = 0;
          break;
        }
        const int8_t* c = cols[n.arg];
        if (n.type == MI355Q_DOUBLE) st[sp++] = *(const int64_t*)(c + pos * 8);
        else if (n.type == MI355Q_FLOAT) st[sp++] = (int64_t)*(const uint32_t*)(c + pos * 4);
        else st[sp++] = decode_int(c, (int)n.ilit, pos);
        es[sp - 1] = 0;
        break;
      }
      case MI355Q_EX_LIT:
        st[sp++] = n.arg ? n.ilit  // (the NULL literal: its pattern was laid down by the lowering)
                   : n.type == MI355Q_DOUBLE ? dbl_bits(n.flit)
                   : n.type == MI355Q_FLOAT ? ex_flt_pattern((float)n.flit) : n.ilit;
        es[sp - 1] = 0;
        break;
      case MI355Q_EX_CAST: {
        const int from = n.arg, to = n.type;
        const bool nullable = (n.flags & EXF_LHS_NULLABLE) != 0;
        const int64_t v = st[sp - 1];
        int64_t r = v;
        if (ex_is_int(from)) {
          const bool is_null = nullable && v == plain_int_null(from);
          if (ex_is_int(to)) {
            if (is_null) {
              r = plain_int_null(to);
            } else if (plain_width(to) < plain_width(from) && (v > ex_int_max(to) || v <= ex_int_min(to))) {
              if (!es[sp - 1]) es[sp - 1] = MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
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
        st[sp - 1] = r;
        break;
      }
      case MI355Q_EX_DIV:
      case MI355Q_EX_MOD: {
        const int64_t b = st[--sp];
        const int64_t a = st[sp - 1];
        int32_t& ev = es[sp - 1];
        if (!ev) ev = es[sp];  // (lhs first, then rhs, then this operation)
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
        st[sp - 1] = r;
        break;
      }
      case MI355Q_EX_EQ:
      case MI355Q_EX_NE:
      case MI355Q_EX_LT:
      case MI355Q_EX_LE:
      case MI355Q_EX_GT:
      case MI355Q_EX_GE: {
        const int64_t b = st[--sp];
        const int64_t a = st[sp - 1];
        if (!es[sp - 1]) es[sp - 1] = es[sp];
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
        st[sp - 1] = is_null ? plain_int_null(MI355Q_INT8) : (v ? 1 : 0);
        break;
      }
      case MI355Q_EX_CASE: {  // stack: ELSE, THEN, cond
        const int64_t c = st[sp - 1];
        const int32_t ce = es[sp - 1];
        sp -= 2;
        const bool take = c == 1;  // TRUE; 0 and NULL are not
        st[sp - 1] = take ? st[sp] : st[sp - 1];
        es[sp - 1] = ce ? ce : take ? es[sp] : es[sp - 1];
        break;
      }
      case MI355Q_EX_NOT: {
        const int64_t v = st[sp - 1];
        if ((n.flags & EXF_LHS_NULLABLE) && v == plain_int_null(MI355Q_INT8)) break;  // logical_not: NULL stays NULL
        st[sp - 1] = (n.flags & EXF_LHS_NULLABLE) ? (v ? 0 : 1) : (v > 0 ? 0 : 1);       // (toBool on the NOT NULL side)
        break;
      }
      case MI355Q_EX_AND:
      case MI355Q_EX_OR: {
        const int64_t b = st[--sp];
        const int64_t a = st[sp - 1];
        const int64_t nul = plain_int_null(MI355Q_INT8);
        const bool is_or = n.op == MI355Q_EX_OR, nullable = (n.flags & EXF_NULLABLE) != 0;
        int64_t r;
        if (n.flags & EXF_SHORT_CIRCUIT) {
          // the first operand alone where it decides — the second one's checks then do not exist
          if (nullable && a == nul) r = nul;
          else if (a == (is_or ? 1 : 0)) r = a;
          else {
            if (!es[sp - 1]) es[sp - 1] = es[sp];
            r = b;
          }
        } else {
          if (!es[sp - 1]) es[sp - 1] = es[sp];
          if (!nullable) r = is_or ? (a > 0 || b > 0) : (a > 0 && b > 0);
          else if (a == nul) r = is_or ? (b == 0 ? nul : b) : (b == 0 ? b : nul);  // logical_or / logical_and
          else if (b == nul) r = is_or ? (a == 0 ? nul : a) : (a == 0 ? a : nul);
          else r = is_or ? (a || b) : (a && b);
        }
        st[sp - 1] = r;
        break;
      }
      case MI355Q_EX_IS_NULL: {
        const int64_t v = st[sp - 1];
        const int t = n.arg;  // the operand's type
        if (!(n.flags & EXF_LHS_NULLABLE)) {  // a NOT NULL operand is never evaluated: constant false, no check of it exists
          st[sp - 1] = 0;
          es[sp - 1] = 0;
        } else {
          st[sp - 1] = ex_is_int(t) ? v == plain_int_null(t) : t == MI355Q_DOUBLE ? bits_dbl(v) == kNullDouble : ex_flt_of(v) == kNullFloat;
        }
        break;
      }
      case MI355Q_EX_UMINUS: {
        const int64_t v = st[sp - 1];
        const int t = n.type;
        const bool nullable = (n.flags & EXF_LHS_NULLABLE) != 0;
        if (ex_is_int(t)) {
          if (v == plain_int_null(t)) {  // the type's minimum: NULL stays NULL, a value cannot be negated
            if (!nullable && !es[sp - 1]) es[sp - 1] = MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
          } else {
            st[sp - 1] = -v;
          }
        } else if (t == MI355Q_DOUBLE) {
          const double x = bits_dbl(v);
          if (!(nullable && x == kNullDouble)) st[sp - 1] = dbl_bits(-x);
        } else {
          const float x = ex_flt_of(v);
          if (!(nullable && x == kNullFloat)) st[sp - 1] = ex_flt_pattern(-x);
        }
        break;
      }
      default: {  // MI355Q_EX_ADD / _SUB / _MUL
        const int64_t b = st[--sp];
        const int64_t a = st[sp - 1];
        if (!es[sp - 1]) es[sp - 1] = es[sp];
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
            if (ovf && !es[sp - 1]) es[sp - 1] = MI355Q_ERR_OVERFLOW_OR_UNDERFLOW;
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
        st[sp - 1] = r;
      }
    }
  }
  if (es[0] && !*err) *err = es[0];
  return st[0];
}

// the value as the dense temporary column stores it (a plain column of the expression's type)
MQ_HD void store_expr_value(int8_t* col, int type, int64_t pos, int64_t v) {
  switch (type) {
    case MI355Q_INT8: *(int8_t*)(col + pos) = (int8_t)v; break;
    case MI355Q_INT16: *(int16_t*)(col + pos * 2) = (int16_t)v; break;
    case MI355Q_INT32:
    case MI355Q_FLOAT: *(int32_t*)(col + pos * 4) = (int32_t)(uint32_t)v; break;
    default: *(int64_t*)(col + pos * 8) = v;
  }
}

}  // namespace mq
