// Mi355qTranslate.h — the type-mapping half of the HeavyDB binding: SQLTypeInfo -> mi355q_type, SQLAgg -> mi355q_agg,
// Analyzer expressions -> mi355q_expr / mi355q_qual / mi355q_target.  Header-only, and written against NOTHING but
//   Shared/sqltypes.h, Shared/sqldefs.h, Analyzer/Analyzer.h
// so that it compiles against the REFERENCE'S OWN headers in this image (g++ -DNO_BOOST, the Boost-free mode
// Logger/Logger.h provides; the way oracle/ref_layout_shim.cpp does): integration/real_headers_check.cpp builds real
// Analyzer::ColumnVar / Constant / UOper / BinOper / AggExpr objects and runs every function below on them
// (tests/test_integration_glue.py::test_translate_half_against_the_reference_headers).  The other half of the binding
// (Mi355qExecutor.cpp: RelAlgExecutionUnit, InputTableInfo, getExpressionRange, ResultSet) needs headers that do NOT
// compile here — QueryEngine/RelAlgExecutionUnit.h, InputMetadata.h, ResultSet.h, ColumnFetcher.h stop at
// Shared/StringTransform.h:21 `#include <boost/config.hpp>` (reached through Fragmenter/../Catalog/), ExpressionRange.h
// at :22 `#include <boost/multiprecision/cpp_int.hpp>`, JoinHashTable/HashJoin.h at :19 `#include <llvm/IR/Value.h>` —
// and stays on integration/mock/heavydb_mock.h.
#pragma once

#include <functional>
#include <vector>
#include <stdexcept>
#include <string>

#include "mi355q.h"

#ifndef MI355Q_GLUE_MOCK_HEADERS
#include "Analyzer/Analyzer.h"
#include "Shared/sqldefs.h"
#include "Shared/sqltypes.h"
#endif

// The plan ABI's caps as this binding sees them.  A test build may LOWER them (-DMI355Q_GLUE_MAX_EXPR_NODES=12
// -DMI355Q_GLUE_MAX_QUALS=4: the caps of ABI 6) so that the splitting machinery below — programs stated as several
// expressions, conjuncts folded into a BOOLEAN expression — stays exercised by the reference's own WHERE clauses.
#ifndef MI355Q_GLUE_MAX_EXPR_NODES
#define MI355Q_GLUE_MAX_EXPR_NODES MI355Q_MAX_EXPR_NODES
#endif
#ifndef MI355Q_GLUE_MAX_QUALS
#define MI355Q_GLUE_MAX_QUALS MI355Q_MAX_QUALS
#endif
static_assert(MI355Q_GLUE_MAX_EXPR_NODES <= MI355Q_MAX_EXPR_NODES && MI355Q_GLUE_MAX_QUALS <= MI355Q_MAX_QUALS, "the binding cannot exceed the ABI");

namespace mi355q_glue {
constexpr int kGlueMaxExprNodes = MI355Q_GLUE_MAX_EXPR_NODES;
constexpr int kGlueMaxQuals = MI355Q_GLUE_MAX_QUALS;


[[noreturn]] inline void unsupported(const char* what) { throw std::runtime_error(std::string("mi355q: ") + what); }
// a program beyond kGlueMaxExprNodes: the executor half may state it as several expressions (split_plain_logic)
struct ExprTooLong : std::runtime_error {
  ExprTooLong() : std::runtime_error("mi355q: expression too long") {}
};

// Types the plan ABI cannot state are refused here instead of being read as integers of their byte width: DECIMAL /
// NUMERIC (scale: casts, multiplication and AVG would need scale_decimal_up / _down), none-encoded strings, arrays,
// geo, TIMESTAMP(3|6|9) and intervals.
inline void check_supported_type(const SQLTypeInfo& ti) {
  if (ti.is_decimal()) unsupported("DECIMAL / NUMERIC column or literal");
  if (ti.is_array()) unsupported("array column");
  if (ti.is_geometry()) unsupported("geo column");
  if (ti.is_string() && ti.get_compression() != kENCODING_DICT) unsupported("none-encoded string column");
  if (ti.is_high_precision_timestamp()) unsupported("TIMESTAMP with a dimension");
  switch (ti.get_type()) {
    case kBOOLEAN: case kTINYINT: case kSMALLINT: case kINT: case kBIGINT: case kFLOAT: case kDOUBLE:
    case kDATE: case kTIME: case kTIMESTAMP: case kTEXT: case kVARCHAR: case kCHAR: break;
    default: unsupported("column type");
  }
}

// the chunk's STORAGE type (ColumnFetcher hands chunks over undecoded)
inline int32_t storage_type(const SQLTypeInfo& ti) {
  check_supported_type(ti);
  if (ti.get_type() == kDOUBLE) return MI355Q_DOUBLE;
  if (ti.get_type() == kFLOAT) return MI355Q_FLOAT;
  switch (ti.get_size()) {
    case 1: return MI355Q_INT8;
    case 2: return MI355Q_INT16;
    case 4: return MI355Q_INT32;
    case 8: return MI355Q_INT64;
    default: unsupported("column width");
  }
}
inline int32_t logical_type(const SQLTypeInfo& ti) {
  check_supported_type(ti);
  if (ti.get_type() == kDOUBLE) return MI355Q_DOUBLE;
  if (ti.get_type() == kFLOAT) return MI355Q_FLOAT;
  switch (ti.get_logical_size()) {
    case 1: return MI355Q_INT8;
    case 2: return MI355Q_INT16;
    case 4: return MI355Q_INT32;
    default: return MI355Q_INT64;
  }
}
inline int32_t encoding_of(const SQLTypeInfo& ti) {
  switch (ti.get_compression()) {
    case kENCODING_FIXED: return MI355Q_ENC_FIXED;
    case kENCODING_DICT: return MI355Q_ENC_DICT;            // ids; 1/2-byte chunks are unsigned
    case kENCODING_DATE_IN_DAYS: return MI355Q_ENC_DATE_IN_DAYS;
    case kENCODING_NONE: return MI355Q_ENC_NONE;
    default: unsupported("column encoding");
  }
}

// SQLAgg (Shared/sqldefs.h:76-90) -> mi355q_agg.  kAVG .. kCOUNT and kCOUNT_IF / kSUM_IF carry the reference's numeric
// values in the ABI; everything else (APPROX_*, SAMPLE, SINGLE_VALUE, MODE) is not an aggregate of this path.
inline int32_t agg_kind(SQLAgg a) {
  switch (a) {
    case kAVG: return MI355Q_AVG;
    case kMIN: return MI355Q_MIN;
    case kMAX: return MI355Q_MAX;
    case kSUM: return MI355Q_SUM;
    case kCOUNT: return MI355Q_COUNT;
    case kCOUNT_IF: return MI355Q_COUNT_IF;
    case kSUM_IF: return MI355Q_SUM_IF;
    default: unsupported("aggregate kind");
  }
}
static_assert((int)kAVG == MI355Q_AVG && (int)kMIN == MI355Q_MIN && (int)kMAX == MI355Q_MAX && (int)kSUM == MI355Q_SUM &&
                  (int)kCOUNT == MI355Q_COUNT && (int)kCOUNT_IF == MI355Q_COUNT_IF && (int)kSUM_IF == MI355Q_SUM_IF,
              "mi355q_agg keeps the numeric values of SQLAgg");
static_assert((int)kEQ == MI355Q_EQ && (int)kNE == MI355Q_NE && (int)kLT == MI355Q_LT && (int)kGT == MI355Q_GT &&
                  (int)kLE == MI355Q_LE && (int)kGE == MI355Q_GE && (int)kISNULL == MI355Q_IS_NULL &&
                  (int)kISNOTNULL == MI355Q_IS_NOT_NULL,
              "mi355q_op keeps the numeric values of SQLOps");

inline int64_t int_literal(const SQLTypeInfo& ti, const Datum& d) {
  switch (ti.get_logical_size()) {
    case 1: return d.tinyintval;
    case 2: return d.smallintval;
    case 4: return d.intval;
    default: return d.bigintval;
  }
}

// x IN (c0, c1, ...) (Analyzer::InValues; CodeGenerator::codegen(InValues*), InValuesIR.cpp:23-72: the logical_or chain of
// `x = ci` comparisons, or a bitmap of the values with the same truth table): taken with non-NULL constants of the
// argument's own type, as the analyzer leaves them
inline std::vector<const Analyzer::Constant*> in_list_constants(const Analyzer::InValues* in) {
  std::vector<const Analyzer::Constant*> out;
  const int32_t t = logical_type(in->get_arg()->get_type_info());
  for (const auto& v : in->get_value_list()) {
    auto c = dynamic_cast<const Analyzer::Constant*>(v.get());
    if (!c || c->get_is_null() || logical_type(c->get_type_info()) != t) unsupported("IN list member");
    out.push_back(c);
  }
  if (out.empty()) unsupported("empty IN list");
  return out;
}

// contains_unsafe_division (LogicalIR.cpp:26-53) over the expression kinds this binding takes: a kDIVIDE whose divisor is
// not a constant, or is the NULL / a zero constant.  (The reference walks with Expr::find_expr; kMODULO is not looked at.)
inline bool contains_unsafe_division(const Analyzer::Expr* e) {
  if (auto b = dynamic_cast<const Analyzer::BinOper*>(e)) {
    if (b->get_optype() == kDIVIDE) {
      auto c = dynamic_cast<const Analyzer::Constant*>(b->get_right_operand());
      if (!c || c->get_is_null()) return true;
      const auto& ti = c->get_type_info();
      const Datum d = c->get_constval();
      if (ti.get_type() == kDOUBLE ? d.doubleval == 0.0 : ti.get_type() == kFLOAT ? d.floatval == 0.0f : int_literal(ti, d) == 0)
        return true;
    }
    return contains_unsafe_division(b->get_left_operand()) || contains_unsafe_division(b->get_right_operand());
  }
  if (auto u = dynamic_cast<const Analyzer::UOper*>(e)) return contains_unsafe_division(u->get_operand());
  if (auto in = dynamic_cast<const Analyzer::InValues*>(e)) return contains_unsafe_division(in->get_arg());
  if (auto ce = dynamic_cast<const Analyzer::CaseExpr*>(e)) {
    for (const auto& pr : ce->get_expr_pair_list())
      if (contains_unsafe_division(pr.first.get()) || contains_unsafe_division(pr.second.get())) return true;
    return ce->get_else_expr() && contains_unsafe_division(ce->get_else_expr());
  }
  return false;
}

// postfix program of a value expression over OUTER columns (CodeGenerator::codegenCast / codegenArith shapes).
// `outer_col` resolves a ColumnVar of the outer table to its position among the input columns.
inline void emit_expr(const Analyzer::Expr* e, mi355q_expr& x,
                      const std::function<int(const Analyzer::ColumnVar*)>& outer_col,
                      const std::function<int(const Analyzer::Expr*)>* hoisted = nullptr) {
  auto push = [&](int32_t op, int32_t type, int32_t arg, int64_t ilit, double flit, int32_t null_lit = 0) {
    if (x.n_nodes >= kGlueMaxExprNodes) throw ExprTooLong();
    x.nodes[x.n_nodes++] = mi355q_expr_node{op, type, arg, null_lit, ilit, flit};
  };
  if (hoisted) {  // a subtree the caller has stated as an earlier expression: its value
    const int at = (*hoisted)(e);
    if (at >= 0) {
      push(MI355Q_EX_COL, 0, at, 0, 0.0);
      return;
    }
  }
  if (auto cv = dynamic_cast<const Analyzer::ColumnVar*>(e)) {
    if (cv->get_rte_idx() != 0) unsupported("expression over an inner column");
    check_supported_type(cv->get_type_info());
    push(MI355Q_EX_COL, 0, outer_col(cv), 0, 0.0);
  } else if (auto c = dynamic_cast<const Analyzer::Constant*>(e)) {
    const auto& ti = c->get_type_info();
    const Datum d = c->get_constval();
    const int32_t t = logical_type(ti);
    if (c->get_is_null()) push(MI355Q_EX_LIT, t, 0, 0, 0.0, 1);  // the NULL constant (a CASE without ELSE)
    else if (t == MI355Q_DOUBLE) push(MI355Q_EX_LIT, t, 0, 0, d.doubleval);
    else if (t == MI355Q_FLOAT) push(MI355Q_EX_LIT, t, 0, 0, d.floatval);
    else push(MI355Q_EX_LIT, t, 0, int_literal(ti, d), 0.0);
  } else if (auto u = dynamic_cast<const Analyzer::UOper*>(e)) {
    emit_expr(u->get_operand(), x, outer_col, hoisted);
    switch (u->get_optype()) {
      case kCAST: {
        // DATE, TIME and TIMESTAMP(0) all live in 8-byte integers, but a cast between two of them (or to / from one of
        // another precision) TRUNCATES or RESCALES in the reference (codegenCastTimestampToDate / ...ToTime /
        // codegenCastBetweenTimestamps, CastIR.cpp:104-124): there is no micro-op for that, and an INT64 -> INT64
        // no-op would group CAST(ts AS DATE) by the wrong values.  Refused (the reference's own path runs the step).
        const SQLTypeInfo& from = u->get_operand()->get_type_info();
        const SQLTypeInfo& to = u->get_type_info();
        if ((from.is_time() || to.is_time()) &&
            (from.get_type() != to.get_type() || from.get_dimension() != to.get_dimension()))
          unsupported("cast between date / time types (or to / from one)");
        push(MI355Q_EX_CAST, logical_type(to), 0, 0, 0.0);
        break;
      }
      case kNOT:  // codegenLogical(UOper), LogicalIR.cpp:363-379
        if (!u->get_operand()->get_type_info().is_boolean()) unsupported("NOT over a value that is not a BOOLEAN");
        push(MI355Q_EX_NOT, MI355Q_INT8, 0, 0, 0.0);
        break;
      case kISNULL: push(MI355Q_EX_IS_NULL, MI355Q_INT8, 0, 0, 0.0); break;  // codegenIsNull, :381-432
      case kUMINUS: push(MI355Q_EX_UMINUS, logical_type(u->get_type_info()), 0, 0, 0.0); break;  // ArithmeticIR.cpp:787-838
      default: unsupported("unary operator");
    }
  } else if (auto b = dynamic_cast<const Analyzer::BinOper*>(e)) {
    const int32_t op = b->get_optype() == kPLUS ? MI355Q_EX_ADD : b->get_optype() == kMINUS ? MI355Q_EX_SUB
                       : b->get_optype() == kMULTIPLY ? MI355Q_EX_MUL : b->get_optype() == kDIVIDE ? MI355Q_EX_DIV
                       : b->get_optype() == kMODULO ? MI355Q_EX_MOD : 0;
    // a comparison of two VALUES (column vs column, column vs expression; CompareIR.cpp:230-330): a BOOLEAN, INT8 1 / 0 / NULL
    const int32_t cmp = b->get_optype() == kEQ ? MI355Q_EX_EQ : b->get_optype() == kNE ? MI355Q_EX_NE
                        : b->get_optype() == kLT ? MI355Q_EX_LT : b->get_optype() == kLE ? MI355Q_EX_LE
                        : b->get_optype() == kGT ? MI355Q_EX_GT : b->get_optype() == kGE ? MI355Q_EX_GE : 0;
    if (b->get_optype() == kAND || b->get_optype() == kOR) {
      // codegenLogical (LogicalIR.cpp:299-342).  Where an operand holds an unsafe division the reference emits the
      // short-circuit form, the unsafe operand evaluated second (codegenLogicalShortCircuit :197-297: rhs unsafe -> as
      // written; else lhs unsafe -> swapped).  Its other motive — LIKELY() / UNLIKELY() annotations on a heavy operand
      // (get_likelihood :79-117) — needs Analyzer::LikelihoodExpr, which this binding does not take.
      const Analyzer::Expr* first = b->get_left_operand();
      const Analyzer::Expr* second = b->get_right_operand();
      if (!first->get_type_info().is_boolean() || !second->get_type_info().is_boolean()) unsupported("AND / OR over values that are not BOOLEANs");
      int32_t sc = 0;
      if (contains_unsafe_division(second)) sc = 1;
      else if (contains_unsafe_division(first)) {
        sc = 1;
        std::swap(first, second);
      }
      emit_expr(first, x, outer_col, hoisted);
      emit_expr(second, x, outer_col, hoisted);
      push(b->get_optype() == kAND ? MI355Q_EX_AND : MI355Q_EX_OR, MI355Q_INT8, 0, 0, 0.0, sc);
      return;
    }
    if (!op && !cmp) unsupported("binary operator");
    if (cmp && logical_type(b->get_left_operand()->get_type_info()) != logical_type(b->get_right_operand()->get_type_info()))
      unsupported("comparison of two types");  // (the analyzer casts both sides to one type: CompareIR.cpp asserts it)
    emit_expr(b->get_left_operand(), x, outer_col, hoisted);
    emit_expr(b->get_right_operand(), x, outer_col, hoisted);
    push(op ? op : cmp, op ? logical_type(b->get_type_info()) : MI355Q_INT8, 0, 0, 0.0);
  } else if (auto in = dynamic_cast<const Analyzer::InValues*>(e)) {
    // (x = c0) OR (x = c1) OR ...: the plain form — every comparison is evaluated, as in the reference's loop
    const auto vals = in_list_constants(in);
    const int32_t t = logical_type(in->get_arg()->get_type_info());
    for (size_t i = 0; i < vals.size(); ++i) {
      emit_expr(in->get_arg(), x, outer_col, hoisted);
      const Datum d = vals[i]->get_constval();
      if (t == MI355Q_DOUBLE) push(MI355Q_EX_LIT, t, 0, 0, d.doubleval);
      else if (t == MI355Q_FLOAT) push(MI355Q_EX_LIT, t, 0, 0, d.floatval);
      else push(MI355Q_EX_LIT, t, 0, int_literal(vals[i]->get_type_info(), d), 0.0);
      push(MI355Q_EX_EQ, MI355Q_INT8, 0, 0, 0.0);
      if (i) push(MI355Q_EX_OR, MI355Q_INT8, 0, 0, 0.0);
    }
  } else if (auto ce = dynamic_cast<const Analyzer::CaseExpr*>(e)) {
    // CASE WHEN c0 THEN t0 WHEN c1 THEN t1 ... ELSE e END (CaseIR.cpp:67-140) = CASE(c0, t0, CASE(c1, t1, ... e)): the plan's
    // operand order is ELSE, THEN, condition, so the later WHENs are emitted first and the stack stays at <= 4 values
    const int32_t t = logical_type(ce->get_type_info());
    std::vector<std::pair<const Analyzer::Expr*, const Analyzer::Expr*>> whens;
    for (const auto& pr : ce->get_expr_pair_list()) whens.emplace_back(pr.first.get(), pr.second.get());
    if (whens.empty() || !ce->get_else_expr()) unsupported("CASE shape");
    std::function<void(size_t)> emit_from = [&](size_t i) {
      if (i == whens.size()) {
        emit_expr(ce->get_else_expr(), x, outer_col, hoisted);
        return;
      }
      emit_from(i + 1);
      emit_expr(whens[i].second, x, outer_col, hoisted);
      emit_expr(whens[i].first, x, outer_col, hoisted);
      push(MI355Q_EX_CASE, t, 0, 0, 0.0);
    };
    emit_from(0);
  } else {
    unsupported("expression kind");
  }
}

// the plan's own qual shapes: <value> <cmp> .., <value> IS NULL, NOT(<value> IS NULL)
inline bool qual_shaped(const Analyzer::Expr* e) {
  if (auto u = dynamic_cast<const Analyzer::UOper*>(e)) {
    if (u->get_optype() == kISNULL) return true;
    auto in = u->get_optype() == kNOT ? dynamic_cast<const Analyzer::UOper*>(u->get_operand()) : nullptr;
    return in && in->get_optype() == kISNULL;
  }
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  if (!b) return false;
  switch (b->get_optype()) {
    case kEQ: case kNE: case kLT: case kGT: case kLE: case kGE: return true;
    default: return false;
  }
}
// simple_quals / quals entry, or the condition of COUNT_IF / SUM_IF: <value> <cmp> <literal>, <value> IS NULL,
// NOT(<value> IS NULL).  `value_col` gives the outer column (or virtual expression column) of a value expression.
inline mi355q_qual translate_qual(const Analyzer::Expr* e, const std::function<int(const Analyzer::Expr*)>& value_col) {
  mi355q_qual q{};
  if (!qual_shaped(e) && e->get_type_info().is_boolean()) {
    // any other BOOLEAN (COUNT_IF(x > 5 AND y IS NULL), SUM_IF(v, k IN (1, 2)), a BOOLEAN column): the condition is a projected
    // expression and the qual `that column = 1` — TRUE; NULL is not (toBool, LogicalIR.cpp:344-352)
    q.col = value_col(e);
    q.op = MI355Q_EQ;
    q.ival = 1;
    return q;
  }
  if (auto u = dynamic_cast<const Analyzer::UOper*>(e)) {
    if (u->get_optype() == kISNULL) {
      q.col = value_col(u->get_operand());
      q.op = MI355Q_IS_NULL;
      return q;
    }
    if (u->get_optype() == kNOT) {  // RelAlgTranslator builds IS NOT NULL as NOT(ISNULL(x))
      auto in = dynamic_cast<const Analyzer::UOper*>(u->get_operand());
      if (in && in->get_optype() == kISNULL) {
        q.col = value_col(in->get_operand());
        q.op = MI355Q_IS_NOT_NULL;
        return q;
      }
    }
    unsupported("unary qual");
  }
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  auto lit = b ? dynamic_cast<const Analyzer::Constant*>(b->get_right_operand()) : nullptr;
  if (!b) unsupported("qual shape");
  switch (b->get_optype()) {  // SQLOps values are the ABI's (mi355q_op)
    case kEQ: case kNE: case kLT: case kGT: case kLE: case kGE: q.op = (int32_t)b->get_optype(); break;
    default: unsupported("comparison operator");
  }
  if (!lit || lit->get_is_null()) {
    // <value> <cmp> <value> (column vs column ...): the comparison itself is a projected BOOLEAN expression and the qual is
    // `that column = 1` — TRUE; a NULL comparison is not (toBool, LogicalIR.cpp:344-352)
    q.col = value_col(e);
    q.op = MI355Q_EQ;
    q.ival = 1;
    return q;
  }
  q.col = value_col(b->get_left_operand());
  const auto& ti = lit->get_type_info();
  check_supported_type(ti);
  const Datum d = lit->get_constval();
  if (ti.get_type() == kDOUBLE) q.fval = d.doubleval;
  else if (ti.get_type() == kFLOAT) q.fval = d.floatval;
  else q.ival = int_literal(ti, d);
  return q;
}

// Can evaluating `e` end the step (overflow, division by zero)?  Comparisons, logic, IS NULL, IN lists and CASEs over
// columns, literals, widening casts and floating-point + - * cannot: such a subtree may be evaluated for every row, whatever
// branch it stands in, without anything to observe — so it can move into an earlier expression from ANYWHERE.
inline bool cannot_raise(const Analyzer::Expr* e) {
  if (dynamic_cast<const Analyzer::ColumnVar*>(e) || dynamic_cast<const Analyzer::Constant*>(e)) return true;
  if (auto u = dynamic_cast<const Analyzer::UOper*>(e)) {
    const auto& from = u->get_operand()->get_type_info();
    const auto& to = u->get_type_info();
    switch (u->get_optype()) {
      case kNOT: case kISNULL: return cannot_raise(u->get_operand());
      case kUMINUS: return to.is_fp() && cannot_raise(u->get_operand());
      case kCAST:  // integer -> a wider integer or floating point, FLOAT -> DOUBLE
        if (from.is_fp() ? !(to.is_fp() && to.get_logical_size() >= from.get_logical_size())
                         : !(to.is_fp() || to.get_logical_size() >= from.get_logical_size()))
          return false;
        return cannot_raise(u->get_operand());
      default: return false;
    }
  }
  if (auto b = dynamic_cast<const Analyzer::BinOper*>(e)) {
    switch (b->get_optype()) {
      case kEQ: case kNE: case kLT: case kGT: case kLE: case kGE: case kAND: case kOR: break;
      case kPLUS: case kMINUS: case kMULTIPLY:
        if (!b->get_type_info().is_fp()) return false;
        break;
      default: return false;
    }
    return cannot_raise(b->get_left_operand()) && cannot_raise(b->get_right_operand());
  }
  if (auto in = dynamic_cast<const Analyzer::InValues*>(e)) return cannot_raise(in->get_arg());
  if (auto ce = dynamic_cast<const Analyzer::CaseExpr*>(e)) {
    for (const auto& pr : ce->get_expr_pair_list())
      if (!cannot_raise(pr.first.get()) || !cannot_raise(pr.second.get())) return false;
    return !ce->get_else_expr() || cannot_raise(ce->get_else_expr());
  }
  return false;
}
// the WHEN conditions of the CASEs under `e` that cannot raise (and are more than a column): what a CASE too long for one
// program states as earlier expressions (`CASE WHEN x BETWEEN 6 AND 7 THEN 1 WHEN x BETWEEN 8 AND 9 THEN 2 ELSE 3 END`,
// Tests/ExecuteTest.cpp:5358, has 19 nodes)
inline void hoistable_conditions(const Analyzer::Expr* e, std::vector<const Analyzer::Expr*>& out) {
  if (auto ce = dynamic_cast<const Analyzer::CaseExpr*>(e)) {
    for (const auto& pr : ce->get_expr_pair_list()) {
      if (cannot_raise(pr.first.get()) && !dynamic_cast<const Analyzer::ColumnVar*>(pr.first.get())) out.push_back(pr.first.get());
      else hoistable_conditions(pr.first.get(), out);
      hoistable_conditions(pr.second.get(), out);
    }
    if (ce->get_else_expr()) hoistable_conditions(ce->get_else_expr(), out);
  } else if (auto b = dynamic_cast<const Analyzer::BinOper*>(e)) {
    hoistable_conditions(b->get_left_operand(), out);
    hoistable_conditions(b->get_right_operand(), out);
  } else if (auto u = dynamic_cast<const Analyzer::UOper*>(e)) {
    hoistable_conditions(u->get_operand(), out);
  }
}

// A BOOLEAN program that does not fit kGlueMaxExprNodes (the reference's own `x > 6 AND x < 8 OR (z > 100 AND z < 103)`
// has 15 nodes): the operands of a PLAIN AND / OR at its root are evaluated whatever the row holds (codegenLogical
// :299-342 emits both), so each can be an expression of its own and the root reads their values (MI355Q_EX_COL n_cols + j).
// Not where the short-circuit form applies: there the second operand must stay unevaluated.  `operand_col` = value_col of the
// executor half (allocates — or finds — the expression of an Analyzer node, splitting it further the same way).
inline void flatten_plain_logic(const Analyzer::Expr* e, SQLOps op, std::vector<const Analyzer::Expr*>& out) {
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  if (b && b->get_optype() == op) {
    flatten_plain_logic(b->get_left_operand(), op, out);
    flatten_plain_logic(b->get_right_operand(), op, out);
    return;
  }
  out.push_back(e);
}
inline bool split_plain_logic(const Analyzer::Expr* e, mi355q_expr& x, const std::function<int(const Analyzer::ColumnVar*)>& outer_col,
                              const std::function<int(const Analyzer::Expr*)>& operand_col) {
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  if (!b || (b->get_optype() != kAND && b->get_optype() != kOR) || contains_unsafe_division(e)) return false;
  // `a OR b OR c` nested any way round: the operands of the whole chain (three-valued AND / OR are associative)
  std::vector<const Analyzer::Expr*> ops;
  flatten_plain_logic(e, b->get_optype(), ops);
  const int32_t op = b->get_optype() == kAND ? MI355Q_EX_AND : MI355Q_EX_OR;
  x = mi355q_expr{};
  for (size_t i = 0; i < ops.size(); ++i) {
    if (!ops[i]->get_type_info().is_boolean()) return false;
    // inline while the rest still fits as (value of an expression, op) pairs; else the operand becomes an expression of its own
    const int reserve = (i ? 1 : 0) + 2 * (int)(ops.size() - 1 - i);
    mi355q_expr tmp = x;
    bool inlined = false;
    try {
      emit_expr(ops[i], tmp, outer_col);
      inlined = tmp.n_nodes + reserve <= kGlueMaxExprNodes;
    } catch (const ExprTooLong&) {
    }
    if (inlined) {
      x = tmp;
    } else {
      const int c = operand_col(ops[i]);
      if (x.n_nodes + 1 + reserve > kGlueMaxExprNodes) throw ExprTooLong();
      x.nodes[x.n_nodes++] = mi355q_expr_node{MI355Q_EX_COL, 0, c, 0, 0, 0.0};
    }
    if (i) x.nodes[x.n_nodes++] = mi355q_expr_node{op, MI355Q_INT8, 0, 0, 0, 0.0};
  }
  return true;
}

// x IN (c0 .. c8) — or NOT IN — beyond one program (`y IN (43, 44, 45, 46, 47, 48, 49)`, Tests/ExecuteTest.cpp:2518: 27 nodes):
// runs of the list are BOOLEAN expressions of their own and the root ORs their values (the logical_or chain of the reference's
// loop is associative); `new_bool_col` of the executor half allocates an expression from a filler.
inline bool split_in_list(const Analyzer::Expr* e, mi355q_expr& x, const std::function<int(const Analyzer::ColumnVar*)>& outer_col,
                          const std::function<int(const std::function<void(mi355q_expr&, const std::function<int(const Analyzer::ColumnVar*)>&)>&)>& new_bool_col) {
  auto u = dynamic_cast<const Analyzer::UOper*>(e);
  const bool negated = u && u->get_optype() == kNOT;
  auto in = dynamic_cast<const Analyzer::InValues*>(negated ? u->get_operand() : e);
  if (!in) return false;
  const auto vals = in_list_constants(in);
  mi355q_expr probe{};
  emit_expr(in->get_arg(), probe, outer_col);
  const int per_value = probe.n_nodes + 2;                                      // the argument, the literal, EQ
  const size_t per_run = (size_t)((kGlueMaxExprNodes + 1) / (per_value + 1));  // k comparisons and k - 1 ORs
  if (per_run < 1) return false;
  std::vector<int> runs;
  for (size_t lo = 0; lo < vals.size(); lo += per_run) {
    const size_t hi = std::min(vals.size(), lo + per_run);
    runs.push_back(new_bool_col([&](mi355q_expr& c, const std::function<int(const Analyzer::ColumnVar*)>& oc) {
      for (size_t k = lo; k < hi; ++k) {
        emit_expr(in->get_arg(), c, oc);
        emit_expr(vals[k], c, oc);
        if (c.n_nodes + 1 + (k > lo ? 1 : 0) > kGlueMaxExprNodes) throw ExprTooLong();
        c.nodes[c.n_nodes++] = mi355q_expr_node{MI355Q_EX_EQ, MI355Q_INT8, 0, 0, 0, 0.0};
        if (k > lo) c.nodes[c.n_nodes++] = mi355q_expr_node{MI355Q_EX_OR, MI355Q_INT8, 0, 0, 0, 0.0};
      }
    }));
  }
  x = mi355q_expr{};
  if ((int)runs.size() * 2 - 1 + (negated ? 1 : 0) > kGlueMaxExprNodes) throw ExprTooLong();
  for (size_t j = 0; j < runs.size(); ++j) {
    x.nodes[x.n_nodes++] = mi355q_expr_node{MI355Q_EX_COL, 0, runs[j], 0, 0, 0.0};
    if (j) x.nodes[x.n_nodes++] = mi355q_expr_node{MI355Q_EX_OR, MI355Q_INT8, 0, 0, 0, 0.0};
  }
  if (negated) x.nodes[x.n_nodes++] = mi355q_expr_node{MI355Q_EX_NOT, MI355Q_INT8, 0, 0, 0, 0.0};
  return true;
}

// the deepest the evaluation stack of a program gets (the library refuses more than MI355Q_MAX_EXPR_STACK values with
// MI355Q_ERR_INVALID_PLAN; the binding says "unsupported" first, so that the caller keeps the native path)
inline int expr_stack_depth(const mi355q_expr& x) {
  int sp = 0, deepest = 0;
  for (int i = 0; i < x.n_nodes; ++i) {
    switch (x.nodes[i].op) {
      case MI355Q_EX_COL: case MI355Q_EX_LIT: ++sp; break;
      case MI355Q_EX_CAST: case MI355Q_EX_NOT: case MI355Q_EX_IS_NULL: case MI355Q_EX_UMINUS: break;
      case MI355Q_EX_CASE: sp -= 2; break;
      default: --sp;
    }
    if (sp > deepest) deepest = sp;
  }
  return deepest;
}
inline void check_expr_stack(const mi355q_expr& x) {
  if (expr_stack_depth(x) > MI355Q_MAX_EXPR_STACK) unsupported("expression too deep");
}

// ---- conjuncts of simple_quals / quals
// members of a disjunction of qual shapes (`a OR b OR ...`, Analyzer::BinOper kOR nested any way round; NOT over a comparison
// counts as the comparison); -1 = not such a disjunction
inline int disjunction_members(const Analyzer::Expr* e) {
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  if (b && b->get_optype() == kOR) {
    const int l = disjunction_members(b->get_left_operand()), r = disjunction_members(b->get_right_operand());
    return l < 0 || r < 0 ? -1 : l + r;
  }
  auto u = dynamic_cast<const Analyzer::UOper*>(e);
  if (u && u->get_optype() == kNOT)
    if (auto inner = dynamic_cast<const Analyzer::BinOper*>(u->get_operand())) {
      // (folding NOT into the operator is exact for integers — NULL stays "not TRUE" — but not for DOUBLE / FLOAT operands:
      // NOT(x < 5) is TRUE for a NaN, x >= 5 is not; those keep the NOT, as a BOOLEAN expression)
      return qual_shaped(inner) && !inner->get_left_operand()->get_type_info().is_fp() ? 1 : -1;
    }
  if (auto in = dynamic_cast<const Analyzer::InValues*>(e)) return (int)in_list_constants(in).size();  // x = c0 OR x = c1 ...
  return qual_shaped(e) ? 1 : -1;
}
// the members of a disjunction (group > 0), or one comparison (group 0); NOT over a comparison of integers is folded into the
// operator (NOT(x < 5) = x >= 5: NULL stays "not TRUE" either way, LogicalIR.cpp:299-352)
inline void emit_disjunction(const Analyzer::Expr* e, const std::function<int(const Analyzer::Expr*)>& value_col,
                             mi355q_qual* quals, int32_t* n_quals, int32_t group) {
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  if (b && b->get_optype() == kOR) {
    emit_disjunction(b->get_left_operand(), value_col, quals, n_quals, group);
    emit_disjunction(b->get_right_operand(), value_col, quals, n_quals, group);
    return;
  }
  if (auto in = dynamic_cast<const Analyzer::InValues*>(e)) {
    const int col = value_col(in->get_arg());
    for (const Analyzer::Constant* c : in_list_constants(in)) {
      mi355q_qual qi{};
      qi.col = col;
      const auto& ti = c->get_type_info();
      const Datum d = c->get_constval();
      if (ti.get_type() == kDOUBLE) qi.fval = d.doubleval;
      else if (ti.get_type() == kFLOAT) qi.fval = d.floatval;
      else qi.ival = int_literal(ti, d);
      if (*n_quals >= kGlueMaxQuals) unsupported("too many quals");
      qi.op = MI355Q_QUAL_IN_OR_GROUP(MI355Q_EQ, group);
      quals[(*n_quals)++] = qi;
    }
    return;
  }
  mi355q_qual q{};
  auto u = dynamic_cast<const Analyzer::UOper*>(e);
  auto inner = u && u->get_optype() == kNOT ? dynamic_cast<const Analyzer::BinOper*>(u->get_operand()) : nullptr;
  if (inner) {
    q = translate_qual(inner, value_col);
    switch (q.op) {
      case MI355Q_EQ: q.op = MI355Q_NE; break;
      case MI355Q_NE: q.op = MI355Q_EQ; break;
      case MI355Q_LT: q.op = MI355Q_GE; break;
      case MI355Q_GE: q.op = MI355Q_LT; break;
      case MI355Q_GT: q.op = MI355Q_LE; break;
      default: q.op = MI355Q_GT; break;  // NOT(<=)
    }
  } else {
    q = translate_qual(e, value_col);
  }
  if (*n_quals >= kGlueMaxQuals) unsupported("too many quals");
  q.op = MI355Q_QUAL_IN_OR_GROUP(q.op, group);
  quals[(*n_quals)++] = q;
}

// One conjunct -> plan quals: a top-level AND splits; a comparison is one qual; a disjunction of comparisons is a group of
// quals while the plan has room (kGlueMaxQuals, MI355Q_MAX_OR_GROUPS).  Every other BOOLEAN conjunct — AND inside OR, NOT
// over a disjunction, one disjunction too many — is ONE projected expression (the NOT / AND / OR / IS NULL micro-ops over its
// comparisons) and the qual `that column = 1`: TRUE; a NULL is not (toBool, LogicalIR.cpp:344-352).  What the plan cannot
// state either way (a program beyond kGlueMaxExprNodes, an expression kind outside the micro-ops) is refused by emit_expr.
inline void translate_conjunct(const Analyzer::Expr* e, const std::function<int(const Analyzer::Expr*)>& value_col,
                               mi355q_qual* quals, int32_t* n_quals, int32_t* n_groups) {
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  if (b && b->get_optype() == kAND) {
    translate_conjunct(b->get_left_operand(), value_col, quals, n_quals, n_groups);
    translate_conjunct(b->get_right_operand(), value_col, quals, n_quals, n_groups);
    return;
  }
  const int m = disjunction_members(e);
  if (m == 1) {
    emit_disjunction(e, value_col, quals, n_quals, 0);
    return;
  }
  if (m > 1 && *n_quals + m <= kGlueMaxQuals && *n_groups < MI355Q_MAX_OR_GROUPS) {
    emit_disjunction(e, value_col, quals, n_quals, ++*n_groups);
    return;
  }
  if (!e->get_type_info().is_boolean()) unsupported("qual shape");
  if (*n_quals >= kGlueMaxQuals) unsupported("too many quals");
  mi355q_qual q{};
  q.col = value_col(e);
  q.op = MI355Q_EQ;
  q.ival = 1;
  quals[(*n_quals)++] = q;
}

// ---- the WHERE clause as a whole: simple_quals and quals, a conjunction
// The reference does not evaluate every conjunct for every row: prioritizeQuals (LogicalIR.cpp:158-195) sets the quals that
// hold an unsafe division aside (should_defer_eval :55-77) and the row function evaluates them INSIDE the branch the primary
// quals open — `WHERE x > 7 AND y / (x - 7) < 44` (Tests/ExecuteTest.cpp:2021) never divides by zero.  The plan evaluates the
// expression of a qual for every row, so such a WHERE becomes ONE projected BOOLEAN expression: the safe conjuncts ANDed in the
// plain form, then each deferred conjunct through a short-circuit AND (the second operand's checks exist only where the first
// is TRUE), and the qual `that column = 1`.  A WHERE without unsafe divisions is translated conjunct by conjunct.
using OuterCol = std::function<int(const Analyzer::ColumnVar*)>;
using ExprFiller = std::function<void(mi355q_expr&, const OuterCol&)>;
inline void flatten_conjuncts(const Analyzer::Expr* e, std::vector<const Analyzer::Expr*>& out) {
  auto b = dynamic_cast<const Analyzer::BinOper*>(e);
  if (b && b->get_optype() == kAND) {
    flatten_conjuncts(b->get_left_operand(), out);
    flatten_conjuncts(b->get_right_operand(), out);
    return;
  }
  out.push_back(e);
}
inline void translate_where(const std::vector<const Analyzer::Expr*>& where, const std::function<int(const Analyzer::Expr*)>& value_col,
                            const std::function<int(const ExprFiller&)>& new_bool_col, mi355q_qual* quals, int32_t* n_quals,
                            int32_t* n_groups) {
  std::vector<const Analyzer::Expr*> all, primary, deferred;
  for (const Analyzer::Expr* e : where) flatten_conjuncts(e, all);
  for (const Analyzer::Expr* e : all) (contains_unsafe_division(e) ? deferred : primary).push_back(e);
  if (deferred.empty()) {
    // more conjuncts than the plan has quals (`x > 6 AND x < 8 AND z > 100 AND z < 102 AND t > 1000 AND t < 1002`,
    // ExecuteTest.cpp:1907): the first ones as quals, the rest ANDed into one BOOLEAN expression `= 1`
    auto needs = [](const Analyzer::Expr* e) {
      const int m = disjunction_members(e);
      return m < 1 ? 1 : m;
    };
    int total = *n_quals;
    for (const Analyzer::Expr* e : primary) total += needs(e);
    size_t i = 0;
    if (total > kGlueMaxQuals)
      for (int used = *n_quals; i < primary.size() && used + needs(primary[i]) <= kGlueMaxQuals - 1; ++i) used += needs(primary[i]);
    else
      i = primary.size();
    for (size_t k = 0; k < i; ++k) translate_conjunct(primary[k], value_col, quals, n_quals, n_groups);
    if (i == primary.size()) return;
    primary.erase(primary.begin(), primary.begin() + (long)i);
  }
  const int col = new_bool_col([&](mi355q_expr& x, const OuterCol& outer_col) {
    auto push_and = [&](int32_t short_circuit) {
      if (x.n_nodes >= kGlueMaxExprNodes) throw ExprTooLong();
      x.nodes[x.n_nodes++] = mi355q_expr_node{MI355Q_EX_AND, MI355Q_INT8, 0, short_circuit, 0, 0.0};
    };
    int n = 0;
    for (const Analyzer::Expr* e : primary) {
      if (!e->get_type_info().is_boolean()) unsupported("qual shape");
      emit_expr(e, x, outer_col);
      if (n++) push_and(0);
    }
    // the deferred conjuncts are ALL evaluated once the primary ones pass (compileBody ANDs them inside sc_true): plain
    // ANDs among themselves, ONE short-circuit AND behind the primary operand — `x > 0 AND a / b > 1 AND c / d > 1` with
    // a / b > 1 FALSE and d = 0 raises DIV_BY_ZERO here as it does there
    int m = 0;
    for (const Analyzer::Expr* e : deferred) {
      if (!e->get_type_info().is_boolean()) unsupported("qual shape");
      emit_expr(e, x, outer_col);
      if (m++) push_and(0);
    }
    if (n && m) push_and(1);
  });
  if (*n_quals >= kGlueMaxQuals) unsupported("too many quals");
  mi355q_qual q{};
  q.col = col;
  q.op = MI355Q_EQ;
  q.ival = 1;
  quals[(*n_quals)++] = q;
}

// an aggregate of target_exprs (get_target_info, Shared/TargetInfo.h:48-56).  `inner_col` resolves a ColumnVar of the
// inner table; COUNT_IF(cond): the argument IS the condition; SUM_IF(value, cond): the condition is arg1
// (RelAlgTranslator.cpp:348-360).
inline mi355q_target translate_agg(const Analyzer::AggExpr* agg, const std::function<int(const Analyzer::Expr*)>& value_col,
                                   const std::function<int(const Analyzer::ColumnVar*)>& inner_col) {
  mi355q_target tg{};
  tg.col = -1;
  if (agg->get_is_distinct()) unsupported("DISTINCT aggregate");
  tg.agg = agg_kind(agg->get_aggtype());
  const Analyzer::Expr* arg = agg->get_arg();
  if (tg.agg == MI355Q_COUNT_IF) {
    if (!arg) unsupported("COUNT_IF without a condition");
    tg.cond = translate_qual(arg, value_col);
    return tg;
  }
  if (tg.agg == MI355Q_SUM_IF) {
    const auto cond = agg->get_arg1();
    if (!arg || !cond) unsupported("SUM_IF shape");
    tg.cond = translate_qual(cond.get(), value_col);
  }
  if (arg) {
    auto cv = dynamic_cast<const Analyzer::ColumnVar*>(arg);
    if (cv && cv->get_rte_idx() != 0) {
      check_supported_type(cv->get_type_info());
      tg.table = 1;
      tg.col = inner_col(cv);
    } else {
      tg.col = value_col(arg);
    }
  }
  return tg;
}

}  // namespace mi355q_glue
