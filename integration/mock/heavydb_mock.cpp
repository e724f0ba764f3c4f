// heavydb_mock.cpp — the little behaviour the mock needs: getExpressionRange over the mock catalog.
// (QueryEngine/ExpressionRange.cpp in a HeavyDB build: a column's bounds come from chunk metadata, a cast to a
// floating-point type keeps the operand's bounds as doubles, + - * combine the operands' bounds.)
#include "heavydb_mock.h"

#include <algorithm>

bool g_bigint_count = false;

ExpressionRange getExpressionRange(const Analyzer::Expr* expr, const std::vector<InputTableInfo>& query_infos,
                                   const Executor* executor) {
  if (auto cv = dynamic_cast<const Analyzer::ColumnVar*>(expr)) {
    auto it = executor->column_ranges.find({cv->getColumnKey().table_id, cv->getColumnKey().column_id});
    return it == executor->column_ranges.end() ? ExpressionRange::makeInvalidRange() : it->second;
  }
  if (auto c = dynamic_cast<const Analyzer::Constant*>(expr)) {
    const auto& ti = c->get_type_info();
    const Datum d = c->get_constval();
    if (ti.is_fp()) {
      const double v = ti.get_type() == kDOUBLE ? d.doubleval : d.floatval;
      return ExpressionRange::makeDoubleRange(v, v, false);
    }
    const int64_t v = ti.get_logical_size() == 1 ? d.tinyintval : ti.get_logical_size() == 2 ? d.smallintval
                      : ti.get_logical_size() == 4 ? d.intval : d.bigintval;
    return ExpressionRange::makeIntRange(v, v, 0, false);
  }
  if (auto u = dynamic_cast<const Analyzer::UOper*>(expr)) {
    const ExpressionRange r = getExpressionRange(u->get_operand(), query_infos, executor);
    if (r.getType() == ExpressionRangeType::Invalid || u->get_optype() != kCAST) return ExpressionRange::makeInvalidRange();
    if (u->get_type_info().is_fp()) {
      return r.getType() == ExpressionRangeType::Integer
                 ? ExpressionRange::makeDoubleRange((double)r.getIntMin(), (double)r.getIntMax(), r.hasNulls())
                 : r;
    }
    return r.getType() == ExpressionRangeType::Integer ? r : ExpressionRange::makeInvalidRange();
  }
  if (auto b = dynamic_cast<const Analyzer::BinOper*>(expr)) {
    const ExpressionRange l = getExpressionRange(b->get_left_operand(), query_infos, executor);
    const ExpressionRange r = getExpressionRange(b->get_right_operand(), query_infos, executor);
    if (l.getType() != ExpressionRangeType::Integer || r.getType() != ExpressionRangeType::Integer)
      return ExpressionRange::makeInvalidRange();
    const bool nul = l.hasNulls() || r.hasNulls();
    const __int128 a0 = l.getIntMin(), a1 = l.getIntMax(), b0 = r.getIntMin(), b1 = r.getIntMax();
    __int128 lo, hi;
    switch (b->get_optype()) {
      case kPLUS: lo = a0 + b0; hi = a1 + b1; break;
      case kMINUS: lo = a0 - b1; hi = a1 - b0; break;
      case kMULTIPLY: {
        const __int128 c[4] = {a0 * b0, a0 * b1, a1 * b0, a1 * b1};
        lo = *std::min_element(c, c + 4);
        hi = *std::max_element(c, c + 4);
        break;
      }
      default: return ExpressionRange::makeInvalidRange();
    }
    if (lo < INT64_MIN || hi > INT64_MAX) return ExpressionRange::makeInvalidRange();
    return ExpressionRange::makeIntRange((int64_t)lo, (int64_t)hi, 0, nul);
  }
  return ExpressionRange::makeInvalidRange();
}
