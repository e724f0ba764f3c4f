// real_headers_check.cpp — the type-mapping half of the binding (integration/Mi355qTranslate.h) compiled against the
// REFERENCE'S OWN Shared/sqltypes.h, Shared/sqldefs.h and Analyzer/Analyzer.h (not the mock), and run on real Analyzer
// objects.  Built by tests/test_integration_glue.py where /root/reference exists:
//   g++ -std=c++17 -DNO_BOOST -include oracle/_stubs/ref_layout_prelude.h -Ioracle/_stubs -I/root/reference
//       -I/root/reference/QueryEngine -Iinclude -Iintegration real_headers_check.cpp analyzer_link_stubs.cpp
// Also asserts that integration/mock/heavydb_mock.h states the reference's enum values (mock_enum_values.inc is
// generated from the mock by the test).
#include <cassert>
#include <cstdio>
#include <memory>

#include "Mi355qTranslate.h"

#ifdef MI355Q_CHECK_MOCK_ENUMS
#include "mock_enum_values.inc"  // static_asserts: <mock value> == <reference enumerator>
#endif

using namespace mi355q_glue;

#define REQ(c) do { if (!(c)) { std::printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

template <typename F>
static bool refuses(F&& f) {
  try {
    f();
  } catch (const std::runtime_error&) {
    return true;
  }
  return false;
}

int main() {
  using Analyzer::AggExpr;
  using Analyzer::BinOper;
  using Analyzer::ColumnVar;
  using Analyzer::Constant;
  using Analyzer::UOper;
  // ---- column types: SQL type + encoding -> storage / logical type of the plan ABI
  SQLTypeInfo t_int(kINT, false), t_big(kBIGINT, true), t_dbl(kDOUBLE, false), t_flt(kFLOAT, false);
  SQLTypeInfo t_small_fixed(kINT, false);
  t_small_fixed.set_compression(kENCODING_FIXED);
  t_small_fixed.set_comp_param(16);
  t_small_fixed.set_fixed_size();
  SQLTypeInfo t_date_days(kDATE, false);
  t_date_days.set_compression(kENCODING_DATE_IN_DAYS);
  t_date_days.set_comp_param(32);
  t_date_days.set_fixed_size();
  SQLTypeInfo t_dict(kTEXT, false);
  t_dict.set_compression(kENCODING_DICT);
  t_dict.set_comp_param(8);
  t_dict.set_size(1);  // what DDL does for TEXT ENCODING DICT(8) (Catalog/Catalog.cpp:3348, :4311: set_size(comp_param / 8));
                       // set_fixed_size() -> get_storage_size() would say 4 for every dictionary width (sqltypes.h:1428-1431)
  REQ(storage_type(t_int) == MI355Q_INT32 && logical_type(t_int) == MI355Q_INT32);
  REQ(storage_type(t_big) == MI355Q_INT64 && storage_type(t_dbl) == MI355Q_DOUBLE && storage_type(t_flt) == MI355Q_FLOAT);
  REQ(storage_type(t_small_fixed) == MI355Q_INT16 && logical_type(t_small_fixed) == MI355Q_INT32);
  REQ(encoding_of(t_small_fixed) == MI355Q_ENC_FIXED && encoding_of(t_int) == MI355Q_ENC_NONE);
  REQ(storage_type(t_date_days) == MI355Q_INT32 && logical_type(t_date_days) == MI355Q_INT64 &&
      encoding_of(t_date_days) == MI355Q_ENC_DATE_IN_DAYS);
  REQ(storage_type(t_dict) == MI355Q_INT8 && encoding_of(t_dict) == MI355Q_ENC_DICT);
  // refused: DECIMAL, none-encoded strings, arrays, TIMESTAMP(6), geo
  SQLTypeInfo t_dec(kDECIMAL, 10, 2, false);
  SQLTypeInfo t_str(kTEXT, false);
  SQLTypeInfo t_arr(kARRAY, false);
  SQLTypeInfo t_ts6(kTIMESTAMP, 6, 0, false);
  SQLTypeInfo t_pt(kPOINT, false);
  REQ(refuses([&] { storage_type(t_dec); }) && refuses([&] { storage_type(t_str); }) && refuses([&] { logical_type(t_arr); }) &&
      refuses([&] { storage_type(t_ts6); }) && refuses([&] { storage_type(t_pt); }));
  // ---- aggregates
  REQ(agg_kind(kAVG) == MI355Q_AVG && agg_kind(kCOUNT) == MI355Q_COUNT && agg_kind(kCOUNT_IF) == MI355Q_COUNT_IF &&
      agg_kind(kSUM_IF) == MI355Q_SUM_IF);
  REQ(refuses([] { agg_kind(kAPPROX_COUNT_DISTINCT); }) && refuses([] { agg_kind(kMODE); }));
  // ---- expressions over real Analyzer objects: CAST(x AS DOUBLE), x + 1, y * 3 - x
  const shared::ColumnKey kx{1, 7, 3}, ky{1, 7, 4}, kw{1, 9, 2};
  auto x = std::make_shared<ColumnVar>(t_int, kx, 0);
  auto y = std::make_shared<ColumnVar>(t_big, ky, 0);
  auto w = std::make_shared<ColumnVar>(t_big, kw, 1);  // inner table
  auto outer_col = [&](const ColumnVar* cv) { return cv->getColumnKey() == kx ? 0 : cv->getColumnKey() == ky ? 1 : -1; };
  auto inner_col = [&](const ColumnVar* cv) { return cv->getColumnKey() == kw ? 0 : -1; };
  auto value_col = [&](const Analyzer::Expr* e) {
    auto cv = dynamic_cast<const ColumnVar*>(e);
    return cv ? outer_col(cv) : 100;  // 100: "a projected expression" (the executor half allocates those)
  };
  {
    auto cast = std::make_shared<UOper>(t_dbl, false, kCAST, x);
    mi355q_expr e{};
    emit_expr(cast.get(), e, outer_col);
    REQ(e.n_nodes == 2 && e.nodes[0].op == MI355Q_EX_COL && e.nodes[0].arg == 0 && e.nodes[1].op == MI355Q_EX_CAST &&
        e.nodes[1].type == MI355Q_DOUBLE);
  }
  {  // CAST(ts AS DATE) / CAST(d AS TIMESTAMP) / TIMESTAMP(0) -> TIMESTAMP(3): truncating / rescaling casts in the reference
     // (CastIR.cpp:104-124), all INT64 -> INT64 by logical type: refused, not lowered to a no-op
    const SQLTypeInfo t_ts(kTIMESTAMP, 0, 0, false), t_date(kDATE, false), t_ts3(kTIMESTAMP, 3, 0, false);
    auto ts = std::make_shared<ColumnVar>(t_ts, kx, 0);
    int refused = 0;
    for (const SQLTypeInfo& to : {t_date, t_ts3, t_big}) {
      auto cast = std::make_shared<UOper>(to, false, kCAST, ts);
      mi355q_expr e{};
      try {
        emit_expr(cast.get(), e, outer_col);
      } catch (const std::runtime_error&) {
        ++refused;
      }
    }
    REQ(refused == 3);
  }
  {
    Datum one;
    one.intval = 1;
    auto lit = std::make_shared<Constant>(t_int, false, one);
    auto add = std::make_shared<BinOper>(t_int, false, kPLUS, kONE, x, lit);
    mi355q_expr e{};
    emit_expr(add.get(), e, outer_col);
    REQ(e.n_nodes == 3 && e.nodes[1].op == MI355Q_EX_LIT && e.nodes[1].ilit == 1 && e.nodes[1].type == MI355Q_INT32 &&
        e.nodes[2].op == MI355Q_EX_ADD && e.nodes[2].type == MI355Q_INT32);
    Datum three;
    three.bigintval = 3;
    auto l3 = std::make_shared<Constant>(t_big, false, three);
    auto mul = std::make_shared<BinOper>(t_big, false, kMULTIPLY, kONE, y, l3);
    auto xb = std::make_shared<UOper>(t_big, false, kCAST, x);
    auto sub = std::make_shared<BinOper>(t_big, false, kMINUS, kONE, mul, xb);
    mi355q_expr e2{};
    emit_expr(sub.get(), e2, outer_col);
    REQ(e2.n_nodes == 6 && e2.nodes[5].op == MI355Q_EX_SUB && e2.nodes[2].op == MI355Q_EX_MUL && e2.nodes[4].op == MI355Q_EX_CAST &&
        e2.nodes[4].type == MI355Q_INT64);
    auto div = std::make_shared<BinOper>(t_big, false, kDIVIDE, kONE, y, l3);
    auto mod = std::make_shared<BinOper>(t_big, false, kMODULO, kONE, div, l3);
    mi355q_expr e3{};
    emit_expr(mod.get(), e3, outer_col);
    REQ(e3.n_nodes == 5 && e3.nodes[2].op == MI355Q_EX_DIV && e3.nodes[4].op == MI355Q_EX_MOD && e3.nodes[4].type == MI355Q_INT64);
    auto band = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kAND, kONE, y, l3);
    mi355q_expr e5{};
    REQ(refuses([&] { emit_expr(band.get(), e5, outer_col); }));
    mi355q_expr e4{};
    REQ(refuses([&] { emit_expr(w.get(), e4, outer_col); }));  // an inner column is not a projectable value
    // ---- quals: x < 1, y IS NULL, NOT(y IS NULL)
    auto lt = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kLT, kONE, x, lit);
    mi355q_qual q = translate_qual(lt.get(), value_col);
    REQ(q.op == MI355Q_LT && q.col == 0 && q.ival == 1);
    auto isnull = std::make_shared<UOper>(SQLTypeInfo(kBOOLEAN, true), false, kISNULL, y);
    q = translate_qual(isnull.get(), value_col);
    REQ(q.op == MI355Q_IS_NULL && q.col == 1);
    auto notnull = std::make_shared<UOper>(SQLTypeInfo(kBOOLEAN, true), false, kNOT, isnull);
    q = translate_qual(notnull.get(), value_col);
    REQ(q.op == MI355Q_IS_NOT_NULL && q.col == 1);
    // x < 1 OR y IS NULL OR NOT(x >= 1)  -> three quals of one group; AND of that with another comparison
    auto ge = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kGE, kONE, x, lit);
    auto not_ge = std::make_shared<UOper>(SQLTypeInfo(kBOOLEAN, false), false, kNOT, ge);
    auto or1 = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, true), false, kOR, kONE, lt, isnull);
    auto or2 = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, true), false, kOR, kONE, or1, not_ge);
    auto both = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, true), false, kAND, kONE, or2, ge);
    mi355q_qual qs[MI355Q_MAX_QUALS];
    int32_t nq = 0, ng = 0;
    translate_conjunct(both.get(), value_col, qs, &nq, &ng);
    REQ(nq == 4 && ng == 1);
    REQ(MI355Q_QUAL_OP(qs[0].op) == MI355Q_LT && MI355Q_QUAL_OR_GROUP(qs[0].op) == 1);
    REQ(MI355Q_QUAL_OP(qs[1].op) == MI355Q_IS_NULL && MI355Q_QUAL_OR_GROUP(qs[1].op) == 1 && qs[1].col == 1);
    REQ(MI355Q_QUAL_OP(qs[2].op) == MI355Q_LT && MI355Q_QUAL_OR_GROUP(qs[2].op) == 1);   // NOT(x >= 1) folded
    REQ(qs[3].op == MI355Q_GE);
    // AND inside OR has no qual shape: the whole conjunct is ONE projected BOOLEAN expression, the qual `that column = 1`
    auto and_in_or = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, true), false, kOR, kONE, both, lt);
    nq = ng = 0;
    translate_conjunct(and_in_or.get(), value_col, qs, &nq, &ng);
    REQ(nq == 1 && ng == 0 && qs[0].op == MI355Q_EQ && qs[0].ival == 1 && qs[0].col == 100);
    {
      // (x < 1 AND y IS NULL) OR NOT(x >= 1) as a program: COL LIT LT | COL IS_NULL | AND | COL LIT GE NOT | OR
      auto conj = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kAND, kONE, lt, isnull);
      auto disj = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kOR, kONE, conj, not_ge);
      mi355q_expr eb{};
      emit_expr(disj.get(), eb, outer_col);
      REQ(eb.n_nodes == 11 && eb.nodes[2].op == MI355Q_EX_LT && eb.nodes[4].op == MI355Q_EX_IS_NULL && eb.nodes[4].type == MI355Q_INT8 &&
          eb.nodes[5].op == MI355Q_EX_AND && eb.nodes[5].reserved == 0 && eb.nodes[8].op == MI355Q_EX_GE &&
          eb.nodes[9].op == MI355Q_EX_NOT && eb.nodes[10].op == MI355Q_EX_OR && eb.nodes[10].type == MI355Q_INT8);
      nq = ng = 0;
      translate_conjunct(disj.get(), value_col, qs, &nq, &ng);
      REQ(nq == 1 && ng == 0 && qs[0].op == MI355Q_EQ && qs[0].ival == 1 && qs[0].col == 100);
      // y <> 0 AND x / y > 1 over BIGINTs: the division's divisor is a column -> the short-circuit form, the unsafe operand
      // second; written the other way round the operands are swapped (codegenLogicalShortCircuit, LogicalIR.cpp:197-297)
      Datum zero;
      zero.bigintval = 0;
      auto l0 = std::make_shared<Constant>(t_big, false, zero);
      auto ne0 = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kNE, kONE, y, l0);
      auto quot = std::make_shared<BinOper>(t_big, false, kDIVIDE, kONE, xb, y);
      auto gt1 = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kGT, kONE, quot, l3);
      REQ(contains_unsafe_division(gt1.get()) && !contains_unsafe_division(ne0.get()));
      auto by_l3 = std::make_shared<BinOper>(t_big, false, kDIVIDE, kONE, y, l3);   // a non-zero constant divisor is safe
      auto by_l0 = std::make_shared<BinOper>(t_big, false, kDIVIDE, kONE, y, l0);   // a zero constant is not
      REQ(!contains_unsafe_division(by_l3.get()) && contains_unsafe_division(by_l0.get()));
      for (int swapped = 0; swapped < 2; ++swapped) {
        auto guarded = swapped ? std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kAND, kONE, gt1, ne0)
                               : std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kAND, kONE, ne0, gt1);
        mi355q_expr eg{};
        emit_expr(guarded.get(), eg, outer_col);
        REQ(eg.n_nodes == 10 && eg.nodes[2].op == MI355Q_EX_NE && eg.nodes[6].op == MI355Q_EX_DIV && eg.nodes[8].op == MI355Q_EX_GT &&
            eg.nodes[9].op == MI355Q_EX_AND && eg.nodes[9].reserved == 1);
      }
      // WHERE x < 1 AND x / y > 3 ... (ExecuteTest.cpp:2021's shape): the conjunct with the unsafe division is a DEFERRED qual in
      // the reference (prioritizeQuals), here the second operand of a short-circuit AND inside one BOOLEAN column
      {
        int filled = 0;
        mi355q_expr ew{};
        auto new_bool_col = [&](const ExprFiller& fill) {
          fill(ew, outer_col);
          ++filled;
          return 200;
        };
        nq = ng = 0;
        translate_where({gt1.get(), lt.get()}, value_col, new_bool_col, qs, &nq, &ng);   // (written unsafe-first: still deferred)
        REQ(filled == 1 && nq == 1 && ng == 0 && qs[0].col == 200 && qs[0].op == MI355Q_EQ && qs[0].ival == 1);
        REQ(ew.n_nodes == 10 && ew.nodes[2].op == MI355Q_EX_LT && ew.nodes[6].op == MI355Q_EX_DIV && ew.nodes[8].op == MI355Q_EX_GT &&
            ew.nodes[9].op == MI355Q_EX_AND && ew.nodes[9].reserved == 1);
        filled = 0;
        nq = ng = 0;
        translate_where({or1.get(), ge.get()}, value_col, new_bool_col, qs, &nq, &ng);   // no unsafe division: conjunct by conjunct
        REQ(filled == 0 && nq == 3 && ng == 1 && qs[2].op == MI355Q_GE);
      }
      // y IN (3, 0): two quals of one OR group; as a value, (y = 3) OR (y = 0); a list member of another type is refused
      {
        using Analyzer::InValues;
        InValues in_y(y, {l3, l0});
        nq = ng = 0;
        translate_conjunct(&in_y, value_col, qs, &nq, &ng);
        REQ(nq == 2 && ng == 1 && MI355Q_QUAL_OP(qs[0].op) == MI355Q_EQ && MI355Q_QUAL_OR_GROUP(qs[0].op) == 1 && qs[0].col == 1 &&
            qs[0].ival == 3 && qs[1].ival == 0 && MI355Q_QUAL_OR_GROUP(qs[1].op) == 1);
        mi355q_expr ei{};
        emit_expr(&in_y, ei, outer_col);
        REQ(ei.n_nodes == 7 && ei.nodes[2].op == MI355Q_EX_EQ && ei.nodes[5].op == MI355Q_EX_EQ && ei.nodes[6].op == MI355Q_EX_OR &&
            ei.nodes[6].reserved == 0 && ei.nodes[4].ilit == 0);
        InValues in_one(y, {l3});
        nq = ng = 0;
        translate_conjunct(&in_one, value_col, qs, &nq, &ng);
        REQ(nq == 1 && ng == 0 && qs[0].op == MI355Q_EQ && qs[0].ival == 3);
        InValues in_mixed(y, {lit});   // an INT constant against a BIGINT argument: the analyzer would have cast it
        REQ(refuses([&] { translate_conjunct(&in_mixed, value_col, qs, &nq, &ng); }));
      }
      // NOT(d < 1.5) over a DOUBLE column is not folded into d >= 1.5 (a NaN passes the first and fails the second): it stays
      // a BOOLEAN expression `= 1`
      {
        auto dcol = std::make_shared<ColumnVar>(t_dbl, ky, 0);
        Datum dv;
        dv.doubleval = 1.5;
        auto dl = std::make_shared<Constant>(t_dbl, false, dv);
        auto dlt = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kLT, kONE, dcol, dl);
        auto not_dlt = std::make_shared<UOper>(SQLTypeInfo(kBOOLEAN, false), false, kNOT, dlt);
        nq = ng = 0;
        translate_conjunct(not_dlt.get(), value_col, qs, &nq, &ng);
        REQ(nq == 1 && qs[0].col == 100 && qs[0].op == MI355Q_EQ && qs[0].ival == 1);
        mi355q_expr ed{};
        emit_expr(not_dlt.get(), ed, outer_col);
        REQ(ed.n_nodes == 4 && ed.nodes[2].op == MI355Q_EX_LT && ed.nodes[3].op == MI355Q_EX_NOT);
      }
      // -y, y IS NULL as values
      auto neg = std::make_shared<UOper>(t_big, false, kUMINUS, y);
      mi355q_expr en{};
      emit_expr(neg.get(), en, outer_col);
      REQ(en.n_nodes == 2 && en.nodes[1].op == MI355Q_EX_UMINUS && en.nodes[1].type == MI355Q_INT64);
      auto not_y = std::make_shared<UOper>(SQLTypeInfo(kBOOLEAN, false), false, kNOT, y);   // NOT over a BIGINT
      mi355q_expr ex{};
      REQ(refuses([&] { emit_expr(not_y.get(), ex, outer_col); }));
    }
    // column-vs-column compare: the comparison is a projected BOOLEAN expression, the qual `that column = 1`
    auto ge_cols = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kGE, kONE, xb, y);
    q = translate_qual(ge_cols.get(), value_col);
    REQ(q.op == MI355Q_EQ && q.ival == 1 && q.col == 100);      // (this check's value_col: 100 = "an expression column")
    mi355q_expr ec{};
    emit_expr(ge_cols.get(), ec, outer_col);
    REQ(ec.n_nodes == 4 && ec.nodes[1].op == MI355Q_EX_CAST && ec.nodes[3].op == MI355Q_EX_GE && ec.nodes[3].type == MI355Q_INT8);
    auto mixed = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kGE, kONE, x, y);   // INT vs BIGINT without the analyzer's cast
    mi355q_expr em{};
    REQ(refuses([&] { emit_expr(mixed.get(), em, outer_col); }));
    // CASE WHEN y <> 3 THEN y WHEN x_as_bigint >= y THEN y ELSE NULL END
    {
      using Analyzer::CaseExpr;
      auto ne3 = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, true), false, kNE, kONE, y, l3);
      Datum nul;
      nul.bigintval = 0;
      auto null_big = std::make_shared<Constant>(t_big, true, nul);
      std::list<std::pair<std::shared_ptr<Analyzer::Expr>, std::shared_ptr<Analyzer::Expr>>> whens{{ne3, y}, {ge_cols, y}};
      CaseExpr ce(t_big, false, whens, null_big);
      mi355q_expr ex{};
      emit_expr(&ce, ex, outer_col);
      // ELSE NULL | THEN y | cond (x::bigint >= y) | CASE | THEN y | cond y <> 3 | CASE
      REQ(ex.n_nodes == 12 && ex.nodes[0].op == MI355Q_EX_LIT && ex.nodes[0].reserved == 1 && ex.nodes[0].type == MI355Q_INT64);
      REQ(ex.nodes[ex.n_nodes - 1].op == MI355Q_EX_CASE && ex.nodes[ex.n_nodes - 1].type == MI355Q_INT64);
      REQ(ex.nodes[6].op == MI355Q_EX_CASE && ex.nodes[5].op == MI355Q_EX_GE && ex.nodes[ex.n_nodes - 2].op == MI355Q_EX_NE);
    }
    // ---- aggregates: SUM(y), COUNT(*), SUM(dim.w), COUNT_IF(x < 1), SUM_IF(y, x < 1), COUNT(DISTINCT x)
    AggExpr sum_y(t_big, kSUM, y, false, nullptr);
    mi355q_target tg = translate_agg(&sum_y, value_col, inner_col);
    REQ(tg.agg == MI355Q_SUM && tg.col == 1 && tg.table == 0);
    AggExpr count_star(SQLTypeInfo(kINT, true), kCOUNT, nullptr, false, nullptr);
    tg = translate_agg(&count_star, value_col, inner_col);
    REQ(tg.agg == MI355Q_COUNT && tg.col == -1);
    AggExpr sum_w(t_big, kSUM, w, false, nullptr);
    tg = translate_agg(&sum_w, value_col, inner_col);
    REQ(tg.agg == MI355Q_SUM && tg.table == 1 && tg.col == 0);
    AggExpr count_if(SQLTypeInfo(kINT, true), kCOUNT_IF, lt, false, nullptr);
    tg = translate_agg(&count_if, value_col, inner_col);
    REQ(tg.agg == MI355Q_COUNT_IF && tg.col == -1 && tg.cond.op == MI355Q_LT && tg.cond.col == 0 && tg.cond.ival == 1);
    {  // COUNT_IF(x < 1 AND y IS NULL): a condition without a qual shape is a projected BOOLEAN column `= 1`
      auto conj2 = std::make_shared<BinOper>(SQLTypeInfo(kBOOLEAN, false), false, kAND, kONE, lt, isnull);
      AggExpr count_if2(SQLTypeInfo(kINT, true), kCOUNT_IF, conj2, false, nullptr);
      tg = translate_agg(&count_if2, value_col, inner_col);
      REQ(tg.agg == MI355Q_COUNT_IF && tg.col == -1 && tg.cond.col == 100 && tg.cond.op == MI355Q_EQ && tg.cond.ival == 1);
    }
    AggExpr sum_if(t_big, kSUM_IF, y, false, lt);
    tg = translate_agg(&sum_if, value_col, inner_col);
    REQ(tg.agg == MI355Q_SUM_IF && tg.col == 1 && tg.cond.op == MI355Q_LT && tg.cond.col == 0);
    AggExpr count_distinct(SQLTypeInfo(kINT, true), kCOUNT, x, true, nullptr);
    REQ(refuses([&] { translate_agg(&count_distinct, value_col, inner_col); }));
  }
  std::printf("real_headers_check ok\n");
  return 0;
}
