"""Query-shape matrix shared by the CPU (oracle vs host-emulated row logic) and GPU (HIP vs
oracle) parity tests.  Shapes follow the reference's hot-path suites in Tests/ExecuteTest.cpp:
Select.FilterAndSimpleAggregation (:1885), GroupBy (:2587), FilterAndGroupBy (:2815),
GroupByBoundariesAndNull (:2874), GroupByKeylessAndNotKeyless (:3158), GroupByPerfectHash
(:11414), GroupByBaselineHash (:11487), Joins_InnerJoin_TwoTables (:12852),
Joins_DifferentIntegerTypes (:12710) — restated as hand-built execution units the way
Tests/GroupByTest.cpp:73-151 does.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from heavydb_amd import capi
from heavydb_amd.capi import (COUNT_IF, SUM_IF, AVG, COUNT, DOUBLE, EQ, GE, GT, INT8, INT16, INT32, INT64, LE, LT,
                              MAX, MIN, NE, PROJECT_KEY, SUM)
from heavydb_amd.executor import (Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit,
                                  TargetExpr)

NP = {INT8: np.int8, INT16: np.int16, INT32: np.int32, INT64: np.int64, DOUBLE: np.float64, capi.FLOAT: np.float32}
NULLS = {INT8: -2**7, INT16: -2**15, INT32: -2**31, INT64: -2**63, DOUBLE: 2.2250738585072014e-308,
         capi.FLOAT: float(np.finfo(np.float32).tiny)}


@dataclass
class Case:
    name: str
    ra: RelAlgExecutionUnit
    frags: List[List[np.ndarray]]                 # [frag][col]
    inner: List[np.ndarray] = field(default_factory=list)
    join_keys: object = None                      # inner key column (or a list of them: composite key)
    join_key_type: object = INT64
    join_range: Optional[ExpressionRange] = None
    join_prefer_baseline: bool = False
    join_one_to_many: int = 0                     # 0 OneToOne only, 1 rebuild on duplicates, 2 OneToMany
    join_key_nullable: object = False             # bool, or a list for composite keys
    expect_error: Optional[int] = None
    fp_rtol: float = 1e-9                         # fp64 SUM/AVG rel <= 1e-9


def col_range(arrs: List[np.ndarray], t: int, nullable: bool) -> ExpressionRange:
    """What getExpressionRange derives from chunk metadata: min/max over non-NULL values."""
    a = np.concatenate(arrs) if arrs else np.zeros(0, NP[t])
    if nullable:
        mask = a != NP[t](NULLS[t])
        has_nulls = bool((~mask).any())
        a = a[mask]
    else:
        has_nulls = False
    if a.size == 0:
        return ExpressionRange(True, 0, -1, has_nulls)  # empty range: min > max
    if t in (DOUBLE, capi.FLOAT):
        return ExpressionRange(True, 0, 0, has_nulls, float(a.min()), float(a.max()))
    return ExpressionRange(True, int(a.min()), int(a.max()), has_nulls)


def expr_values(e: Expr, descs, cols, prior=()):
    """The expression over whole columns with numpy (test infrastructure: ranges of the virtual columns, the
    way getExpressionRange would bound them, and SQLite's input): returns (values, null mask, type).  Integer
    arithmetic is done in Python ints so that an overflow of the node's type is visible (-> None entries)."""
    st = []
    for nd in e.nodes:
        if nd.op == capi.EX_COL and nd.arg >= len(descs):   # the value of an earlier expression of the plan
            j = nd.arg - len(descs)
            st.append(expr_values(prior[j], descs, cols, prior[:j]))
        elif nd.op == capi.EX_COL:
            d = descs[nd.arg]
            a = np.asarray(cols[nd.arg])
            if d.type in (DOUBLE, capi.FLOAT):
                null = (a == NP[d.type](NULLS[d.type])) if d.nullable else np.zeros(len(a), bool)
                st.append((a.astype(np.float64) if d.type == DOUBLE else a.astype(np.float32), null, d.type))
            else:
                null = (a == NULLS[d.type]) if d.nullable else np.zeros(len(a), bool)
                st.append((a.astype(object), null, d.type))
        elif nd.op == capi.EX_LIT:
            n = len(np.asarray(cols[0])) if len(cols) else (len(st[0][0]) if st else 1)
            v = nd.flit if nd.type in (DOUBLE, capi.FLOAT) else nd.ilit
            arr = np.full(n, v, dtype=np.float64 if nd.type == DOUBLE else np.float32 if nd.type == capi.FLOAT else object)
            st.append((arr, np.full(n, bool(nd.null_lit)), nd.type))
        elif nd.op == capi.EX_CAST:
            v, null, t = st.pop()
            if nd.type in (DOUBLE, capi.FLOAT):
                r = np.array([float(x) for x in v], dtype=np.float64 if nd.type == DOUBLE else np.float32)
            elif t in (DOUBLE, capi.FLOAT):
                r = np.array([int(x + (-0.5 if x < 0 else 0.5)) if np.isfinite(x) else 0 for x in v], dtype=object)
            else:
                r = v
            st.append((r, null, nd.type))
        elif nd.op == capi.EX_NOT:
            v, null, _ = st.pop()
            st.append((np.array([0 if x else 1 for x in v], dtype=object), null, capi.INT8))
        elif nd.op in (capi.EX_AND, capi.EX_OR):   # three-valued (a filter / CASE only asks "is it TRUE", where both forms agree)
            (b, bn, _), (a, an, _) = st.pop(), st.pop()
            at, bt = np.array([bool(x) for x in a]) & ~an, np.array([bool(x) for x in b]) & ~bn
            af, bf = ~np.array([bool(x) for x in a]) & ~an, ~np.array([bool(x) for x in b]) & ~bn
            if nd.op == capi.EX_AND:
                val, known = at & bt, (at & bt) | af | bf
            else:
                val, known = at | bt, at | bt | (af & bf)
            st.append((np.array([int(x) for x in val], dtype=object), ~known, capi.INT8))
        elif nd.op == capi.EX_IS_NULL:
            v, null, _ = st.pop()
            st.append((np.array([int(x) for x in null], dtype=object), np.zeros(len(null), bool), capi.INT8))
        elif nd.op == capi.EX_UMINUS:
            v, null, t = st.pop()
            st.append((-v, null, t))
        elif nd.op == capi.EX_CASE:   # stack: ELSE, THEN, condition
            (c, cn, _), (t, tn, _), (e, en, _) = st.pop(), st.pop(), st.pop()
            take = (~cn) & np.array([bool(x) for x in c])
            st.append((np.where(take, t, e), np.where(take, tn, en), nd.type))
        elif capi.EX_EQ <= nd.op <= capi.EX_GE:
            (b, bn, _), (a, an, _) = st.pop(), st.pop()
            f = {capi.EX_EQ: lambda x, y: x == y, capi.EX_NE: lambda x, y: x != y, capi.EX_LT: lambda x, y: x < y,
                 capi.EX_LE: lambda x, y: x <= y, capi.EX_GT: lambda x, y: x > y, capi.EX_GE: lambda x, y: x >= y}[nd.op]
            st.append((np.array([int(f(x, y)) for x, y in zip(a, b)], dtype=object), an | bn, capi.INT8))
        else:
            (b, bn, _), (a, an, _) = st.pop(), st.pop()
            if nd.op in (capi.EX_DIV, capi.EX_MOD):
                if nd.type in (DOUBLE, capi.FLOAT):
                    with np.errstate(divide="ignore", invalid="ignore"):
                        r = a / b
                else:  # truncating quotient, remainder with the dividend's sign; a zero divisor errors at run time (0 here)
                    def one(x, y):
                        if y == 0:
                            return 0
                        q = abs(x) // abs(y) * (1 if (x < 0) == (y < 0) else -1)
                        return q if nd.op == capi.EX_DIV else x - q * y
                    r = np.array([one(int(x), int(y)) for x, y in zip(a, b)], dtype=object)
            else:
                r = a + b if nd.op == capi.EX_ADD else a - b if nd.op == capi.EX_SUB else a * b
            st.append((r, an | bn, nd.type))
    return st[0]


def expr_range(e: Expr, descs, frags, prior=()) -> ExpressionRange:
    cols = [np.concatenate([f[c] for f in frags]) for c in range(len(descs))] if frags else []
    if not frags or len(cols[0]) == 0:
        return ExpressionRange(True, 0, -1)
    v, null, t = expr_values(e, descs, cols, prior)
    ok = v[~null]
    if len(ok) == 0:
        return ExpressionRange(True, 0, -1, bool(null.any()))
    if t in (DOUBLE, capi.FLOAT):
        return ExpressionRange(True, 0, 0, bool(null.any()), float(ok.min()), float(ok.max()))
    lo, hi = int(min(ok)), int(max(ok))
    # an expression whose exact range leaves its type overflows at run time; the declared range stays inside
    tmin, tmax = NULLS[t], -NULLS[t] - 1
    return ExpressionRange(True, max(lo, tmin), min(hi, tmax), bool(null.any()))


def split(arr: np.ndarray, sizes: List[int]) -> List[np.ndarray]:
    out, o = [], 0
    for s in sizes:
        out.append(np.ascontiguousarray(arr[o:o + s]))
        o += s
    assert o == len(arr)
    return out


def with_nulls(rng, a: np.ndarray, t: int, frac: float) -> np.ndarray:
    a = a.copy()
    a[rng.random(len(a)) < frac] = NP[t](NULLS[t])
    return a


def make_table(rng, n: int, frag_sizes: List[int], spec: List[tuple]):
    """spec: [(type, nullable, generator(rng, n) -> array)].  Returns (descs, frags)."""
    cols = []
    for t, nullable, gen in spec:
        a = np.asarray(gen(rng, n)).astype(NP[t])
        if nullable:
            a = with_nulls(rng, a, t, 0.07)
        cols.append(a)
    descs = [InputColDescriptor(t, nullable, col_range([c], t, nullable))
             for (t, nullable, _), c in zip(spec, cols)]
    per_col = [split(c, frag_sizes) for c in cols]
    frags = [[per_col[c][f] for c in range(len(cols))] for f in range(len(frag_sizes))]
    return descs, frags


def build_cases(seed: int = 1234, scale: int = 1) -> List[Case]:
    rng = np.random.default_rng(seed)
    cases: List[Case] = []
    n = 20000 * scale
    fs = [n // 4 + 3, n // 4 - 3, n // 2 - 5, 5]  # ragged fragments incl. a tiny one
    assert sum(fs) == n

    # columns: 0 i32 uniform, 1 i64 small-range key, 2 i64 value, 3 f64 value,
    #          4 i64 sparse key, 5 i8, 6 i16 nullable, 7 i32 nullable, 8 i64 nullable,
    #          9 f64 nullable, 10 i32 small key nullable
    spec = [
        (INT32, False, lambda r, m: r.integers(0, 2**31 - 1, m)),
        (INT64, False, lambda r, m: r.integers(0, 100, m)),
        (INT64, False, lambda r, m: r.integers(-500000, 500001, m)),
        (DOUBLE, False, lambda r, m: r.random(m) * 1000.0),
        (INT64, False, lambda r, m: r.integers(0, 3000, m) * 1000003 + 7),
        (INT8, False, lambda r, m: r.integers(-100, 100, m)),
        (INT16, True, lambda r, m: r.integers(-3000, 3000, m)),
        (INT32, True, lambda r, m: r.integers(-10**6, 10**6, m)),
        (INT64, True, lambda r, m: r.integers(-10**12, 10**12, m)),
        (DOUBLE, True, lambda r, m: r.normal(0, 100, m)),
        (INT32, True, lambda r, m: r.integers(5, 40, m)),
    ]
    descs, frags = make_table(rng, n, fs, spec)

    def ra(targets, quals=(), group=(), guess=16384):
        return RelAlgExecutionUnit(list(descs), list(targets), list(quals), list(group),
                                   max_groups_buffer_entry_guess=guess)

    # ---- Select.FilterAndSimpleAggregation
    cases.append(Case("count_star_filter_i32_lt", ra([TargetExpr(COUNT)], [Qual(0, LT, 2**30)]), frags))
    for op, lit in [(LE, 10**9), (GT, 10**9), (GE, 2**30), (EQ, int(frags[0][0][5])), (NE, int(frags[0][0][5]))]:
        cases.append(Case(f"count_star_filter_i32_op{op}", ra([TargetExpr(COUNT)], [Qual(0, op, lit)]), frags))
    cases.append(Case("count_star_filter_i64", ra([TargetExpr(COUNT)], [Qual(2, GT, 0)]), frags))
    cases.append(Case("count_star_nofilter", ra([TargetExpr(COUNT)]), frags))
    cases.append(Case("simple_aggs_all", ra([TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(MIN, 2),
                                             TargetExpr(MAX, 2), TargetExpr(AVG, 2), TargetExpr(SUM, 3),
                                             TargetExpr(AVG, 3)]), frags))
    cases.append(Case("simple_aggs_minmax_f64", ra([TargetExpr(MIN, 3), TargetExpr(MAX, 3),
                                                    TargetExpr(MIN, 5), TargetExpr(MAX, 5)],
                                                   [Qual(3, LT, 500.0)]), frags))
    cases.append(Case("simple_aggs_nullable", ra([TargetExpr(COUNT, 7), TargetExpr(SUM, 7), TargetExpr(AVG, 8),
                                                  TargetExpr(MIN, 6), TargetExpr(MAX, 9), TargetExpr(SUM, 9),
                                                  TargetExpr(COUNT, 9), TargetExpr(AVG, 9)],
                                                 [Qual(7, GT, -500000), Qual(0, LT, 2**30)]), frags))
    cases.append(Case("simple_aggs_empty_result", ra([TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(MIN, 3),
                                                      TargetExpr(AVG, 2), TargetExpr(MAX, 7)],
                                                     [Qual(0, LT, -5)]), frags))  # all NULL but COUNT
    cases.append(Case("filter_on_nullable_eq", ra([TargetExpr(COUNT), TargetExpr(SUM, 8)],
                                                  [Qual(10, NE, 17)]), frags))   # NULL <> 17 is not true

    # ---- IS [NOT] NULL quals and constrained_not_null (OutputBufferInitialization.cpp:287,301-324;
    # ExecuteTest FilterAndGroupBy: `... WHERE x IS NOT NULL GROUP BY ...`): a grouped aggregate whose
    # argument is constrained starts from the NOT NULL init value and uses the plain agg functions
    IS_NULL, IS_NOT_NULL = capi.IS_NULL, capi.IS_NOT_NULL
    cases.append(Case("isnotnull_count_nongrouped", ra([TargetExpr(COUNT), TargetExpr(COUNT, 8), TargetExpr(SUM, 8)],
                                                       [Qual(8, IS_NOT_NULL)]), frags))
    cases.append(Case("isnull_count_nongrouped", ra([TargetExpr(COUNT), TargetExpr(COUNT, 8), TargetExpr(MIN, 7)],
                                                    [Qual(8, IS_NULL)]), frags))
    cases.append(Case("isnull_on_notnull_col_is_false", ra([TargetExpr(COUNT), TargetExpr(SUM, 2)],
                                                           [Qual(2, IS_NULL)]), frags))
    cases.append(Case("isnotnull_on_notnull_col_is_true", ra([TargetExpr(COUNT), TargetExpr(SUM, 2)],
                                                             [Qual(2, IS_NOT_NULL)]), frags))
    cases.append(Case("isnotnull_f64_nongrouped", ra([TargetExpr(COUNT), TargetExpr(AVG, 9), TargetExpr(MAX, 9)],
                                                     [Qual(9, IS_NOT_NULL)]), frags))
    cases.append(Case("constrained_perfect_sum_min_max", ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 8), TargetExpr(MIN, 8),
                                                            TargetExpr(MAX, 8), TargetExpr(COUNT, 8), TargetExpr(AVG, 8)],
                                                           [Qual(8, IS_NOT_NULL)], [1]), frags))
    cases.append(Case("constrained_perfect_sum_keyless_rule", ra([TargetExpr(SUM, 7), TargetExpr(COUNT)],
                                                                 [Qual(7, IS_NOT_NULL)], [1]), frags))
    cases.append(Case("constrained_other_column_unaffected", ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 7), TargetExpr(MIN, 8)],
                                                                [Qual(8, IS_NOT_NULL)], [1]), frags))
    cases.append(Case("constrained_f64_perfect", ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 9), TargetExpr(MIN, 9), TargetExpr(AVG, 9)],
                                                    [Qual(9, IS_NOT_NULL), Qual(0, LT, 2**30)], [1]), frags))
    cases.append(Case("constrained_baseline_avg_min", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 8),
                                                          TargetExpr(MIN, 8)], [Qual(8, IS_NOT_NULL)], [4], guess=8192), frags))
    cases.append(Case("isnull_grouped_baseline", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 8), TargetExpr(MAX, 2)],
                                                    [Qual(8, IS_NULL)], [4], guess=8192), frags))
    cases.append(Case("isnotnull_nullable_group_key", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 2)],
                                                         [Qual(10, IS_NOT_NULL)], [10]), frags))

    # ---- GroupBy / GroupByPerfectHash / keyless vs keyed
    cases.append(Case("perfect_key_sum_projectkey", ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 2)], group=[1]), frags))
    cases.append(Case("perfect_sum_only_keyed", ra([TargetExpr(SUM, 2)], group=[1]), frags))       # range spans 0 -> keyed
    cases.append(Case("perfect_count_keyless", ra([TargetExpr(COUNT), TargetExpr(SUM, 2)], group=[1]), frags))
    cases.append(Case("perfect_avg_keyless_idx1", ra([TargetExpr(AVG, 3), TargetExpr(MIN, 2)], group=[1]), frags))
    cases.append(Case("perfect_minmax_f64_i64", ra([TargetExpr(MIN, 3), TargetExpr(MAX, 3), TargetExpr(MIN, 2),
                                                    TargetExpr(MAX, 2), TargetExpr(COUNT)], group=[1]), frags))
    cases.append(Case("perfect_filtered", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 3)],
                                             [Qual(0, LT, 2**30)], [1]), frags))
    cases.append(Case("perfect_int8_key", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 5)],
                                             group=[5]), frags))
    # ---- GroupByBoundariesAndNull: nullable key (NULL group = max+1), nullable args
    cases.append(Case("perfect_nullable_key", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 2)],
                                                 group=[10]), frags))
    cases.append(Case("perfect_nullable_args", ra([TargetExpr(COUNT, 7), TargetExpr(SUM, 7), TargetExpr(AVG, 9),
                                                   TargetExpr(MIN, 8), TargetExpr(MAX, 6), TargetExpr(SUM, 9)],
                                                  group=[1]), frags))
    cases.append(Case("perfect_nullable_key_and_args", ra([TargetExpr(MIN, 9), TargetExpr(MAX, 7), TargetExpr(AVG, 6)],
                                                          [Qual(2, GE, -400000)], [10]), frags))

    # ---- GroupByBaselineHash
    cases.append(Case("baseline_count_avg", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 3)],
                                              group=[4], guess=8192), frags))
    cases.append(Case("baseline_filtered_count_avg", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 3)],
                                                        [Qual(0, LT, 2**30)], [4], guess=8192), frags))
    cases.append(Case("baseline_sum_min_max_i64", ra([TargetExpr(SUM, 2), TargetExpr(MIN, 2), TargetExpr(MAX, 2),
                                                      TargetExpr(COUNT)], group=[4], guess=6001), frags))
    cases.append(Case("baseline_nullable_args", ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 8), TargetExpr(AVG, 7),
                                                    TargetExpr(MIN, 9), TargetExpr(COUNT, 6)], group=[4], guess=7000),
                      frags))
    cases.append(Case("baseline_nullable_i64_key", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT)], group=[8],
                                                      guess=3 * n), frags))
    cases.append(Case("baseline_exact_fit", ra([TargetExpr(COUNT)], group=[4], guess=3000), frags))  # 100 % fill
    cases.append(Case("baseline_out_of_slots", ra([TargetExpr(COUNT)], group=[4], guess=1000), frags,
                      expect_error=-1))

    # baseline with a 4-byte compact key: wide-ranged int32 values (fits int32, too wide for perfect)
    descs32, frags32 = make_table(rng, n, fs, [
        (INT32, False, lambda r, m: r.integers(0, 2000, m) * 1000003 % (2**31 - 3)),
        (INT64, False, lambda r, m: r.integers(1, 10**6, m)),
        (DOUBLE, False, lambda r, m: r.random(m)),
    ])
    cases.append(Case("baseline_key32_compact",
                      RelAlgExecutionUnit(descs32, [TargetExpr(PROJECT_KEY), TargetExpr(SUM, 1), TargetExpr(AVG, 2)],
                                          groupby_exprs=[0], max_groups_buffer_entry_guess=5000), frags32))

    # ---- FLOAT arguments: single-precision slot arithmetic on the low 4 bytes (takes_float_argument)
    FL = capi.FLOAT
    dfl, ffl = make_table(rng, n, fs, [
        (INT64, False, lambda r, m: r.integers(0, 60, m)),                      # perfect key
        (FL, False, lambda r, m: r.random(m) * 100.0 + 1.0),                    # positive floats
        (FL, True, lambda r, m: r.random(m) * 50.0 - 10.0),                     # nullable, spans 0
        (INT64, False, lambda r, m: r.integers(0, 2500, m) * 1000003 + 7),      # baseline key
        (INT32, False, lambda r, m: r.integers(0, 2**31 - 1, m)),
    ])

    def fra(targets, quals=(), group=(), guess=16384):
        return RelAlgExecutionUnit(list(dfl), list(targets), list(quals), list(group),
                                   max_groups_buffer_entry_guess=guess)
    cases.append(Case("float_nongrouped_aggs", fra([TargetExpr(SUM, 1), TargetExpr(MIN, 1), TargetExpr(MAX, 1),
                                                    TargetExpr(AVG, 1), TargetExpr(COUNT, 2), TargetExpr(SUM, 2),
                                                    TargetExpr(MIN, 2), TargetExpr(AVG, 2)],
                                                   [Qual(1, LT, 80.5)]), ffl))
    cases.append(Case("float_nongrouped_all_null", fra([TargetExpr(SUM, 2), TargetExpr(MAX, 1), TargetExpr(AVG, 2),
                                                        TargetExpr(COUNT)], [Qual(4, LT, -1)]), ffl))
    cases.append(Case("float_perfect_keyed", fra([TargetExpr(SUM, 2), TargetExpr(MAX, 2), TargetExpr(MIN, 1)], group=[0]),
                      ffl))                    # SUM(nullable) / MAX(has nulls) / MIN(float quirk): keyed
    cases.append(Case("float_perfect_keyless_sum", fra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 1), TargetExpr(AVG, 2),
                                                        TargetExpr(MAX, 1)], group=[0]), ffl))  # SUM > 0 -> keyless
    cases.append(Case("float_baseline", fra([TargetExpr(PROJECT_KEY), TargetExpr(AVG, 1), TargetExpr(MIN, 2),
                                             TargetExpr(SUM, 2), TargetExpr(COUNT, 2), TargetExpr(MAX, 1),
                                             TargetExpr(SUM_IF, 1, cond=Qual(2, GT, 0.0))],
                                            [Qual(4, LT, 2**30)], group=[3], guess=8192), ffl))

    # ---- 4-byte slots (pick_target_compact_width): one group column, COUNT(*) and projections of
    # keys of at most 4 bytes only, <= UINT32_MAX input tuples, g_bigint_count off
    cases.append(Case("compact_perfect_nullable_int32_key", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT)], group=[10]),
                      frags))                                          # keyless: [key i32][count u32]
    cases.append(Case("compact_perfect_int8_key_filtered", ra([TargetExpr(COUNT), TargetExpr(PROJECT_KEY)],
                                                              [Qual(0, LT, 2**30)], group=[5]), frags))
    cases.append(Case("compact_perfect_count_only_int64_key", ra([TargetExpr(COUNT)], group=[1]), frags))
    cases.append(Case("compact_baseline_count_only", ra([TargetExpr(COUNT)], [Qual(2, GT, 0)], group=[4], guess=8192),
                      frags))                                          # baseline never narrows: key 8 B + count 8 B
    c8 = Case("compact_off_bigint_count", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT)], group=[10]), frags)
    c8.ra.bigint_count = True
    cases.append(c8)
    c9 = Case("compact_off_many_tuples", ra([TargetExpr(COUNT)], group=[4], guess=8192), frags)
    c9.ra.num_tuples = 2**32
    cases.append(c9)
    cases.append(Case("compact_off_int64_projection", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT)], group=[1]), frags))

    # ---- 4-byte value columns (plain INT / nullable INT) through the perfect and baseline layouts
    cases.append(Case("perfect_int32_values", ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 0), TargetExpr(MAX, 0),
                                                  TargetExpr(AVG, 0)], group=[1]), frags))
    cases.append(Case("baseline_int32_nullable_values", ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 7), TargetExpr(COUNT, 7),
                                                            TargetExpr(MIN, 7), TargetExpr(AVG, 7)], group=[4], guess=8192),
                      frags))

    # ---- conditional aggregates (COUNT_IF / SUM_IF, ExecuteTest.cpp Select.ConditionalAggregate shapes)
    CI = lambda c, op, lit: TargetExpr(COUNT_IF, cond=Qual(c, op, lit))            # noqa: E731
    SI = lambda v, c, op, lit: TargetExpr(SUM_IF, v, cond=Qual(c, op, lit))        # noqa: E731
    cases.append(Case("cond_aggs_nongrouped", ra([TargetExpr(COUNT), CI(0, LT, 2**30), CI(7, GT, 0), CI(9, LE, 0.0),
                                                  SI(2, 0, GE, 2**30), SI(3, 7, LT, 0), SI(8, 10, EQ, 17),
                                                  SI(9, 3, GT, 500.0)]), frags))       # nullable cond / value columns
    cases.append(Case("cond_aggs_perfect_keyed", ra([CI(0, LT, 2**30), TargetExpr(SUM, 2), SI(3, 7, GE, 0)],
                                                    group=[1]), frags))               # COUNT_IF first -> not keyless
    cases.append(Case("cond_aggs_perfect_keyless", ra([TargetExpr(COUNT), CI(7, NE, 5), SI(8, 0, LT, 10**9)],
                                                      [Qual(2, GT, -400000)], group=[10]), frags))
    cases.append(Case("cond_aggs_baseline", ra([TargetExpr(PROJECT_KEY), CI(6, GT, 0), SI(2, 5, LT, 0),
                                                SI(9, 9, GT, 0.0), TargetExpr(COUNT)], group=[4], guess=8192), frags))
    cases.append(Case("cond_aggs_multi_col", ra([TargetExpr(PROJECT_KEY, 1), TargetExpr(PROJECT_KEY, 0), CI(0, LT, 2**29), SI(3, 0, GE, 2**29)],
                                                group=[4, 1], guess=3 * n), frags))

    # ---- multi-column group by (GroupBy tests with several keys, ExecuteTest.cpp:2587-2873;
    # multi-column perfect hash :11414-11486, baseline :11487-11530)
    K0, K1 = TargetExpr(PROJECT_KEY, 0), TargetExpr(PROJECT_KEY, 1)
    cases.append(Case("multi_perfect_2col_keyed", ra([K0, K1, TargetExpr(SUM, 2)], group=[1, 5]),
                      frags))                                   # 100 x 200 entries, SUM spans 0 -> keyed
    cases.append(Case("multi_perfect_2col_keyless", ra([TargetExpr(COUNT), TargetExpr(AVG, 3)], group=[1, 5]), frags))
    cases.append(Case("multi_perfect_nullable_translate", ra([K0, K1, TargetExpr(SUM, 2), TargetExpr(COUNT, 7)],
                                                             [Qual(0, LT, 2**30)], group=[10, 1]), frags))
    cases.append(Case("multi_perfect_keyless_nullable", ra([TargetExpr(COUNT), TargetExpr(MAX, 9)], group=[1, 10]),
                      frags))
    cases.append(Case("multi_perfect_3col", ra([TargetExpr(PROJECT_KEY, 2), K0, K1, TargetExpr(COUNT),
                                                TargetExpr(SUM, 2)], group=[5, 10, 1]), frags))  # 200 x 36 x 100
    cases.append(Case("multi_baseline_i64_2col", ra([K0, K1, TargetExpr(COUNT), TargetExpr(AVG, 3)], group=[4, 1],
                                                    guess=3 * n), frags))
    cases.append(Case("multi_baseline_nullable_i64", ra([K1, K0, TargetExpr(SUM, 2), TargetExpr(MIN, 9)],
                                                        [Qual(0, GE, 2**29)], group=[8, 10], guess=3 * n), frags))
    cases.append(Case("multi_baseline_4col", ra([K0, K1, TargetExpr(PROJECT_KEY, 2), TargetExpr(PROJECT_KEY, 3),
                                                 TargetExpr(COUNT), TargetExpr(MAX, 2)], group=[4, 6, 5, 1],
                                                guess=4 * n), frags))
    cases.append(Case("multi_baseline_out_of_slots", ra([TargetExpr(COUNT)], group=[4, 1], guess=2000), frags,
                      expect_error=-1))
    # 4-byte components: every key range is a valid int32 range (pick_baseline_key_width)
    d4, f4 = make_table(rng, n, fs, [
        (INT32, False, lambda r, m: r.integers(0, 300, m) * 1000003 % (2**31 - 3)),
        (INT16, True, lambda r, m: r.integers(-20, 20, m)),
        (INT64, False, lambda r, m: r.integers(10**6, 10**6 + 5, m)),   # int64 column, int32-sized range
        (DOUBLE, False, lambda r, m: r.random(m)),
        (INT64, True, lambda r, m: r.integers(-1000, 1000, m)),
    ])
    cases.append(Case("multi_baseline_key32_2col",
                      RelAlgExecutionUnit(d4, [K0, K1, TargetExpr(COUNT), TargetExpr(AVG, 3)],
                                          groupby_exprs=[0, 1], max_groups_buffer_entry_guess=3 * n), f4))
    cases.append(Case("multi_baseline_key32_3col_padded",
                      RelAlgExecutionUnit(d4, [TargetExpr(PROJECT_KEY, 2), K1, K0, TargetExpr(SUM, 4),
                                               TargetExpr(MIN, 3)],
                                          groupby_exprs=[0, 1, 2], max_groups_buffer_entry_guess=4 * n), f4))

    # ---- encoded columns (kENCODING_FIXED / _DICT / _DATE_IN_DAYS): decoders + NULL widening
    def enc_table():
        m = n
        fixed16 = rng.integers(-3000, 3000, m).astype(np.int16)             # BIGINT stored in 16 bits
        fixed16[rng.random(m) < 0.07] = np.int16(-2**15)
        fixed32 = (rng.integers(0, 500, m) * 4000003 - 10**9).astype(np.int32)  # BIGINT in 32 bits, NOT NULL
        dict8 = rng.integers(0, 200, m).astype(np.uint8)                    # string ids, 1 byte, nullable
        dict8[rng.random(m) < 0.05] = np.uint8(255)
        dict16 = rng.integers(0, 40000, m).astype(np.uint16)                # 2-byte ids, NOT NULL (ids > 32767!)
        date32 = rng.integers(18000, 18012, m).astype(np.int32)             # DATE in days, nullable
        date32[rng.random(m) < 0.05] = np.int32(-2**31)
        date16 = rng.integers(-50, 50, m).astype(np.int16)                  # DATE in 16-bit days
        val = rng.integers(-10**6, 10**6, m).astype(np.int64)
        raw = [fixed16, fixed32, dict8, dict16, date32, date16, val]

        def decoded(i):
            a = raw[i].astype(np.int64)
            if i == 0:
                return np.where(raw[0] == np.int16(-2**15), -2**63, a)
            if i == 2:
                return np.where(raw[2] == 255, -2**31, a)
            if i == 4:
                return np.where(raw[4] == np.int32(-2**31), -2**63, a * 86400)
            if i == 5:
                return np.where(raw[5] == np.int16(-2**15), -2**63, a * 86400)
            return a

        def rng_of(i, null_val, nullable, bucket=0):
            d = decoded(i)
            live = d[d != null_val] if nullable else d
            return ExpressionRange(True, int(live.min()), int(live.max()), bool(nullable and (d == null_val).any()),
                                   bucket=bucket)
        descs_e = [
            InputColDescriptor(INT16, True, rng_of(0, -2**63, True), capi.ENC_FIXED, INT64),
            InputColDescriptor(INT32, False, rng_of(1, 0, False), capi.ENC_FIXED, INT64),
            InputColDescriptor(INT8, True, rng_of(2, -2**31, True), capi.ENC_DICT),
            InputColDescriptor(INT16, False, rng_of(3, 0, False), capi.ENC_DICT),
            InputColDescriptor(INT32, True, rng_of(4, -2**63, True, 86400), capi.ENC_DATE_IN_DAYS),
            InputColDescriptor(INT16, True, rng_of(5, -2**63, True, 86400), capi.ENC_DATE_IN_DAYS),
            InputColDescriptor(INT64, False, col_range([val], INT64, False)),
        ]
        # chunks travel as signed arrays of the storage width (the bytes are what matter)
        stored = [raw[0], raw[1], raw[2].view(np.int8), raw[3].view(np.int16), raw[4], raw[5], raw[6]]
        per_col = [split(c, fs) for c in stored]
        return descs_e, [[per_col[c][f] for c in range(len(stored))] for f in range(len(fs))]

    de, fe = enc_table()

    def era(targets, quals=(), group=(), guess=16384):
        return RelAlgExecutionUnit(list(de), list(targets), list(quals), list(group),
                                   max_groups_buffer_entry_guess=guess)

    cases.append(Case("enc_nongrouped_fixed_nulls", era([TargetExpr(COUNT, 0), TargetExpr(SUM, 0), TargetExpr(MIN, 0),
                                                         TargetExpr(MAX, 0), TargetExpr(AVG, 0), TargetExpr(SUM, 1)]),
                      fe))
    cases.append(Case("enc_filter_date_and_fixed", era([TargetExpr(COUNT), TargetExpr(MIN, 4), TargetExpr(MAX, 4),
                                                        TargetExpr(MAX, 5)],
                                                       [Qual(4, GE, 18005 * 86400), Qual(0, GT, -1000)]), fe))
    cases.append(Case("enc_group_dict8_nullable", era([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 0),
                                                       TargetExpr(MIN, 0)], group=[2]), fe))
    cases.append(Case("enc_group_dict16_unsigned", era([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 6)],
                                                       group=[3]), fe))
    cases.append(Case("enc_group_date_bucketed", era([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 0)],
                                                     group=[4]), fe))           # (key - min) / 86400
    cases.append(Case("enc_group_date16_dict8_multi", era([TargetExpr(PROJECT_KEY, 0), TargetExpr(PROJECT_KEY, 1),
                                                           TargetExpr(COUNT), TargetExpr(MAX, 1)], group=[5, 2]), fe))
    cases.append(Case("enc_group_fixed32_baseline", era([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 0)],
                                                        group=[1], guess=4000), fe))
    cases.append(Case("enc_group_fixed16_nullable_key", era([TargetExpr(PROJECT_KEY), TargetExpr(COUNT)], group=[0]), fe))

    # kENCODING_FIXED(32) BIGINT key and value, NOT NULL: the plan-time choice is the LDS perfect-hash
    # kernel reading both as 4-byte chunks
    k32 = rng.integers(0, 500, n).astype(np.int32)
    v32 = rng.integers(-10**6, 10**6, n).astype(np.int32)
    fx_descs = [InputColDescriptor(INT32, False, col_range([k32], INT32, False), capi.ENC_FIXED, INT64),
                InputColDescriptor(INT32, False, col_range([v32], INT32, False), capi.ENC_FIXED, INT64)]
    fx_frags = [[a, b] for a, b in zip(split(k32, fs), split(v32, fs))]
    cases.append(Case("enc_fixed32_key_and_value_perfect",
                      RelAlgExecutionUnit(fx_descs, [TargetExpr(PROJECT_KEY), TargetExpr(SUM, 1), TargetExpr(COUNT),
                                                     TargetExpr(MIN, 1), TargetExpr(AVG, 1)], groupby_exprs=[0]), fx_frags))

    cases.append(Case("compact_baseline_key32",
                      RelAlgExecutionUnit(descs32, [TargetExpr(PROJECT_KEY), TargetExpr(COUNT)], groupby_exprs=[0],
                                          max_groups_buffer_entry_guess=5000), frags32))  # 4-byte key (padded to 8), 8-byte count

    # ---- empty and tiny inputs
    empty = [[np.zeros(0, NP[t]) for t, _, _ in spec]]
    cases.append(Case("empty_input_nongrouped", ra([TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(AVG, 3)]), empty))
    cases.append(Case("empty_input_perfect", ra([TargetExpr(COUNT), TargetExpr(SUM, 2)], group=[1]), empty))
    cases.append(Case("no_fragments", ra([TargetExpr(COUNT), TargetExpr(MIN, 2)]), []))
    one = [[c[:1] for c in frags[0]]]
    cases.append(Case("single_row", ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 3)], group=[1]), one))

    # ---- Joins_InnerJoin_TwoTables / Joins_DifferentIntegerTypes
    m = 700
    dim_dense = rng.permutation(m).astype(np.int64)            # dense unique keys 0..m-1
    dim_w = rng.integers(-1000, 1000, m).astype(np.int64)
    dim_f = rng.random(m)
    inner_descs = [InputColDescriptor(INT64, False, col_range([dim_dense], INT64, False)),
                   InputColDescriptor(INT64, False, col_range([dim_w], INT64, False)),
                   InputColDescriptor(DOUBLE, False, col_range([dim_f], DOUBLE, False))]
    fdescs, ffrags = make_table(rng, n, fs, [
        (INT64, False, lambda r, mm: r.integers(-50, m + 50, mm)),      # some keys miss
        (INT64, False, lambda r, mm: r.integers(-10**6, 10**6, mm)),
        (INT32, True, lambda r, mm: r.integers(0, m, mm)),              # int32 nullable join key
        (INT64, False, lambda r, mm: r.integers(0, 30, mm)),            # group key
    ])

    def jra(targets, outer_col=0, quals=(), group=(), guess=16384):
        return RelAlgExecutionUnit(list(fdescs), list(targets), list(quals), list(group),
                                   inner_col_descs=list(inner_descs), join_outer_col=outer_col,
                                   max_groups_buffer_entry_guess=guess)

    dense_rng = col_range([dim_dense], INT64, False)
    for pb, tag in [(False, "perfect"), (True, "keyed")]:
        cases.append(Case(f"join_{tag}_sum_fact", jra([TargetExpr(SUM, 1)]), ffrags, [dim_dense, dim_w, dim_f],
                          dim_dense, INT64, dense_rng, pb))
        cases.append(Case(f"join_{tag}_sum_dim_count", jra([TargetExpr(SUM, 1, 1), TargetExpr(COUNT), TargetExpr(SUM, 1)]),
                          ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, pb))
        cases.append(Case(f"join_{tag}_int32_nullable_key", jra([TargetExpr(COUNT), TargetExpr(AVG, 2, 1),
                                                                 TargetExpr(MIN, 1, 1)], outer_col=2),
                          ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, pb))
        cases.append(Case(f"join_{tag}_groupby", jra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 1, 1),
                                                      TargetExpr(AVG, 2, 1)], group=[3], quals=[Qual(1, GT, 0)]),
                          ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, pb))
    # grouped joins on one-to-one tables (the gather route over large inputs): LEFT with unmatched rows (inner side NULL),
    # a nullable int32 key, MIN / MAX / COUNT over the inner columns, two group columns, several quals
    for pb, tag in [(False, "perfect"), (True, "keyed")]:
        cases.append(Case(f"join_left_{tag}_1to1_groupby", jra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(COUNT, 1, 1),
                                                               TargetExpr(SUM, 1, 1), TargetExpr(MIN, 2, 1), TargetExpr(MAX, 1, 1),
                                                               TargetExpr(SUM, 1)], group=[3]),
                          ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, pb))
        cases[-1].ra.join_kind = capi.JOIN_LEFT
        cases.append(Case(f"join_{tag}_groupby_two_keys_nullable_join_key",
                          jra([TargetExpr(PROJECT_KEY, 0), TargetExpr(PROJECT_KEY, 1), TargetExpr(COUNT), TargetExpr(AVG, 1, 1),
                               TargetExpr(MAX, 2, 1), TargetExpr(MIN, 1)], outer_col=2, group=[3, 2],
                              quals=[Qual(1, GT, -900000), Qual(0, GE, 0), Qual(3, LT, 25)]),
                          ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, pb))
    cases.append(Case("join_no_match_at_all", jra([TargetExpr(SUM, 1), TargetExpr(COUNT)], quals=[Qual(0, LT, -10)]),
                      ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, False))
    sparse = (rng.permutation(5 * m)[:m].astype(np.int64)) * 1000003
    sdescs = [InputColDescriptor(INT64, False, col_range([sparse], INT64, False))] + inner_descs[1:]
    sfd, sff = make_table(rng, n, fs, [
        (INT64, False, lambda r, mm: sparse[r.integers(0, m, mm)] + (r.random(mm) < 0.1)),
        (INT64, False, lambda r, mm: r.integers(1, 1000, mm)),
    ])
    cases.append(Case("join_sparse_keyed_sum",
                      RelAlgExecutionUnit(sfd, [TargetExpr(SUM, 1), TargetExpr(SUM, 1, 1), TargetExpr(COUNT)],
                                          inner_col_descs=sdescs, join_outer_col=0),
                      sff, [sparse, dim_w, dim_f], sparse, INT64, col_range([sparse], INT64, False), False))

    # ---- one-to-many tables, LEFT joins, composite keys (JoinHashTableTest.cpp:286-378,539-597;
    # ExecuteTest Joins_LeftOuterJoin / Joins_OneToMany shapes)
    reps = rng.integers(0, 4, m)                                  # each key 0..3 times
    dim_dup = np.repeat(np.arange(m, dtype=np.int64), reps)
    rng.shuffle(dim_dup)
    md = len(dim_dup)
    dup_w = rng.integers(-1000, 1000, md).astype(np.int64)
    dup_f = rng.random(md)
    dup_descs = [InputColDescriptor(INT64, False, col_range([dim_dup], INT64, False)),
                 InputColDescriptor(INT64, False, col_range([dup_w], INT64, False)),
                 InputColDescriptor(DOUBLE, False, col_range([dup_f], DOUBLE, False))]
    dup_rng = ExpressionRange(True, 0, m - 1)

    def dra(targets, outer_col=0, quals=(), group=(), kind=capi.JOIN_INNER, inner=dup_descs):
        return RelAlgExecutionUnit(list(fdescs), list(targets), list(quals), list(group),
                                   inner_col_descs=list(inner), join_outer_col=outer_col, join_kind=kind)

    for pb, tag in [(False, "perfect"), (True, "keyed")]:
        cases.append(Case(f"join_1n_{tag}_sums", dra([TargetExpr(SUM, 1), TargetExpr(COUNT), TargetExpr(SUM, 1, 1),
                                                      TargetExpr(AVG, 2, 1)]),
                          ffrags, [dim_dup, dup_w, dup_f], dim_dup, INT64, dup_rng, pb, join_one_to_many=1))
        cases.append(Case(f"join_1n_{tag}_groupby_int32_key", dra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT),
                                                                   TargetExpr(MAX, 1, 1), TargetExpr(AVG, 2, 1)],
                                                                  outer_col=2, group=[3]),
                          ffrags, [dim_dup, dup_w, dup_f], dim_dup, INT64, dup_rng, pb, join_one_to_many=2))
        cases.append(Case(f"join_left_{tag}_1to1", jra([TargetExpr(COUNT), TargetExpr(COUNT, 1, 1), TargetExpr(SUM, 1, 1),
                                                        TargetExpr(SUM, 1), TargetExpr(MIN, 2, 1), TargetExpr(AVG, 2, 1)]),
                          ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, pb))
        cases[-1].ra.join_kind = capi.JOIN_LEFT
        cases.append(Case(f"join_left_{tag}_1n_groupby", dra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT),
                                                              TargetExpr(COUNT, 1, 1), TargetExpr(SUM, 1, 1),
                                                              TargetExpr(MAX, 2, 1)], outer_col=2, group=[3],
                                                             kind=capi.JOIN_LEFT),
                          ffrags, [dim_dup, dup_w, dup_f], dim_dup, INT64, dup_rng, pb, join_one_to_many=1))
    cases.append(Case("join_left_no_match_at_all", dra([TargetExpr(COUNT), TargetExpr(SUM, 1, 1), TargetExpr(COUNT, 2, 1)],
                                                       quals=[Qual(0, LT, -10)], kind=capi.JOIN_LEFT),
                      ffrags, [dim_dup, dup_w, dup_f], dim_dup, INT64, dup_rng, False, join_one_to_many=1))
    # int32 inner key column, keyed: 4-byte components (getKeyComponentWidth), some inner NULLs
    dim32 = rng.permutation(3 * m)[:m].astype(np.int32)
    dim32[rng.random(m) < 0.05] = np.int32(-2**31)
    d32_descs = [InputColDescriptor(INT32, True, col_range([dim32], INT32, True))] + inner_descs[1:]
    c32 = Case("join_keyed_int32_inner_width4",
               RelAlgExecutionUnit(list(fdescs), [TargetExpr(COUNT), TargetExpr(SUM, 1, 1)],
                                   inner_col_descs=d32_descs, join_outer_col=2),
               ffrags, [dim32, dim_w, dim_f], dim32, INT32, col_range([dim32], INT32, True), True,
               join_key_nullable=True)
    cases.append(c32)
    # composite keys: (int32, int16) -> 4-byte components one-to-one; (int64, int32) with
    # duplicates -> 8-byte components one-to-many
    ca = rng.integers(0, 60, m).astype(np.int32)
    cb = (np.arange(m) // 60).astype(np.int16)                      # (ca, cb) not unique in general ...
    pair = np.unique(np.stack([ca.astype(np.int64), cb.astype(np.int64)], 1), axis=0)
    ca, cb = pair[:, 0].astype(np.int32), pair[:, 1].astype(np.int16)  # ... so keep the distinct pairs
    mc = len(ca)
    cw = rng.integers(-1000, 1000, mc).astype(np.int64)
    comp_descs = [InputColDescriptor(INT32, False, col_range([ca], INT32, False)),
                  InputColDescriptor(INT16, False, col_range([cb], INT16, False)),
                  InputColDescriptor(INT64, False, col_range([cw], INT64, False))]
    cfd, cff = make_table(rng, n, fs, [
        (INT64, False, lambda r, mm: r.integers(-3, 63, mm)),           # pairs with ca (int64 outer, int32 inner)
        (INT32, True, lambda r, mm: r.integers(0, 13, mm)),             # pairs with cb, nullable
        (INT64, False, lambda r, mm: r.integers(1, 1000, mm)),
        (INT64, False, lambda r, mm: r.integers(0, 9, mm)),             # group key
    ])
    for kind, tag in [(capi.JOIN_INNER, "inner"), (capi.JOIN_LEFT, "left")]:
        cases.append(Case(f"join_composite_key32_1to1_{tag}",
                          RelAlgExecutionUnit(cfd, [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 2, 1),
                                                    TargetExpr(SUM, 2), TargetExpr(COUNT, 2, 1)], groupby_exprs=[3],
                                              inner_col_descs=comp_descs, join_outer_col=[0, 1], join_kind=kind),
                          cff, [ca, cb, cw], [ca, cb], [INT32, INT16], ExpressionRange(), False))
    ka = (rng.integers(0, 40, 3 * m) * 10**10).astype(np.int64)
    kb = rng.integers(0, 5, 3 * m).astype(np.int32)
    kw_ = rng.integers(-1000, 1000, 3 * m).astype(np.int64)
    k_descs = [InputColDescriptor(INT64, False, col_range([ka], INT64, False)),
               InputColDescriptor(INT32, False, col_range([kb], INT32, False)),
               InputColDescriptor(INT64, False, col_range([kw_], INT64, False))]
    kfd, kff = make_table(rng, n, fs, [
        (INT64, False, lambda r, mm: r.integers(0, 45, mm) * 10**10),
        (INT8, False, lambda r, mm: r.integers(0, 6, mm)),
        (DOUBLE, False, lambda r, mm: r.random(mm)),
    ])
    cases.append(Case("join_composite_key64_1n",
                      RelAlgExecutionUnit(kfd, [TargetExpr(COUNT), TargetExpr(SUM, 2, 1), TargetExpr(SUM, 2), TargetExpr(MIN, 2, 1)],
                                          inner_col_descs=k_descs, join_outer_col=[0, 1]),
                      kff, [ka, kb, kw_], [ka, kb], [INT64, INT32], ExpressionRange(), False, join_one_to_many=1))
    # (int64, int64) one-to-one composite key, non-grouped COUNT / SUM(outer) / SUM(inner): the shape the
    # streaming probe kernel takes (k_join_sum with a second key component); sparse components, ~60 % matches,
    # inner pairs that share a first component (so the probe must compare both), ragged fragments
    qa = (rng.integers(0, 400, 4 * m) * 10**9 + 5).astype(np.int64)
    qb = rng.integers(-3, 4, 4 * m).astype(np.int64) * 2**33
    qpair = np.unique(np.stack([qa, qb], 1), axis=0)
    qa, qb = np.ascontiguousarray(qpair[:, 0]), np.ascontiguousarray(qpair[:, 1])
    qw = rng.integers(-1000, 1000, len(qa)).astype(np.int64)
    q_descs = [InputColDescriptor(INT64, False, col_range([qa], INT64, False)),
               InputColDescriptor(INT64, False, col_range([qb], INT64, False)),
               InputColDescriptor(INT64, False, col_range([qw], INT64, False))]
    qfd, qff = make_table(rng, n, fs, [
        (INT64, False, lambda r, mm: r.integers(0, 500, mm) * 10**9 + 5),
        (INT64, False, lambda r, mm: r.integers(-3, 4, mm) * 2**33),
        (INT64, False, lambda r, mm: r.integers(-50, 1000, mm)),
    ])
    cases.append(Case("join_composite_key64_1to1_sum",
                      RelAlgExecutionUnit(qfd, [TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(SUM, 2, 1)],
                                          inner_col_descs=q_descs, join_outer_col=[0, 1]),
                      qff, [qa, qb, qw], [qa, qb], [INT64, INT64], ExpressionRange(), False))
    cases.append(Case("join_composite_key64_1to1_count",
                      RelAlgExecutionUnit(qfd, [TargetExpr(COUNT)], inner_col_descs=q_descs, join_outer_col=[0, 1]),
                      qff, [qa, qb, qw], [qa, qb], [INT64, INT64], ExpressionRange(), False))
    # ---- projected expressions (SURVEY north_star "scan/filter/PROJECT"): the shapes of the
    # reference's own synthetic benchmark — GROUP BY cast(x as double) (Benchmarks/synthetic_benchmark/queries/
    # BaselineHash/BH001.sql), max(x10 + 1) / sum(x10 + 1) next to plain columns (MultiStep/MSBS001.sql, grouped by
    # cast(x1k as float)) — and casts / + - * in aggregate arguments and quals, incl. the overflow error
    def xra(exprs, targets, quals=(), group=(), guess=16384, src=(descs, frags)):
        d, fr = src
        xs = [e.with_range(expr_range(e, d, fr, exprs[:i])) for i, e in enumerate(exprs)]
        return RelAlgExecutionUnit(list(d), list(targets), list(quals), list(group), max_groups_buffer_entry_guess=guess,
                                   exprs=xs)
    NC = len(descs)
    C = Expr.col
    cases.append(Case("expr_group_cast_i32_double",     # BH001: group by cast(x as double) -> baseline layout
                      xra([C(10).cast(DOUBLE)], [TargetExpr(PROJECT_KEY), TargetExpr(COUNT, 7), TargetExpr(SUM, 7),
                                                 TargetExpr(MAX, 7), TargetExpr(MIN, 7), TargetExpr(AVG, 7)],
                          group=[NC], guess=256), frags))
    cases.append(Case("expr_group_cast_float_multistep",  # MSBS001, first step
                      xra([C(10).cast(capi.FLOAT), C(7).add(Expr.lit(INT32, 1), INT32)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(MAX, 7), TargetExpr(MAX, 5),
                           TargetExpr(MAX, NC + 1), TargetExpr(SUM, 7), TargetExpr(SUM, NC + 1)], group=[NC], guess=256),
                      frags))
    cases.append(Case("expr_arg_cast_mul_perfect",
                      xra([C(5).cast(INT64).mul(C(2), INT64)], [TargetExpr(PROJECT_KEY), TargetExpr(SUM, NC),
                                                               TargetExpr(MIN, NC), TargetExpr(COUNT)], group=[1]), frags))
    cases.append(Case("expr_qual_on_nullable_sum",
                      xra([C(7).add(C(10), INT32)], [TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(COUNT, NC)],
                          [Qual(NC, LT, 100000)]), frags))
    cases.append(Case("expr_double_arith_nongrouped",
                      xra([C(3).mul(Expr.lit(DOUBLE, 2.5), DOUBLE).sub(C(9), DOUBLE)],
                          [TargetExpr(SUM, NC), TargetExpr(MIN, NC), TargetExpr(COUNT, NC), TargetExpr(AVG, NC)],
                          [Qual(0, LT, 2**30)]), frags))
    cases.append(Case("expr_group_cast_double_to_int",   # rounding cast as a perfect-hash key
                      xra([C(3).cast(INT32)], [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(MAX, 3)],
                          group=[NC]), frags))
    cases.append(Case("expr_baseline_key_plus_literal",
                      xra([C(4).add(Expr.lit(INT64, 5), INT64)], [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 3)],
                          group=[NC], guess=8192), frags))
    cases.append(Case("expr_narrowing_cast_and_widening",
                      xra([C(1).cast(INT8), C(6).cast(INT64).mul(Expr.lit(INT64, 1000), INT64)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(SUM, NC + 1), TargetExpr(COUNT, NC + 1)], group=[NC]), frags))
    cases.append(Case("expr_overflow_is_an_error",       # int32 + 2^30 overflows for half of the rows
                      xra([C(0).add(Expr.lit(INT32, 2**30), INT32)], [TargetExpr(MAX, NC), TargetExpr(COUNT)]), frags,
                      expect_error=capi.ERR_OVERFLOW_OR_UNDERFLOW))
    cases.append(Case("expr_overflow_only_in_filtered_rows",  # ... but not among the rows that pass the qual
                      xra([C(0).add(Expr.lit(INT32, 2**30), INT32)], [TargetExpr(MAX, NC), TargetExpr(COUNT)],
                          [Qual(0, LT, 2**30)]), frags))
    cases.append(Case("expr_overflow_in_a_qual_counts_for_every_row",
                      xra([C(0).add(Expr.lit(INT32, 2**30), INT32)], [TargetExpr(COUNT)],
                          [Qual(0, LT, 2**30), Qual(NC, GT, 0)]), frags, expect_error=capi.ERR_OVERFLOW_OR_UNDERFLOW))
    # OR among the quals (LogicalIR.cpp:299-340): disjunctions of comparisons AND-ed with plain conjuncts; a NULL member is not
    # TRUE; an `x IS NOT NULL` inside a disjunction is not a constrained_not_null witness
    cases.append(Case("qual_or_two_ranges_count",
                      ra([TargetExpr(COUNT), TargetExpr(SUM, 2)], [Qual(0, LT, 2**28, 1), Qual(0, GT, 2**31 - 2**28, 1)]), frags))
    cases.append(Case("qual_or_with_null_members_and_a_conjunct_grouped",
                      ra([TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 7), TargetExpr(MIN, 8)],
                         [Qual(2, GT, -400000), Qual(7, LT, 0, 1), Qual(8, GT, 0, 1), Qual(6, capi.IS_NULL, 0, 1)],
                         group=[1]), frags))
    cases.append(Case("qual_two_disjunctions_and_not_null_member",
                      ra([TargetExpr(PROJECT_KEY), TargetExpr(SUM, 7), TargetExpr(AVG, 9), TargetExpr(COUNT, 7)],
                         [Qual(7, capi.IS_NOT_NULL, 0, 1), Qual(3, LT, 100.0, 1), Qual(10, EQ, 7, 2), Qual(10, GE, 30, 2)],
                         group=[10]), frags))
    cases.append(Case("qual_or_on_an_expression",
                      xra([C(0).mod(Expr.lit(INT32, 10), INT32)], [TargetExpr(COUNT), TargetExpr(MAX, NC)],
                          [Qual(NC, EQ, 3, 1), Qual(NC, EQ, 7, 1), Qual(1, LT, 50)]), frags))
    # / and % (ArithmeticIR.cpp:431-560, :731-760): column 5 (int32, small values around 0) as a divisor
    cases.append(Case("expr_div_by_zero_is_error_1",
                      xra([C(0).div(C(5).cast(INT32), INT32)], [TargetExpr(MAX, NC), TargetExpr(COUNT)]), frags,
                      expect_error=capi.ERR_DIV_BY_ZERO))
    cases.append(Case("expr_mod_and_div_by_a_literal_grouped",
                      xra([C(0).mod(Expr.lit(INT32, 7), INT32), C(2).div(Expr.lit(INT64, -3), INT64)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, NC + 1), TargetExpr(MIN, NC + 1)],
                          group=[NC], guess=64), frags))
    cases.append(Case("expr_double_division_by_a_positive_column",
                      xra([C(3).div(C(9).add(Expr.lit(DOUBLE, 1e9), DOUBLE), DOUBLE)],
                          [TargetExpr(SUM, NC), TargetExpr(COUNT, NC)]), frags))
    # comparisons of two values and CASE (CompareIR.cpp:230-330, CaseIR.cpp:67-140): a column-vs-column filter is a qual on the
    # BOOLEAN expression; the guard of a division keeps error 1 away; CASE as a group key and as an argument
    cases.append(Case("expr_filter_column_less_than_column",          # WHERE c7 < c2 (NULL c7: not TRUE)
                      xra([C(7).cast(INT64).cmp(capi.EX_LT, C(2))], [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 7)],
                          [Qual(NC, EQ, 1)], group=[1]), frags))
    cases.append(Case("expr_filter_not_equal_columns_nongrouped",     # WHERE c6 <> c5 (INT16 vs INT8 through casts)
                      xra([C(6).cast(INT32).cmp(capi.EX_NE, C(5).cast(INT32))], [TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(COUNT, 6)],
                          [Qual(NC, EQ, 1), Qual(1, LT, 60)]), frags))
    cases.append(Case("expr_case_guards_a_division",                  # SUM(CASE WHEN c5 <> 0 THEN c2 / c5 ELSE 0 END): never error 1
                      xra([Expr.case(C(5).cast(INT64).cmp(capi.EX_NE, Expr.lit(INT64, 0)), C(2).div(C(5).cast(INT64), INT64),
                                     Expr.lit(INT64, 0), INT64)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(SUM, NC), TargetExpr(MIN, NC), TargetExpr(COUNT, NC)], group=[1]), frags))
    cases.append(Case("expr_case_as_group_key_and_null_else",         # GROUP BY CASE WHEN c10 < 20 THEN 0 ELSE 1 END; MAX(CASE WHEN c7 > 0 THEN c7 END)
                      xra([Expr.case(C(10).cmp(capi.EX_LT, Expr.lit(INT32, 20)), Expr.lit(INT32, 0), Expr.lit(INT32, 1), INT32),
                           Expr.case(C(7).cmp(capi.EX_GT, Expr.lit(INT32, 0)), C(7), Expr.null(INT32), INT32)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(MAX, NC + 1), TargetExpr(COUNT, NC + 1),
                           TargetExpr(SUM, NC + 1)], group=[NC]), frags))
    cases.append(Case("expr_case_two_whens_double",                   # CASE WHEN c9 < -100 THEN -1.0 WHEN c9 < 100 THEN c3 ELSE c9 END
                      xra([Expr.case(C(9).cmp(capi.EX_LT, Expr.lit(DOUBLE, -100.0)), Expr.lit(DOUBLE, -1.0),
                                     Expr.case(C(9).cmp(capi.EX_LT, Expr.lit(DOUBLE, 100.0)), C(3), C(9), DOUBLE), DOUBLE)],
                          [TargetExpr(SUM, NC), TargetExpr(COUNT, NC), TargetExpr(MIN, NC)], [Qual(1, GE, 10)]), frags))
    cases.append(Case("expr_unguarded_division_in_the_taken_branch_is_error_1",
                      xra([Expr.case(C(5).cast(INT64).cmp(capi.EX_LT, Expr.lit(INT64, 50)), C(2).div(C(5).cast(INT64), INT64),
                                     Expr.lit(INT64, 0), INT64)],
                          [TargetExpr(SUM, NC), TargetExpr(COUNT)]), frags, expect_error=capi.ERR_DIV_BY_ZERO))
    # NOT / AND / OR / IS NULL / unary minus as values (LogicalIR.cpp:197-432, ArithmeticIR.cpp:787-838): a filter that has no
    # qual shape is one BOOLEAN expression `= 1`; the short-circuit form guards a division; IS NULL feeds a CASE; -x in aggregates
    lt60 = C(1).cmp(capi.EX_LT, Expr.lit(INT64, 60))
    c7pos = C(7).cmp(capi.EX_GT, Expr.lit(INT32, 0))
    cases.append(Case("expr_filter_and_inside_or",                    # WHERE (c1 < 60 AND c7 > 0) OR c6 IS NULL
                      xra([lt60.logical(capi.EX_AND, c7pos).logical(capi.EX_OR, C(6).is_null())],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(COUNT, 7)],
                          [Qual(NC, EQ, 1)], group=[1]), frags))
    cases.append(Case("expr_filter_not_over_a_disjunction",           # WHERE NOT (c1 < 60 OR c7 > 0): a NULL c7 with c1 >= 60 is not TRUE
                      xra([lt60.logical(capi.EX_OR, c7pos).logical_not()],
                          [TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(COUNT, 7), TargetExpr(MIN, 1)], [Qual(NC, EQ, 1)]), frags))
    cases.append(Case("expr_short_circuit_and_guards_a_division",     # WHERE c5 <> 0 AND c2 / c5 > 100: never error 1
                      xra([C(5).cast(INT64).cmp(capi.EX_NE, Expr.lit(INT64, 0)).logical(
                              capi.EX_AND, C(2).div(C(5).cast(INT64), INT64).cmp(capi.EX_GT, Expr.lit(INT64, 100)), True)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 2)], [Qual(NC, EQ, 1)], group=[1]), frags))
    cases.append(Case("expr_plain_and_evaluates_the_division_is_error_1",
                      xra([C(5).cast(INT64).cmp(capi.EX_NE, Expr.lit(INT64, 0)).logical(
                              capi.EX_AND, C(2).div(C(5).cast(INT64), INT64).cmp(capi.EX_GT, Expr.lit(INT64, 100)))],
                          [TargetExpr(COUNT)], [Qual(NC, EQ, 1)]), frags, expect_error=capi.ERR_DIV_BY_ZERO))
    cases.append(Case("expr_is_null_in_case_and_uminus_arguments",    # SUM(CASE WHEN c8 IS NULL THEN 1 ELSE 0 END), SUM(-c2), MIN(-c7), MAX(-c9)
                      xra([Expr.case(C(8).is_null(), Expr.lit(INT32, 1), Expr.lit(INT32, 0), INT32), C(2).neg(INT64), C(7).neg(INT32),
                           C(9).neg(DOUBLE)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(SUM, NC), TargetExpr(SUM, NC + 1), TargetExpr(MIN, NC + 2),
                           TargetExpr(COUNT, NC + 2), TargetExpr(MAX, NC + 3)], group=[10]), frags))
    cases.append(Case("expr_group_by_a_boolean",                      # GROUP BY (c7 > 0 OR c6 IS NOT NULL): keys 1 / 0 / NULL
                      xra([c7pos.logical(capi.EX_OR, C(6).is_null().logical_not())],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(AVG, 3)], group=[NC], guess=64), frags))
    # a program longer than 12 nodes as several expressions: the reference's own
    # `WHERE x > 6 AND x < 8 OR (z > 100 AND z < 103) [OR (t > ..)]` (Tests/ExecuteTest.cpp:1906-1913) has 15 / 23 nodes
    band = lambda c, t, lo, hi: C(c).cmp(capi.EX_GT, Expr.lit(t, lo)).logical(capi.EX_AND, C(c).cmp(capi.EX_LT, Expr.lit(t, hi)))
    cases.append(Case("expr_three_bands_composed_of_earlier_expressions",
                      xra([band(1, INT64, 10, 20), band(6, INT16, 100, 900), band(7, INT32, -5000, 5000),
                           Expr.col(NC).logical(capi.EX_OR, Expr.col(NC + 1)).logical(capi.EX_OR, Expr.col(NC + 2))],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT), TargetExpr(SUM, 2), TargetExpr(COUNT, 6)],
                          [Qual(NC + 3, EQ, 1)], group=[1]), frags))
    cases.append(Case("expr_value_of_an_earlier_expression_in_arithmetic",   # SUM((c2 + c8) * 2 - (c2 + c8) / 3), MAX(c2 + c8)
                      xra([C(2).add(C(8), INT64),
                           Expr.col(NC).mul(Expr.lit(INT64, 2), INT64).sub(Expr.col(NC).div(Expr.lit(INT64, 3), INT64), INT64)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(SUM, NC + 1), TargetExpr(MAX, NC), TargetExpr(COUNT, NC + 1)],
                          group=[10]), frags))
    cases.append(Case("expr_error_in_an_earlier_expression_read_by_a_qual",   # (c5 / (c5 - c5)) read by the filter's expression: error 1 on every row
                      xra([C(5).div(C(5).sub(C(5), INT8), INT8), Expr.col(NC).cmp(capi.EX_GT, Expr.lit(INT8, 0))],
                          [TargetExpr(COUNT)], [Qual(NC + 1, EQ, 1), Qual(1, LT, 0)]), frags, expect_error=capi.ERR_DIV_BY_ZERO))
    cases.append(Case("expr_error_in_an_earlier_expression_of_filtered_rows_only",   # the same division behind a target: no row passes, no error
                      xra([C(5).div(C(5).sub(C(5), INT8), INT8), Expr.col(NC).cast(INT64).add(C(2), INT64)],
                          [TargetExpr(COUNT), TargetExpr(SUM, NC + 1)], [Qual(1, LT, 0)]), frags))
    # COUNT_IF / SUM_IF over a condition that has no qual shape: the condition is a BOOLEAN expression, `that column = 1`
    cases.append(Case("expr_conditional_aggregates_over_boolean_expressions",   # COUNT_IF(c1 < 60 AND c7 > 0), SUM_IF(c2, c6 IS NULL OR c7 > 0)
                      xra([lt60.logical(capi.EX_AND, c7pos), C(6).is_null().logical(capi.EX_OR, c7pos)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(COUNT_IF, cond=Qual(NC, EQ, 1)), TargetExpr(SUM_IF, 2, cond=Qual(NC + 1, EQ, 1)),
                           TargetExpr(COUNT)], group=[10]), frags))
    # the derived-plan routes (cast key, column +- literal) take expressions OUT of the plan and renumber the rest: not where
    # an expression reads another one's value
    cases.append(Case("expr_cast_key_and_shifted_argument_read_by_a_later_expression",
                      xra([C(10).cast(DOUBLE), C(7).add(Expr.lit(INT32, 1), INT32), Expr.col(NC + 1).cast(INT64).mul(Expr.lit(INT64, 2), INT64)],
                          [TargetExpr(PROJECT_KEY), TargetExpr(SUM, NC + 2), TargetExpr(MAX, NC + 1), TargetExpr(COUNT, NC + 2)],
                          group=[NC], guess=256), frags))
    um = np.array([5, -2**31, 7], dtype=np.int32)
    um_src = ([InputColDescriptor(INT32, False, col_range([um], INT32, False))], [[um]])
    cases.append(Case("expr_uminus_of_the_type_minimum_is_error_7",   # -c0 where a NOT NULL INT column holds INT32_MIN
                      xra([C(0).neg(INT32)], [TargetExpr(SUM, 1)], src=um_src), um_src[1],
                      expect_error=capi.ERR_OVERFLOW_OR_UNDERFLOW))
    jx = [Expr.col(1).add(Expr.lit(INT64, 2**63 - 10**6), INT64)]   # overflows for positive values of column 1
    jxr = [e.with_range(expr_range(e, fdescs, ffrags)) for e in jx]
    cases.append(Case("expr_join_overflow_only_in_filtered_rows",  # rows with col1 > 0 are dropped by the qual
                      RelAlgExecutionUnit(list(fdescs), [TargetExpr(MAX, len(fdescs)), TargetExpr(COUNT), TargetExpr(SUM, 1, 1)],
                                          [Qual(1, LE, 0)], [], inner_col_descs=list(inner_descs), join_outer_col=0, exprs=jxr),
                      ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, False))
    # key + (INT64_MAX - (m - 1)) overflows exactly for the keys >= m, which find no inner row: an INNER join drops
    # those rows before the target is evaluated (no error); under a LEFT join they survive and the step fails
    jx3 = [Expr.col(0).add(Expr.lit(INT64, 2**63 - 1 - (m - 1)), INT64)]
    jx3r = [e.with_range(expr_range(e, fdescs, ffrags)) for e in jx3]
    for kind, tag, err in [(capi.JOIN_INNER, "inner", None), (capi.JOIN_LEFT, "left", capi.ERR_OVERFLOW_OR_UNDERFLOW)]:
        cases.append(Case(f"expr_join_overflow_only_in_unmatched_rows_{tag}",
                          RelAlgExecutionUnit(list(fdescs), [TargetExpr(MAX, len(fdescs)), TargetExpr(COUNT), TargetExpr(SUM, 1, 1)],
                                              [], [], inner_col_descs=list(inner_descs), join_outer_col=0, join_kind=kind,
                                              exprs=jx3r),
                          ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, False, expect_error=err))
    jx2 = [Expr.col(1).mul(Expr.lit(INT64, 3), INT64).sub(Expr.col(3), INT64)]
    jx2r = [e.with_range(expr_range(e, fdescs, ffrags)) for e in jx2]
    cases.append(Case("expr_join_groupby_expression_target",
                      RelAlgExecutionUnit(list(fdescs), [TargetExpr(PROJECT_KEY), TargetExpr(SUM, len(fdescs)), TargetExpr(SUM, 1, 1),
                                                         TargetExpr(COUNT)], [], [3], inner_col_descs=list(inner_descs),
                                          join_outer_col=0, exprs=jx2r),
                      ffrags, [dim_dense, dim_w, dim_f], dim_dense, INT64, dense_rng, False))

    # ---- edge cases of the wider shapes: empty inputs, empty / all-NULL inner tables
    cases.append(Case("multi_col_empty_input", ra([K0, K1, TargetExpr(COUNT), TargetExpr(SUM, 2)], group=[4, 1],
                                                  guess=4096), empty))
    cases.append(Case("multi_col_no_fragments", ra([TargetExpr(COUNT), TargetExpr(AVG, 3)], group=[1, 5]), []))
    e_key = np.zeros(0, np.int64)
    e_descs = [InputColDescriptor(INT64, False, ExpressionRange(True, 0, -1)),
               InputColDescriptor(INT64, False, ExpressionRange(True, 0, -1)),
               InputColDescriptor(DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 0.0))]
    for kind, tag in [(capi.JOIN_INNER, "inner"), (capi.JOIN_LEFT, "left")]:
        cases.append(Case(f"join_empty_inner_table_{tag}",
                          dra([TargetExpr(COUNT), TargetExpr(SUM, 1), TargetExpr(SUM, 1, 1), TargetExpr(COUNT, 2, 1)],
                              kind=kind, inner=e_descs),
                          ffrags, [e_key, e_key, np.zeros(0, np.float64)], e_key, INT64, ExpressionRange(True, 0, -1),
                          False, join_one_to_many=1))
    all_null = np.full(50, -2**63, dtype=np.int64)
    n_descs = [InputColDescriptor(INT64, True, ExpressionRange(True, 0, -1, True))] + dup_descs[1:]
    cases.append(Case("join_all_null_inner_keys_left",
                      dra([TargetExpr(COUNT), TargetExpr(COUNT, 1, 1), TargetExpr(MIN, 2, 1)], kind=capi.JOIN_LEFT,
                          inner=n_descs),
                      ffrags, [all_null, dup_w[:50], dup_f[:50]], all_null, INT64, ExpressionRange(True, 0, -1, True),
                      False, join_one_to_many=2, join_key_nullable=True))
    cases.append(Case("compact_join_1n_count", dra([TargetExpr(COUNT)], outer_col=2, group=[3], kind=capi.JOIN_LEFT),
                      ffrags, [dim_dup, dup_w, dup_f], dim_dup, INT64, dup_rng, False, join_one_to_many=1))
    return cases
