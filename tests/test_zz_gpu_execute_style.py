"""GPU leg of tests/test_execute_style.py: the reference's `test` table and query texts
(Tests/ExecuteTest.cpp) through the HIP library, compared with SQLite running the SQL text — the
reference's own `c(query, dt)` with dt = GPU."""
import math

import pytest

from tests.cases import Case
from tests.helpers import F32_ATOL, F32_RTOL
from tests.test_execute_style import QUERIES, REPEAT, _key, _rows, _table, _unit
from tests.test_gpu_parity import _fetch_result, _upload, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("qi", range(len(QUERIES)), ids=[s[0][7:60].replace(" ", "_") for s in QUERIES])
def test_reference_queries_on_gpu(torch_cuda, oracle, qi):
    from heavydb_amd.executor import Executor
    sql, targets, quals, group = QUERIES[qi]
    descs, frags, db = _table()
    ra, frags = _unit(descs, frags, targets, quals, group, num_tuples=sum(REPEAT))
    case = Case("ref", ra, frags)
    frag_t, inner_t = _upload(torch_cuda, case)
    rs = Executor(0).executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    qm = rs.getQueryMemDesc()
    fp = [bool(qm.target_is_fp[t]) for t in range(qm.n_targets)]
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp))
                   for r in db.execute(sql).fetchall()), key=_key)
    got = sorted(_rows(rs.fetch(), qm), key=_key)
    assert len(got) == len(want), (sql, want, got)
    for w, g in zip(want, got):
        for t, (a, b) in enumerate(zip(w, g)):
            if a is None or b is None:
                assert a is None and b is None, (sql, t, w, g)
            elif fp[t]:
                rt, at = (F32_RTOL, F32_ATOL) if qm.target_arg_is_f32[t] else (1e-9, 0.0)
                assert math.isclose(a, b, rel_tol=rt, abs_tol=at), (sql, t, w, g)
            else:
                assert a == b, (sql, t, w, g)


def test_reference_expression_queries_on_gpu(torch_cuda, oracle):
    """Select.FilterAndSimpleAggregation's queries with expressions (tests/test_execute_style.py EXPR_QUERIES: arithmetic over
    several columns, narrowing casts, unary minus, AND inside OR split over expressions, the deferred qual behind a short-circuit
    AND, IS NULL of an expression) through k_project and the kernel families, against SQLite running the reference's text."""
    from heavydb_amd.executor import Executor
    from tests.test_execute_style import EXPR_QUERIES, _check_rows, _unit_x
    assert len(EXPR_QUERIES) == 76
    ex = Executor(0)
    for sql, targets, quals, group, exprs in EXPR_QUERIES:
        descs, frags, db = _table()
        ra, frags = _unit_x(descs, frags, targets, quals, group, exprs, num_tuples=sum(REPEAT))
        case = Case("ref", ra, frags)
        frag_t, inner_t = _upload(torch_cuda, case)
        rs = ex.executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
        qm = rs.getQueryMemDesc()
        fp = [bool(qm.target_is_fp[t]) for t in range(qm.n_targets)]
        want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp)) for r in db.execute(sql).fetchall()), key=_key)
        _check_rows(sql, want, sorted(_rows(rs.fetch(), qm), key=_key), fp, qm, "hip", 1e-9)


def test_reference_div_by_zero_queries_on_gpu(torch_cuda, oracle):
    """Select.DivByZero (tests/test_execute_style.py DIVZERO_QUERIES): error 1 where the reference EXPECT_THROWs, and its own
    literal — every row — for `WHERE x = x OR y / (x - x) = y` (the short-circuit OR)."""
    from heavydb_amd import capi
    from heavydb_amd.executor import Executor
    from tests.test_execute_style import DIVZERO_QUERIES, _unit_x
    ex = Executor(0)
    for sql, targets, quals, group, exprs, expect in DIVZERO_QUERIES:
        descs, frags, db = _table()
        ra, frags = _unit_x(descs, frags, targets, quals, group, exprs, num_tuples=sum(REPEAT))
        case = Case("ref", ra, frags)
        frag_t, inner_t = _upload(torch_cuda, case)
        if expect == capi.ERR_DIV_BY_ZERO:
            with pytest.raises(capi.Mi355qError) as err:
                ex.executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
            assert err.value.code == capi.ERR_DIV_BY_ZERO, (sql, err.value.code)
        else:
            rs = ex.executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
            assert _rows(rs.fetch(), rs.getQueryMemDesc()) == [(expect,)], sql


def test_reference_boolean_column_queries_on_gpu(torch_cuda, oracle):
    """Select.BooleanColumn (tests/test_execute_style.py BOOLEAN_QUERIES): the reference's own ASSERT_EQ literals."""
    from heavydb_amd.executor import Executor
    from tests.test_execute_style import BOOLEAN_QUERIES, _unit_x
    ex = Executor(0)
    for sql, targets, quals, group, exprs, expect in BOOLEAN_QUERIES:
        descs, frags, db = _table()
        ra, frags = _unit_x(descs, frags, targets, quals, group, exprs, num_tuples=sum(REPEAT))
        case = Case("ref", ra, frags)
        frag_t, inner_t = _upload(torch_cuda, case)
        rs = ex.executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
        want = sorted(expect if expect is not None else [tuple(r) for r in db.execute(sql).fetchall()])
        assert sorted(_rows(rs.fetch(), rs.getQueryMemDesc())) == want, (sql, want)


def test_reference_overflow_queries_on_gpu(torch_cuda, oracle):
    """Select.OverflowAndUnderFlow's filters (tests/test_execute_style.py OVERFLOW_QUERIES): SQLite's count where the reference
    runs c(..), error 7 where it EXPECT_THROWs (unary minus of a NOT NULL INT64_MIN among them)."""
    from heavydb_amd import capi
    from heavydb_amd.executor import Executor
    from tests.test_execute_style import OVERFLOW_QUERIES, _unit_x, agg, q
    ex = Executor(0)
    for sql, exprs, op, lit, expect in OVERFLOW_QUERIES:
        descs, frags, db = _table()
        ra, frags = _unit_x(descs, frags, [agg("COUNT")], [q(("x", 0), op, lit)], [], exprs, num_tuples=sum(REPEAT))
        case = Case("ref", ra, frags)
        frag_t, inner_t = _upload(torch_cuda, case)
        if expect:
            with pytest.raises(capi.Mi355qError) as err:
                ex.executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
            assert err.value.code == expect, (sql, err.value.code)
        else:
            rs = ex.executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
            assert _rows(rs.fetch(), rs.getQueryMemDesc()) == [tuple(r) for r in db.execute(sql).fetchall()], sql


@pytest.mark.parametrize("ji", range(11))
def test_reference_join_queries_on_gpu(torch_cuda, oracle, ji):
    from heavydb_amd.executor import Executor
    from tests.test_execute_style import JOIN_QUERIES, _compare, _join_case
    from tests.test_gpu_parity import _build_join
    assert len(JOIN_QUERIES) == 11
    descs, frags, db = _table()
    case, sql = _join_case(descs, frags, db, JOIN_QUERIES[ji])
    frag_t, inner_t = _upload(torch_cuda, case)
    hj, keep = _build_join(torch_cuda, case)
    case.ra.join_table = hj
    rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    qm = rs.getQueryMemDesc()
    _compare(sql, db, qm, [("HIP library", _rows(rs.fetch(), qm))])


def test_group_by_test_baseline_no_filters_on_gpu(torch_cuda):
    """Tests/GroupByTest.cpp BaselineNoFilters (:264-338): a dictionary-encoded string key whose cached range is
    [0, 134217728], no filter, max_groups_buffer_entry_guess = 1 — the perfect hash is kept (134 217 729 entries,
    2 GB) and the two groups come back with COUNT = 1 each."""
    import numpy as np
    from heavydb_amd import capi
    from heavydb_amd.executor import Executor
    from tests.test_groupby_test_style import STR, TOO_BIG, X, _unit
    ra = _unit(TOO_BIG, False, 1)
    case = Case("BaselineNoFilters", ra, [[STR, X]])
    frag_t, inner_t = _upload(torch_cuda, case)
    rs = Executor(0).executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    qm = rs.getQueryMemDesc()
    assert qm.desc_type == capi.GROUP_BY_PERFECT_HASH and qm.entry_count == TOO_BIG + 1
    assert rs.rowCount() == 2
    iv, dv, nu = rs.fetch()
    assert sorted(int(v) for v in np.asarray(iv)[:, 0]) == [1, 1]


def _gpu_rows(torch_cuda, ra, frags):
    from heavydb_amd.executor import Executor
    case = Case("ref", ra, frags)
    frag_t, inner_t = _upload(torch_cuda, case)
    rs = Executor(0).executeWorkUnit(ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    qm = rs.getQueryMemDesc()
    return qm, _rows(rs.fetch(), qm)


def test_select_count_if_and_sum_if_on_gpu(torch_cuda):
    """Select.CountIf / Select.SumIf (ExecuteTest.cpp:4020-4198) with dt = GPU: COUNT_IF(cond) against COUNT(1) WHERE cond
    and SUM_IF(v, cond) against SUM(CASE WHEN cond THEN v END) (SQLite) on the reference's data_types_basic5 fixture,
    conditions `x IS NULL`, `x IS NOT NULL`, `x > 0`; then the grouped SUM_IF queries on the `test` table."""
    import numpy as np
    from heavydb_amd import capi
    from heavydb_amd.executor import Qual, RelAlgExecutionUnit, TargetExpr
    from tests.test_execute_style import (_B5_CONDS, NAMES, _compare, _expected_sum, basic5_unit, count_if_queries,
                                          sum_if_queries)
    (names, arrays, descs, db), queries = count_if_queries()
    for sql, alt, (cols, targets, quals, group) in queries:
        want = db.execute(alt).fetchone()[0]
        ra, frags = basic5_unit(arrays, descs, cols, targets, quals, group)
        qm, rows = _gpu_rows(torch_cuda, ra, frags)
        if group:
            assert [r[-1] for r in rows] == ([want] if want else []), (sql, rows, want)
        else:
            assert rows == [(want,)], (sql, rows, want)
    (names, arrays, descs, db), queries = sum_if_queries()
    for sql, alt, (cols, targets, quals, group), value_col in queries:
        want = _expected_sum(db, arrays, alt, value_col)
        ra, frags = basic5_unit(arrays, descs, cols, targets, quals, group)
        qm, rows = _gpu_rows(torch_cuda, ra, frags)
        got = rows[0][0]
        if want is None or got is None:
            assert want is None and got is None, (sql, got, want)
        elif isinstance(got, float):
            rt, at = (F32_RTOL, F32_ATOL) if value_col == "Float_" else (1e-12, 0.0)
            assert math.isclose(got, want, rel_tol=rt, abs_tol=at), (sql, got, want)
        else:
            assert got == want, (sql, got, want)
    descs_t, frags_t, db_t = _table()
    for col in ["fn", "dn", "u", "ofd", "smallint_nulls"]:
        for op, (opc, lit) in _B5_CONDS.items():
            alt = f"SELECT {col}, SUM(CASE WHEN {col}{op} THEN {col} END) FROM test GROUP BY 1"
            i = NAMES.index(col)
            ra = RelAlgExecutionUnit([descs_t[i]], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.SUM_IF, 0, cond=Qual(0, opc, lit))],
                                     [], [0], num_tuples=sum(REPEAT))
            qm, rows = _gpu_rows(torch_cuda, ra, [[f[i]] for f in frags_t])
            _compare(alt, db_t, qm, [("HIP library", rows)])


@pytest.mark.parametrize("n_frags", [0, 1])
def test_select_aggregate_on_empty_table_and_null_group_by_on_gpu(torch_cuda, n_frags):
    """Select.AggregateOnEmptyTable (ExecuteTest.cpp:2298-2331: no fragments / one empty fragment, with and without a
    qual) and Select.NullGroupBy (:1870-1883) with dt = GPU."""
    from tests.test_execute_style import EMPTY_TABLE_QUERIES, empty_table_unit, null_group_by_unit
    for sql, fn, cols, with_where in EMPTY_TABLE_QUERIES:
        ra, frags = empty_table_unit(fn, cols, with_where, n_frags)
        qm, rows = _gpu_rows(torch_cuda, ra, frags)
        assert rows == [tuple([0 if fn == "COUNT" else None] * len(cols))], (sql, rows)
    for t in ("TEXT", "DOUBLE"):
        ra, frags = null_group_by_unit(t)
        qm, rows = _gpu_rows(torch_cuda, ra, frags)
        assert rows == [(None,)], (t, rows)
