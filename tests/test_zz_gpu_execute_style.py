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
