"""Full ORDER BY on the device (mi355q_result_sort: several order entries, NULLS FIRST / LAST per entry,
any LIMIT / OFFSET) against the oracle's restatement of ResultSet::sort with ResultSetComparator
(ResultSet.cpp:781-851, :1310-1470).  Rows that tie on every order entry may come out in any order (in
the reference too), so what is compared position by position is the tuple of ORDER values; every
returned row must be a distinct row of the table."""
from __future__ import annotations

import numpy as np
import pytest

from heavydb_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


def _order_values(q, rows, order):
    """[n, len(order)] float64 matrix + null mask of the order entries' values of whole rows."""
    kq = q.key_bytes // 8
    vals = np.zeros((rows.shape[0], len(order)), dtype=np.float64)
    nulls = np.zeros((rows.shape[0], len(order)), dtype=bool)
    for j, (t, _, _) in enumerate(order):
        s = q.target_slot[t]
        if q.target_agg[t] == capi.PROJECT_KEY and s < 0:
            v = rows[:, q.target_key_idx[t]]
            nulls[:, j] = v == q.target_null[t]
            vals[:, j] = v.astype(np.float64)
        elif q.target_agg[t] == capi.AVG:
            cnt = rows[:, kq + s + 1]
            nulls[:, j] = cnt == 0
            sm = rows[:, kq + s].view(np.float64) if q.target_arg_is_fp[t] else rows[:, kq + s].astype(np.float64)
            vals[:, j] = np.where(cnt == 0, 0.0, sm / np.maximum(cnt, 1))
        elif q.target_is_fp[t]:
            v = rows[:, kq + s]
            nulls[:, j] = (v == q.target_null[t]) & bool(q.target_skip_null[t])
            vals[:, j] = v.view(np.float64)
        else:
            v = rows[:, kq + s]
            nulls[:, j] = (v == q.target_null[t]) & bool(q.target_skip_null[t] or q.target_agg[t] == capi.PROJECT_KEY)
            vals[:, j] = v.astype(np.float64)
    vals[nulls] = 0.0
    return vals, nulls


ORDERS = {
    "count_desc_key_asc": [(1, True, False), (0, False, False)],
    "max_nullable_nulls_first_then_avg_desc": [(4, False, True), (2, True, False)],
    "max_nullable_desc_nulls_last_then_count_then_key": [(4, True, False), (1, False, False), (0, True, False)],
    "min_double_asc": [(3, False, False)],
    "avg_asc_key_desc": [(2, False, False), (0, True, False)],
}


@pytest.mark.parametrize("layout", ["baseline", "perfect"])
@pytest.mark.parametrize("name", list(ORDERS))
def test_sort_on_device_matches_oracle(torch_cuda, oracle, name, layout):
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, InputColDescriptor, RelAlgExecutionUnit,
                                      TargetExpr)
    torch = torch_cuda
    rng = np.random.default_rng(31)
    n, n_keys = 400_000, 30_000
    stride = 1000003 if layout == "baseline" else 1
    key = (rng.integers(0, n_keys, n) * stride + 7).astype(np.int64)
    ival = rng.integers(-50, 50, n).astype(np.int64)                 # few distinct values: many ties
    dval = np.round(rng.random(n) * 20.0).astype(np.float64)
    nval = rng.integers(-10**3, 10**3, n).astype(np.int64)
    nval[(key % 7 == 0) | (rng.random(n) < 0.2)] = -2**63            # groups without any value -> NULL
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * stride + 7)),
             InputColDescriptor(capi.INT64, False, ExpressionRange(True, -50, 49)),
             InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 20.0)),
             InputColDescriptor(capi.INT64, True, ExpressionRange(True, -10**3, 10**3, True))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1),
                                     TargetExpr(capi.MIN, 2), TargetExpr(capi.MAX, 3)], [], [0],
                             max_groups_buffer_entry_guess=2 * n_keys)
    cols = [key, ival, dval, nval]
    dev = [torch.from_numpy(c).cuda() for c in cols]
    fr = FetchResult([[int(t.data_ptr()) for t in dev]], [n], keepalive=dev)
    rs = Executor(0).executeWorkUnit(ra, fr, allow_retry=False)
    q = rs.getQueryMemDesc()
    assert q.desc_type == (capi.GROUP_BY_BASELINE_HASH if layout == "baseline" else capi.GROUP_BY_PERFECT_HASH)
    rq = q.row_size // 8
    qo, want, code = oracle.execute(ra.to_plan(), [cols], n_threads=1)
    assert code == 0
    want = want.reshape(-1, rq)
    order = ORDERS[name]
    live = rs.rowCount()
    table = rs.getStorage().reshape(-1, rq)
    for limit, offset in ((0, 0), (10, 0), (5000, 0), (4097, 123), (0, live - 3), (50, live + 5)):
        cap = limit if limit else live
        out = torch.zeros((max(cap, 1), rq), dtype=torch.int64, device="cuda")
        got_n = rs.sort_by(order, int(out.data_ptr()), limit=limit, offset=offset)
        perm = oracle.sort(qo, want, order, limit=limit, offset=offset)
        assert got_n == len(perm), (limit, offset, got_n, len(perm))
        if got_n == 0:
            continue
        got = out.cpu().numpy()[:got_n]
        gv, gn = _order_values(q, got, order)
        wv, wn = _order_values(qo, want[perm], order)
        assert np.array_equal(gn, wn), (name, limit, offset)
        assert np.array_equal(gv, wv), (name, limit, offset, np.nonzero((gv != wv).any(axis=1))[0][:5])
        # every returned row is a distinct row of the device table
        kq = max(q.key_bytes // 8, 1)
        if not q.keyless:
            keys = got[:, 0]
            assert len(np.unique(keys)) == got_n
            lookup = {int(r[0]): r for r in table[table[:, 0] != 2**63 - 1]} if got_n <= 5000 else None
            if lookup is not None:
                for r in got:
                    assert np.array_equal(lookup[int(r[0])], r)


@pytest.mark.parametrize("qi", range(8))
def test_select_gpu_sort_and_speculative_top_n_on_gpu(torch_cuda, oracle, qi):
    """Select.GpuSort / Select.SpeculativeTopNSort (ExecuteTest.cpp:11338-11378) with dt = GPU: the reference's
    gpu_sort_test table and query texts, GROUP BY + ORDER BY val [LIMIT n] through mi355q_execute +
    mi355q_result_sort, against SQLite running the text."""
    from heavydb_amd.executor import Executor, FetchResult
    from tests.test_oracle_sort import GPU_SORT_QUERIES, gpu_sort_db, gpu_sort_unit
    torch = torch_cuda
    assert len(GPU_SORT_QUERIES) == 8
    sql, col, n_counts, order_t, desc, limit, project = GPU_SORT_QUERIES[qi]
    ra, frags = gpu_sort_unit(col, n_counts)
    dev = [[torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in f] for f in frags]
    fr = FetchResult([[int(t.data_ptr()) for t in f] for f in dev], [len(f[0]) for f in frags], keepalive=dev)
    rs = Executor(0).executeWorkUnit(ra, fr, allow_retry=False)
    q = rs.getQueryMemDesc()
    rq = q.row_size // 8
    n_live = rs.rowCount()
    out = torch.empty((max(n_live, 1), rq), dtype=torch.int64, device="cuda")
    n = rs.sort_by([(order_t, desc, False)], int(out.data_ptr()), limit=limit)
    assert n == (min(limit, n_live) if limit else n_live)
    top = oracle.init_buffer(q).reshape(q.entry_count, rq)
    top[:n] = out[:n].cpu().numpy()
    iv, dv, nu = oracle.fetch_rows(q, top.reshape(-1))      # iteration of the sorted rows, in order
    got = [tuple(int(v) for v in iv[i]) for i in range(n)]
    if project is not None:
        got = [tuple(r[c] for c in project) for r in got]
    want = [tuple(r) for r in gpu_sort_db().execute(sql).fetchall()]
    assert got == want, (sql, got, want)
