"""The PROJECTION family on the CPU: the real kernels_proj.hip (and api_projection.cpp / plan.cpp) compiled for the host
(tests/hostsim) against the oracle's restatement of the reference's projection runtime.  The same cases run on the device
in tests/test_zz_gpu_projection.py."""
import ctypes as C

import numpy as np
import pytest

from heavydb_amd import capi
from tests import proj_cases
from tests.helpers import compare_buffers, compare_rows, hostsim_lib, qmd_equal

CASES = proj_cases.build_cases()


@pytest.fixture(scope="module")
def sim():
    lib = capi.load_library(hostsim_lib())
    saved = capi._lib
    capi._lib = lib
    yield lib
    capi._lib = saved


def aligned(a, offset=0):
    a = np.ascontiguousarray(a)
    raw = np.empty(a.nbytes + 128, np.uint8)
    off = (-raw.ctypes.data) % 64 + offset
    out = raw[off:off + a.nbytes].view(a.dtype)
    out[...] = a
    return out


def check_projection(oracle, case, make_fetch_result, **opts):
    """oracle vs product for one case; make_fetch_result(case) -> FetchResult over the case's fragments"""
    from heavydb_amd.executor import Executor
    make_join = opts.pop("make_join", None)
    plan = case.ra.to_plan()
    if case.join_keys is not None:
        from tests import test_hostsim_flow as flow
        q, want, code = oracle.execute(plan, case.frags, case.inner, flow._oracle_join(oracle, case))
        hj, keep_join = (make_join or flow._build_join)(case)
        case.ra.join_table = hj
    else:
        q, want, code = oracle.execute(plan, case.frags)
    ex = Executor(0)
    fr = make_fetch_result(case)
    if case.expect_error is not None:
        with pytest.raises(capi.Mi355qError) as ei:
            ex.executeWorkUnit(case.ra, fr, allow_retry=False, **opts)
        if case.expect_error == capi.ERR_UNSUPPORTED:   # (a shape the oracle runs and this family refuses)
            assert ei.value.code == capi.ERR_UNSUPPORTED and code == 0
        elif case.expect_error > 0:
            assert code == case.expect_error and ei.value.code == case.expect_error, (code, ei.value.code)
        else:
            assert code < 0 and ei.value.code < 0, (code, ei.value.code)
            # the product's negative code carries the count the caller needs for the retry
            if case.join_keys is None:
                matched = sum(int(np.count_nonzero(_passes(case, f))) for f in range(len(case.frags)))
                assert ei.value.code == -matched
            else:   # through a join: the joined rows (SQLite counts them, tests/test_sqlite_semantics.py)
                from tests.test_sqlite_semantics import _load, _sql_for
                assert ei.value.code == -len(_load(case).execute(_sql_for(case)).fetchall())
        return None
    assert code == 0
    rs = ex.executeWorkUnit(case.ra, fr, allow_retry=False, **opts)
    qg = rs.getQueryMemDesc()
    qmd_equal(q, qg)
    assert q.desc_type == capi.PROJECTION
    n_live = oracle.row_count(q, want)
    assert rs.rowCount() == n_live
    if case.ra.scan_limit == 0:
        assert rs.totalMatched() == oracle.last_total_matched() == n_live
    else:
        assert rs.totalMatched() >= n_live
    got = rs.getStorage()
    if getattr(case, "join_one_to_many", 0) and case.join_keys is not None:
        # a ONE-TO-MANY table: the order of the row ids inside one key's payload run depends on the build order (mi355q.h; the
        # device fills the runs with atomics), so the entries of ONE outer row may come out permuted among themselves.  What is
        # pinned: the key sequence (row offsets in (fragment, row) order) entry by entry, and per run of equal keys the
        # multiset of rows.  (With a LIMIT the cut may fall inside a run: the last run is then compared as a subset.)
        n = q.entry_count
        kw = want.view(np.int64)[:n] if q.output_columnar else want.view(np.int64).reshape(n, -1)[:, 0]
        kg = got.view(np.int64)[:n] if q.output_columnar else got.view(np.int64).reshape(n, -1)[:, 0]
        assert (kw == kg).all()
        rw, rg = oracle.fetch_rows(q, want), rs.fetch()
        run = np.concatenate([[0], np.cumsum(kw[1:n_live] != kw[:n_live - 1])]) if n_live else np.zeros(0, np.int64)

        def canon(rows):
            iv, dv, nu = (np.asarray(x)[:n_live] for x in rows)
            cols = [run] + [c for t in range(iv.shape[1]) for c in (nu[:, t].astype(np.int64), iv[:, t], dv[:, t])]
            order = np.lexsort(cols[::-1])
            return [c[order] for c in cols]
        cw, cg = canon(rw), canon(rg)
        last = run == (run[-1] if n_live else 0)
        keep = ~last if case.ra.scan_limit else np.ones(n_live, bool)
        for a_, b_ in zip(cw, cg):
            assert (a_[keep] == b_[keep]).all()
        return rs
    if not q.output_columnar:
        # the whole buffer, entry by entry: same rows in the same (fragment, row) order, the tail EMPTY / zero
        compare_buffers(q, want, got, 0.0)
    else:
        raw_w, raw_g = want.view(np.int8), got.view(np.int8)
        n = q.entry_count
        assert (raw_w[:8 * n].view(np.int64) == raw_g[:8 * n].view(np.int64)).all()   # the key column, EMPTY tail included
        for s in range(q.slot_count):
            o = oracle.col_slot_off(q, s)
            assert o == capi.load_library().mi355q_qmd_slot_col_offset(C.byref(qg), s)
            w = q.slot_bytes[s]
            assert (raw_w[o:o + w * n_live] == raw_g[o:o + w * n_live]).all(), s   # (the slot columns' tails are uninitialised)
    compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), 0.0)
    return rs


def _passes(case, f):
    """numpy restatement of the case's plain quals (used only to count matches of the buffer-full case)"""
    cols = case.frags[f]
    ok = np.ones(len(cols[0]), bool)
    for ql in case.ra.simple_quals:
        v = cols[ql.col]
        ok &= {capi.LT: v < ql.literal, capi.GT: v > ql.literal, capi.LE: v <= ql.literal, capi.GE: v >= ql.literal,
               capi.EQ: v == ql.literal, capi.NE: v != ql.literal}[ql.op]
    return ok


def host_fetch_result(case, offset=0):
    from heavydb_amd.executor import FetchResult
    frags = [[aligned(a, offset) for a in cols] for cols in case.frags]
    inner = [aligned(a) for a in case.inner]
    return FetchResult([[a.ctypes.data for a in cols] for cols in frags], [len(cols[0]) for cols in frags],
                       [a.ctypes.data for a in inner], len(inner[0]) if inner else 0, 0, [frags, inner])


JOIN_CASES = proj_cases.build_join_cases()


@pytest.mark.parametrize("case", JOIN_CASES, ids=[c.name for c in JOIN_CASES])
def test_projection_through_a_join_on_the_host_simulation(sim, oracle, case):
    """one output entry per joined row: one-to-one perfect and keyed tables, INNER and LEFT, a nullable INT32 key, inner
    columns of every width read through the matched row (NULL where a LEFT join found none)"""
    rs = check_projection(oracle, case, host_fetch_result)
    if rs is not None and "nothing_matches" not in case.name:
        assert rs.rowCount() > 1000


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_projection_case_on_the_host_simulation(sim, oracle, case):
    rs = check_projection(oracle, case, host_fetch_result)
    if case.name.startswith("expr_filter_"):
        from heavydb_amd.executor import Executor
        route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags])
        assert "k_filter_mask" in route and "k_proj_compact" in route, route
        if rs is not None:
            assert rs.report.variant == 0, rs.report.variant   # the fast member, on the mask
            check_projection(oracle, case, host_fetch_result, flags=capi.OPT_NO_COMPILED_FILTER)   # the general member agrees
    if case.name.startswith("expr_form_") and rs is not None:
        # targets that are `[CAST](column) <op> literal` stay in the fast member (report.variant 0) ...
        assert rs.report.variant == 0, rs.report.variant
        # ... and the general member's interpreter (MI355Q_OPT_LDS_GENERIC_MEMBER keeps the forms away) agrees
        rs2 = check_projection(oracle, case, host_fetch_result, flags=capi.OPT_LDS_GENERIC_MEMBER)
        assert rs2.report.variant != 0


_FAST_SPLIT = [c for c in CASES if c.expect_error is None or c.expect_error != capi.ERR_UNSUPPORTED]


@pytest.mark.parametrize("case", _FAST_SPLIT, ids=[c.name for c in _FAST_SPLIT])
def test_projection_split_route_on_the_host_simulation(sim, oracle, case):
    """round 6: large inputs run the fast member as three kernels (k_proj_mask: the match bits and the tiles' counts;
    k_proj_scan_tiles; k_proj_fast as pass B alone).  pass_rows = -3 forces that route at any size: every case the fused fast
    member takes gives the same buffer (same entries, same order, same errors, same LIMIT cut), the others are untouched."""
    rs = check_projection(oracle, case, host_fetch_result, pass_rows=-3)
    if rs is not None and case.name != "empty_input":   # (no tile: no launch)
        rs0 = check_projection(oracle, case, host_fetch_result, pass_rows=-2)   # the fused launch, forced
        assert (rs.report.variant == 16) == (rs0.report.variant == 0), (rs.report.variant, rs0.report.variant)


@pytest.mark.parametrize("name", ["i32_filter_50pct_3cols", "all_types_nullable_columnar", "encoded_columns"])
def test_projection_unaligned_chunks_take_the_scalar_loads(sim, oracle, name):
    case = next(c for c in CASES if c.name == name)
    check_projection(oracle, case, lambda c: host_fetch_result(c, offset=8))


def test_projection_descriptor_matches_the_reference_rules(oracle):
    """QueryMemoryDescriptor::init, case Projection (QueryMemoryDescriptor.cpp:394-410) + constructor (:507,:540-546)"""
    from heavydb_amd.executor import InputColDescriptor as D, RelAlgExecutionUnit, TargetExpr
    descs = [D(capi.INT8, True), D(capi.INT32), D(capi.DOUBLE, True), D(capi.FLOAT, True)]
    targets = [TargetExpr(capi.PROJECT, c) for c in range(4)]
    lib = capi.load_library()
    for hint in (0, capi.OUTPUT_COLUMNAR):
        for limit, guess, want_entries in ((0, 0, 16384), (0, 5000, 5000), (300, 5000, 300)):
            ra = RelAlgExecutionUnit(descs, targets, max_groups_buffer_entry_guess=guess, scan_limit=limit, output_columnar_hint=hint)
            q = capi.QMD()
            assert lib.mi355q_qmd_init(C.byref(ra.to_plan()), C.byref(q)) == 0
            qmd_equal(oracle.qmd_init(ra.to_plan()), q)
            assert (q.desc_type, q.entry_count, q.group_col_count, q.key_bytes, q.slot_count) == (capi.PROJECTION, want_entries, 1, 8, 4)
            assert list(q.slot_bytes[:4]) == ([1, 4, 8, 4] if hint else [8, 8, 8, 8])
            want_bytes = want_entries * 40 if not hint else sum((w * want_entries + 7) // 8 * 8 for w in (8, 1, 4, 8, 4))
            assert lib.mi355q_qmd_buffer_bytes(C.byref(q)) == want_bytes == oracle.buffer_bytes(q)
    # a target list is all aggregates or none; scan_limit belongs to projections; no GROUP BY
    bad = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, 0), TargetExpr(capi.COUNT)])
    q = capi.QMD()
    assert lib.mi355q_qmd_init(C.byref(bad.to_plan()), C.byref(q)) == capi.ERR_INVALID_PLAN
    bad = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT)], scan_limit=5)
    assert lib.mi355q_qmd_init(C.byref(bad.to_plan()), C.byref(q)) == capi.ERR_INVALID_PLAN
    bad = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT, 0)], groupby_exprs=[1])
    assert lib.mi355q_qmd_init(C.byref(bad.to_plan()), C.byref(q)) == capi.ERR_INVALID_PLAN


@pytest.mark.parametrize("name", ["i32_filter_columnar_3cols", "empty_middle_fragment", "scan_limit_cuts", "scan_limit_cuts_columnar", "many_small_fragments"])
def test_projection_results_are_appended_not_reduced(sim, oracle, name):
    """ResultSet::append (ResultSet.cpp:307-335; Executor::resultsUnion, Execute.cpp:1670-1694): one result per fragment,
    laid one behind the other, holds the rows of the step over all fragments in (fragment, row) order; entry counts and
    total_matched add up; mi355q_result_reduce on two projections does the same"""
    from heavydb_amd.executor import Executor, FetchResult
    case = next((c for c in CASES if c.name == name), None)
    if case is None:
        pytest.skip("no such case in this matrix")
    if case.expect_error is not None or len(case.frags) < 2:
        pytest.skip("needs two fragments and a step that succeeds")
    plan = case.ra.to_plan()
    q, want, code = oracle.execute(plan, case.frags)
    assert code == 0
    ex = Executor(0)
    parts = []
    keep = []
    for f in range(len(case.frags)):
        cols = [aligned(a) for a in case.frags[f]]
        keep.append(cols)
        fr = FetchResult([[a.ctypes.data for a in cols]], [len(cols[0])], [], 0, 0, [cols])
        parts.append(ex.executeWorkUnit(case.ra, fr, allow_retry=False))
    n_each = [p.rowCount() for p in parts]
    whole = parts[0]
    for k, p in enumerate(parts[1:]):
        if k % 2 == 0:
            whole.append(p)
        else:   # (the reduce entry point: the same thing for two projections)
            check_code = capi.load_library().mi355q_result_reduce(whole.handle, p.handle, None)
            assert check_code == 0
    qa = whole.getQueryMemDesc()
    assert qa.entry_count == q.entry_count * len(parts) and whole.rowCount() == sum(n_each)
    if case.ra.scan_limit:   # (how far past the limit the count runs is the kernel's business: check_projection)
        assert whole.totalMatched() >= whole.rowCount()
    else:
        assert whole.totalMatched() == sum(oracle_total(oracle, plan, case, f) for f in range(len(case.frags))) == whole.rowCount()
    # the rows, in order: a scan limit cuts each part on its own, so the union is compared part by part against the oracle
    iv, dv, nl = whole.fetch()
    off = 0
    for f in range(len(case.frags)):
        qf, wf, cf = oracle.execute(plan, [case.frags[f]])
        assert cf == 0
        wi, wd, wn = oracle.fetch_rows(qf, wf)
        n = len(wi)
        assert n == n_each[f]
        assert (iv[off:off + n] == wi).all() and (nl[off:off + n] == wn).all()
        assert ((dv[off:off + n] == wd) | (np.isnan(dv[off:off + n]) & np.isnan(wd))).all()
        off += n
    assert off == len(iv)
    # the tail of the key column is EMPTY_KEY_64
    raw = whole.getStorage().view(np.int64)
    keys = raw[:qa.entry_count] if qa.output_columnar else raw.reshape(qa.entry_count, -1)[:, 0]
    assert (keys[off:] == 0x7FFFFFFFFFFFFFFF).all() and (keys[:off] != 0x7FFFFFFFFFFFFFFF).all()


def oracle_total(oracle, plan, case, f):
    oracle.execute(plan, [case.frags[f]])
    return oracle.last_total_matched()
