"""The PROJECTION family on the device (kernels_proj.hip through the C-ABI):
  * every case of tests/proj_cases.py against the oracle — whole buffers, entry by entry (the family keeps the
    (fragment, row) order of the reference's CPU executor);
  * 1 B-row tables at selectivity 1 % / 50 % / 99 % with 1 / 3 / 6 projected columns: the first fragment against the
    oracle, the whole result through size-independent properties that pin it completely — the match count equals the
    count of the (independent) scan-count family and the expected binomial mass; the keys (row offsets) strictly
    increase inside every fragment, so no row appears twice; every emitted row satisfies the predicate and carries
    exactly the input values of its offset (checked on the device against the input columns)."""
from __future__ import annotations

import numpy as np
import pytest

from heavydb_amd import capi
from tests import proj_cases
from tests.helpers import compare_buffers
from tests.test_projection import check_projection

pytestmark = pytest.mark.gpu
CASES = proj_cases.build_cases()


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


def device_fetch_result(torch, case):
    from heavydb_amd.executor import FetchResult
    frags = [[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in cols] for cols in case.frags]
    inner = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in case.inner]
    return FetchResult([[int(t.data_ptr()) for t in cols] for cols in frags], [len(cols[0]) for cols in case.frags],
                       [int(t.data_ptr()) for t in inner], len(case.inner[0]) if case.inner else 0, 0, [frags, inner])


JOIN_CASES = proj_cases.build_join_cases()


@pytest.mark.parametrize("case", JOIN_CASES, ids=[c.name for c in JOIN_CASES])
def test_projection_through_a_join_on_the_device(torch_cuda, oracle, case):
    """SELECT t.a, d.w, ... FROM t JOIN d ON ...: one entry per joined row, inner columns through the matched row id"""
    from tests.test_gpu_parity import _build_join
    rs = check_projection(oracle, case, lambda c: device_fetch_result(torch_cuda, c), make_join=lambda c: _build_join(torch_cuda, c))
    if rs is not None:
        assert rs.report.kernel_name.decode() == "k_proj_compact"


@pytest.mark.parametrize("case", [c for c in JOIN_CASES if c.expect_error is None and not c.ra.scan_limit],
                         ids=[c.name for c in JOIN_CASES if c.expect_error is None and not c.ra.scan_limit])
def test_projection_through_a_join_on_the_device_agrees_with_sqlite(torch_cuda, case):
    """the DEVICE's rows against SQLite's JOIN / LEFT JOIN over the same tables (the arbiter of the reference's own
    ExecuteTest): a pin of the joined-row semantics that owes nothing to the oracle's restatement of the join loop"""
    from heavydb_amd.executor import Executor
    from tests.test_gpu_parity import _build_join
    from tests.test_sqlite_semantics import _key, projection_rows_sqlite, rows_agree
    hj, keep = _build_join(torch_cuda, case)
    case.ra.join_table = hj
    try:
        rs = Executor(0).executeWorkUnit(case.ra, device_fetch_result(torch_cuda, case), allow_retry=False)
    finally:
        case.ra.join_table = None
    q = rs.getQueryMemDesc()
    iv, dv, nu = rs.fetch()
    got = sorted((tuple(None if nu[r, t] else float(dv[r, t]) if q.target_is_fp[t] else int(iv[r, t]) for t in range(q.n_targets))
                  for r in range(iv.shape[0])), key=_key)
    fp = [bool(q.target_is_fp[t]) for t in range(q.n_targets)]
    want = sorted((tuple(float(v) if f and v is not None else v for v, f in zip(r, fp)) for r in projection_rows_sqlite(case)), key=_key)
    rows_agree(case, q, want, got)


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_projection_case_on_the_device(torch_cuda, oracle, case):
    rs = check_projection(oracle, case, lambda c: device_fetch_result(torch_cuda, c))
    if rs is not None:
        assert rs.report.kernel_name.decode() == "k_proj_compact"


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_projection_split_route_on_the_device(torch_cuda, oracle, case):
    """the fast member's split route (k_proj_mask + k_proj_scan_tiles + pass B; what inputs from 67 M rows take) forced on the
    small cases: whole buffers against the oracle"""
    check_projection(oracle, case, lambda c: device_fetch_result(torch_cuda, c), pass_rows=-3)


def test_projection_larger_random_tables(torch_cuda, oracle):
    """4 M-row versions of the generic cases: hundreds of tiles per fragment, every workgroup busy, look-back chains"""
    for case in proj_cases.build_cases(scale=100):
        if case.name in ("i32_filter_50pct_3cols", "i32_filter_columnar_3cols", "all_types_nullable_columnar", "scan_limit_cuts",
                         "many_small_fragments", "expr_targets", "expr_in_qual_and_case"):
            check_projection(oracle, case, lambda c: device_fetch_result(torch_cuda, c))
            check_projection(oracle, case, lambda c: device_fetch_result(torch_cuda, c), pass_rows=-3)   # the split route


_TABLE = {}


@pytest.mark.parametrize("columnar", [False, True], ids=["rowwise", "columnar"])
@pytest.mark.parametrize("n_out", [1, 3, 6])
@pytest.mark.parametrize("sel", [0.01, 0.5, 0.99])
def test_projection_1b_rows(torch_cuda, oracle, sel, n_out, columnar):
    from heavydb_amd import synth
    from heavydb_amd.executor import Executor, Qual, RelAlgExecutionUnit, TargetExpr
    torch = torch_cuda
    if columnar and n_out != 3:
        pytest.skip("columnar: the 3-column shape only (same kernel, other image)")
    n = 1_000_000_000
    free, _ = torch.cuda.mem_get_info(0)
    if free < 140 * 2**30 and not _TABLE:
        pytest.skip("needs 52 GB of columns + up to 56 GB of output in HBM")
    synth.projection(torch, n, 6, sel, cols_cache=_TABLE)  # (the table with all six value columns, generated once)
    ra, fr, info = synth.projection(torch, n, n_out, sel, columnar=columnar, cols_cache=_TABLE)
    import ctypes
    lib = capi.load_library()
    q = capi.QMD()
    assert lib.mi355q_qmd_init(ctypes.byref(ra.to_plan()), ctypes.byref(q)) == 0
    nbytes = lib.mi355q_qmd_buffer_bytes(ctypes.byref(q))
    raw = torch.empty(nbytes // 8, dtype=torch.int64, device="cuda")   # the result buffer, owned by the caller
    rs = Executor(0).executeWorkUnit(ra, fr, allow_retry=False, out_buffer=int(raw.data_ptr()))
    assert rs.report.kernel_name.decode() == "k_proj_compact"
    total = rs.totalMatched()
    assert rs.rowCount() == total <= q.entry_count
    # the count: an independent family (k_scan_count) and the binomial expectation
    rc = Executor(0).executeWorkUnit(RelAlgExecutionUnit(ra.input_col_descs[:1], [TargetExpr(capi.COUNT)], [Qual(0, capi.LT, info["k"])]),
                                     type(fr)([b[:1] for b in fr.col_buffers], fr.num_rows, keepalive=fr.keepalive))
    assert int(rc.getNextRow()[0]) == total
    assert abs(total / n - sel) < 1e-3
    cols = info["cols"]
    if columnar:
        keys = raw[:q.entry_count][:total]
        slot = lambda s: raw.view(torch.int8)[lib.mi355q_qmd_slot_col_offset(ctypes.byref(q), s):][:8 * total].view(torch.int64)
        tail_keys = raw[:q.entry_count][total:]
    else:
        rows = raw.view(q.entry_count, q.row_size // 8)
        keys = rows[:total, 0]
        slot = lambda s: rows[:total, 1 + s]
        tail_keys = rows[total:, 0]
    assert bool((tail_keys == 2**63 - 1).all())
    # fragment of every output row: the keys restart (decrease or stay) exactly at fragment boundaries
    frag_rows = torch.tensor(fr.num_rows, device="cuda", dtype=torch.int64)
    restart = torch.ones(total, dtype=torch.bool, device="cuda")
    restart[1:] = keys[1:] <= keys[:-1]
    n_restarts = int(restart.sum())
    assert n_restarts <= len(fr.num_rows)          # strictly increasing inside a fragment: no row twice
    # (every fragment of 32 M rows has matches at these selectivities, so the k-th run is fragment k)
    assert n_restarts == len(fr.num_rows)
    frag_of = torch.cumsum(restart.to(torch.int64), 0) - 1
    frag_start = torch.cumsum(frag_rows, 0) - frag_rows
    gpos = keys + frag_start[frag_of]              # row index in the whole (contiguous) column
    assert bool((keys >= 0).all()) and bool((keys < frag_rows[frag_of]).all())
    assert bool((cols[0][gpos] < info["k"]).all())  # every emitted row passes the filter; with the count: exactly the matching set
    for s in range(n_out):
        src = cols[1 + s]
        assert bool((src.view(torch.int64)[gpos] == slot(s)).all()), s
    del raw, restart, frag_of, gpos
    # the first fragment against the oracle, entry by entry
    f0 = fr.num_rows[0]
    host = [c[:f0].cpu().numpy() for c in cols[:1 + n_out]]
    ra.max_groups_buffer_entry_guess = int(f0 * sel * 1.01) + 4096
    qo, want, code = oracle.execute(ra.to_plan(), [host])
    assert code == 0
    n0 = oracle.row_count(qo, want)
    first = type(fr)([fr.col_buffers[0]], [f0], keepalive=fr.keepalive)
    rs0 = Executor(0).executeWorkUnit(ra, first, allow_retry=False)
    assert rs0.rowCount() == n0
    if columnar:
        got = rs0.getStorage().view(np.int8)
        w8 = want.view(np.int8)
        assert (w8[:8 * qo.entry_count] == got[:8 * qo.entry_count]).all()
        for s in range(n_out):
            o = oracle.col_slot_off(qo, s)
            assert (w8[o:o + 8 * n0] == got[o:o + 8 * n0]).all()
    else:
        compare_buffers(qo, want, rs0.getStorage(), 0.0)
