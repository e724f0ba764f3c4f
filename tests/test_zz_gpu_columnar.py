"""GPU parity of columnar output buffers (output_columnar_hint): the HIP library's
columnar result against the oracle's NATIVE columnar step on the whole case matrix, and the result
operations (reduce, row count, iteration, top-k, ColumnarResults, shard pieces) on columnar handles.
The layout itself is pinned on CPU by tests/test_columnar.py."""
import copy
import ctypes as C

import numpy as np
import pytest

from heavydb_amd import capi
from tests import cases as cases_mod
from tests.helpers import columnar_to_rows, compare_buffers, compare_rows, qmd_equal, rowwise_qmd
from tests.test_gpu_parity import _build_join, _fetch_result, _oracle_join, _upload, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu

CASES = [c for c in cases_mod.build_cases() if c.expect_error is None]


def _columnar(case):
    c = copy.copy(case)
    c.ra = copy.copy(case.ra)
    c.ra.output_columnar_hint = capi.OUTPUT_COLUMNAR
    return c


@pytest.mark.parametrize("force_generic", [True, False], ids=["generic", "planned"])
@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_hip_columnar_matches_oracle(torch_cuda, oracle, case, force_generic):
    from heavydb_amd.executor import Executor
    case = _columnar(case)
    plan = case.ra.to_plan()
    if oracle.lib().orc_qmd_init(C.byref(plan), C.byref(capi.QMD())) == capi.ERR_UNSUPPORTED:
        with pytest.raises(capi.Mi355qError) as ei:
            Executor(0).initQueryMemoryDescriptor(case.ra)
        assert ei.value.code == capi.ERR_UNSUPPORTED
        return
    oj = _oracle_join(oracle, case)
    q, want, code = oracle.execute(plan, case.frags, case.inner, oj, n_threads=2)
    assert code == 0 and q.output_columnar == 1
    frag_t, inner_t = _upload(torch_cuda, case)
    hj, keep = _build_join(torch_cuda, case)
    case.ra.join_table = hj
    rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case, frag_t, inner_t), force_generic=force_generic,
                                     allow_retry=False)
    qmd_equal(q, rs.getQueryMemDesc())
    got = rs.getStorage()
    assert got.nbytes == want.nbytes == oracle.buffer_bytes(q)
    qr = rowwise_qmd(q)
    compare_buffers(qr, columnar_to_rows(q, want), columnar_to_rows(q, got), case.fp_rtol)
    assert rs.rowCount() == oracle.row_count(q, want)
    compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), case.fp_rtol)
    keys, slots = rs.columns()
    rows = columnar_to_rows(q, got)
    kq = q.key_bytes // 8
    assert len(keys) == kq and all(np.array_equal(keys[k], rows[:, k]) for k in range(kq))
    if q.slot_width == 8:
        assert all(np.array_equal(slots[s], rows[:, kq + s]) for s in range(q.slot_count))


@pytest.mark.parametrize("name", ["perfect_key_sum_projectkey", "perfect_nullable_args", "baseline_count_avg",
                                  "multi_perfect_2col_keyless", "multi_baseline_i64_2col",
                                  "compact_perfect_nullable_int32_key", "compact_baseline_count_only"])
def test_columnar_result_operations(torch_cuda, oracle, name):
    """reduce of two columnar halves, ColumnarResults, top-k and a caller-owned buffer initialised by
    mi355q_result_create, on columnar handles."""
    from heavydb_amd.executor import Executor
    torch = torch_cuda
    case = _columnar(next(c for c in CASES if c.name == name))
    plan = case.ra.to_plan()
    half = len(case.frags) // 2
    q, a, _ = oracle.execute(plan, case.frags[:half], case.inner)
    _, b, _ = oracle.execute(plan, case.frags[half:], case.inner)
    want = a.copy()
    assert oracle.reduce(q, want, b) == 0
    _, full, _ = oracle.execute(plan, case.frags, case.inner)
    frag_t, inner_t = _upload(torch, case)
    ex = Executor(0)
    c1, c2 = copy.copy(case), copy.copy(case)
    c1.frags, c2.frags = case.frags[:half], case.frags[half:]
    r1 = ex.executeWorkUnit(case.ra, _fetch_result(c1, frag_t[:half], inner_t), allow_retry=False)
    r2 = ex.executeWorkUnit(case.ra, _fetch_result(c2, frag_t[half:], inner_t), allow_retry=False)
    r1.reduce(r2)
    qr = rowwise_qmd(q)
    red = columnar_to_rows(q, r1.getStorage())
    compare_buffers(qr, columnar_to_rows(q, want), red, case.fp_rtol)
    compare_buffers(qr, columnar_to_rows(q, full), red, case.fp_rtol)
    # the row-wise step of the same decisions holds the same entries
    ra_row = copy.copy(case.ra)
    ra_row.output_columnar_hint = capi.OUTPUT_ROWWISE_COLUMNAR_DECISIONS
    rs_c = ex.executeWorkUnit(case.ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    rs_r = ex.executeWorkUnit(ra_row, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    assert rs_r.getQueryMemDesc().output_columnar == 0
    compare_buffers(qr, rs_r.getStorage(), columnar_to_rows(q, rs_c.getStorage()), case.fp_rtol)
    # ColumnarResults and top-k of the columnar handle = those of the row-wise one
    cols_c, n1 = rs_c.to_columns(torch)
    cols_r, n2 = rs_r.to_columns(torch)
    assert n1 == n2 == rs_c.rowCount() == oracle.row_count(q, full)
    if q.desc_type != capi.GROUP_BY_BASELINE_HASH:  # both in entry order
        for t, (x, y) in enumerate(zip(cols_c, cols_r)):
            xi, yi = x.cpu().numpy(), y.cpu().numpy()
            if q.target_is_fp[t]:
                assert np.allclose(xi.view(np.float64), yi.view(np.float64), rtol=max(case.fp_rtol, 2e-4), atol=0.05,
                                   equal_nan=True)
            else:
                assert np.array_equal(xi, yi)
    k, rq = 5, q.row_size // 8
    oc = torch.zeros((k, rq), dtype=torch.int64, device="cuda")
    orr = torch.zeros((k, rq), dtype=torch.int64, device="cuda")
    t_last = q.n_targets - 1
    assert rs_c.sort(t_last, k, int(oc.data_ptr()), desc=True) == rs_r.sort(t_last, k, int(orr.data_ptr()), desc=True)
    # mi355q_result_create on a caller-owned columnar buffer = initColumnarGroups
    lib = capi.load_library()
    buf = torch.zeros(oracle.buffer_bytes(q) // 8, dtype=torch.int64, device="cuda")
    h = C.c_void_p()
    capi.check(lib.mi355q_result_create(C.byref(q), 0, int(buf.data_ptr()), C.byref(h)), "result_create")
    try:
        torch.cuda.synchronize()
        assert np.array_equal(buf.cpu().numpy(), oracle.init_buffer(q))
        assert lib.mi355q_result_row_count(h) == 0
    finally:
        lib.mi355q_result_free(h)


def test_columnar_at_scale(torch_cuda, oracle):
    """2 M rows through the partitioned baseline family (8-byte key, COUNT + AVG) and the LDS
    perfect-hash family with the columnar hint: equal to the oracle's native columnar step and to the
    row-wise step of the same decisions."""
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, InputColDescriptor, RelAlgExecutionUnit,
                                      TargetExpr)
    torch = torch_cuda
    rng = np.random.default_rng(11)
    n = 2_000_000
    for baseline in (True, False):
        key = ((rng.integers(0, 50_000, n) * 1000003 + 7) if baseline else rng.integers(0, 1000, n)).astype(np.int64)
        val = rng.random(n) * 100.0
        kr = ExpressionRange(False) if baseline else ExpressionRange(True, 0, 999)
        descs = [InputColDescriptor(capi.INT64, False, kr),
                 InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 100.0))]
        dev = [torch.from_numpy(key).cuda(), torch.from_numpy(val).cuda()]
        fr = FetchResult([[int(t.data_ptr()) for t in dev]], [n], keepalive=dev)
        ex = Executor(0)

        def unit(hint):
            return RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.COUNT),
                                               TargetExpr(capi.AVG, 1)], [], [0],
                                       max_groups_buffer_entry_guess=131072, output_columnar_hint=hint)
        rs_c = ex.executeWorkUnit(unit(capi.OUTPUT_COLUMNAR), fr, allow_retry=False, kernel_variant=2 if baseline else 0)
        rs_r = ex.executeWorkUnit(unit(capi.OUTPUT_ROWWISE_COLUMNAR_DECISIONS), fr, allow_retry=False,
                                  kernel_variant=2 if baseline else 0)
        qc = rs_c.getQueryMemDesc()
        assert qc.output_columnar == 1 and rs_r.getQueryMemDesc().output_columnar == 0
        got = columnar_to_rows(qc, rs_c.getStorage())
        compare_buffers(rowwise_qmd(qc), rs_r.getStorage(), got, 1e-9)
        q_o, want, code = oracle.execute(unit(capi.OUTPUT_COLUMNAR).to_plan(), [[key, val]], n_threads=4)
        assert code == 0
        compare_buffers(rowwise_qmd(qc), columnar_to_rows(q_o, want), got, 1e-9)
        assert rs_c.rowCount() == rs_r.rowCount() == (50_000 if baseline else 1000)
