"""mi355q_result_export_arrow (Arrow C Data Interface, produced by the library itself from the device-side
ColumnarResults) against the ORACLE: the same step run by oracle.execute on the host, iterated by
oracle.fetch_rows (ResultSet::getNextRow / pair_to_double restated, ResultSetBufferAccessors.h:197-227), and laid
out the way ArrowResultSetConverter::convertToArrow does (int64 / float64 columns, SQL NULL as validity bits).
Baseline tables keep their rows in insertion-dependent slots (in the reference too), so the rows are compared as
a multiset (sorted by their values)."""
import pytest

from tests import cases as cases_mod

pytestmark = pytest.mark.gpu
NAMES = ["perfect_avg_keyless_idx1", "perfect_nullable_key_and_args", "baseline_nullable_args",
         "count_star_filter_i32_lt", "simple_aggs_empty_result", "constrained_baseline_avg_min", "float_baseline",
         "compact_baseline_key32"]
CASES = [c for c in cases_mod.build_cases() if c.name in NAMES]


def _sorted_rows(cols, n_rows):
    # None sorts first; a float column may differ in the last bits between the two sides (FLOAT arguments are
    # summed in single precision, in another order on the device), so the sort key of a float is rounded well
    # above that (the comparison below uses the real values)
    def k(v):
        if v is None:
            return (0, 0.0)
        return (1, float(f"{v:.3e}") if isinstance(v, float) else v)
    order = sorted(range(n_rows), key=lambda r: tuple(k(c[r]) for c in cols))
    return [tuple(c[r] for c in cols) for r in order]


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_native_arrow_export_matches_oracle(case, oracle):
    import pyarrow as pa
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from heavydb_amd.executor import Executor
    from tests.test_gpu_parity import _fetch_result, _upload
    frag_t, inner_t = _upload(torch, case)
    rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    names = [f"c{i}" for i in range(rs.getQueryMemDesc().n_targets)]
    got = rs.to_arrow_native(names)
    # the checker: the oracle's own step and its own iteration of the same plan, on the host
    q, buf, code = oracle.execute(case.ra.to_plan(), case.frags, case.inner, None, n_threads=2)
    assert code == 0
    ival, dval, nul = oracle.fetch_rows(q, buf)
    n_rows = ival.shape[0]
    assert got.schema.names == names
    assert got.num_rows == n_rows
    want_cols, got_cols = [], []
    for t, n in enumerate(names):
        g = got.column(n)
        is_fp = bool(q.target_is_fp[t])
        assert g.type == (pa.float64() if is_fp else pa.int64()), (n, g.type)
        assert g.null_count == int(nul[:, t].sum()), n
        want_cols.append([None if nul[r, t] else (float(dval[r, t]) if is_fp else int(ival[r, t]))
                          for r in range(n_rows)])
        got_cols.append(g.to_pylist())
    from tests.helpers import F32_ATOL, F32_RTOL
    # the parity bar of DESIGN section 2: doubles 1e-9 relative; targets over a FLOAT argument 2e-4 (+ absolute)
    rtol = [max(1e-9, F32_RTOL) if q.target_arg_is_f32[t] else 1e-9 for t in range(len(names))]
    atol = [F32_ATOL if q.target_arg_is_f32[t] else 0.0 for t in range(len(names))]
    for a_row, b_row in zip(_sorted_rows(want_cols, n_rows), _sorted_rows(got_cols, n_rows)):
        for t, (a, b) in enumerate(zip(a_row, b_row)):
            assert (a is None) == (b is None), (a_row, b_row)
            if a is not None:
                assert a == b or (isinstance(a, float) and
                                  abs(a - b) <= rtol[t] * max(abs(a), abs(b)) + atol[t]), (t, a_row, b_row)
