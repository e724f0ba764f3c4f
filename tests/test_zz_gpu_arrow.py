"""mi355q_result_export_arrow (Arrow C Data Interface, produced by the library itself from the device-side
ColumnarResults) against the host mirror's pyarrow table built from fetch_rows
(ArrowResultSetConverter::convertToArrow semantics: int64 / float64 columns, SQL NULL as validity bits)."""
import numpy as np
import pytest

from heavydb_amd import capi
from tests import cases as cases_mod

pytestmark = pytest.mark.gpu
NAMES = ["perfect_avg_keyless_idx1", "perfect_nullable_key_and_args", "baseline_nullable_args",
         "count_star_filter_i32_lt", "simple_aggs_empty_result", "constrained_baseline_avg_min", "float_baseline",
         "compact_baseline_key32"]
CASES = [c for c in cases_mod.build_cases() if c.name in NAMES]


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_native_arrow_export_matches_host_mirror(case):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from heavydb_amd.executor import Executor
    from tests.test_gpu_parity import _fetch_result, _upload
    frag_t, inner_t = _upload(torch, case)
    rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
    names = [f"c{i}" for i in range(rs.getQueryMemDesc().n_targets)]
    want = rs.to_arrow(names)
    got = rs.to_arrow_native(names)
    assert got.schema.names == want.schema.names
    assert got.num_rows == want.num_rows
    for n in names:
        w, g = want.column(n), got.column(n)
        assert g.type == w.type, (n, g.type, w.type)
        assert g.null_count == w.null_count, n
        wl, gl = w.to_pylist(), g.to_pylist()
        for a, b in zip(wl, gl):
            assert (a is None) == (b is None)
            if a is not None:
                assert a == b or (isinstance(a, float) and abs(a - b) <= 1e-12 * max(1.0, abs(a))), (n, a, b)
