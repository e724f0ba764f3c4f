"""The typed few-groups members of k_groupby_lds under filters, on the device: range quals over up to three INT32 columns,
filters compiled at plan time (atoms + truth table), COUNT(*)-only steps — the shapes of
tests/test_hostsim_real_kernels.py::test_typed_lds_member_*, here at 4 M rows through the C-ABI against the oracle, the
typed member (report.variant 5) next to the run-time-role member (MI355Q_OPT_LDS_GENERIC_MEMBER, variant 4)."""
from __future__ import annotations

import numpy as np
import pytest

from heavydb_amd import capi
from tests.helpers import compare_buffers, compare_rows, qmd_equal
from tests.test_hostsim_real_kernels import _filtered_lds_case, _typed_filter_quals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


def _run(torch, oracle, case, flags=0):
    from heavydb_amd.executor import Executor, FetchResult
    q, want, code = oracle.execute(case.ra.to_plan(), case.frags, n_threads=8)
    assert code == 0
    frags = [[torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in cols] for cols in case.frags]
    fr = FetchResult([[int(t.data_ptr()) for t in cols] for cols in frags], [len(cols[0]) for cols in case.frags], keepalive=[frags])
    rs = Executor(0).executeWorkUnit(case.ra, fr, allow_retry=False, flags=flags)
    qmd_equal(q, rs.getQueryMemDesc())
    compare_buffers(q, want, rs.getStorage(), case.fp_rtol)
    compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), case.fp_rtol)
    return rs


@pytest.mark.parametrize("member", ["typed", "generic"])
@pytest.mark.parametrize("baseline", [False, True], ids=["perfect", "baseline"])
@pytest.mark.parametrize("count_only", [False, True], ids=["values", "count_only"])
@pytest.mark.parametrize("shape", ["two_bounds_one_column", "three_columns", "not_equal", "is_not_null", "is_null", "bound_beyond_int32",
                                   "empty_range", "empty_range_negated"])
def test_typed_members_under_range_filters_on_the_device(torch_cuda, oracle, shape, count_only, baseline, member):
    case = _filtered_lds_case(oracle, _typed_filter_quals()[shape], baseline=baseline, nullable_flt=True, n=4_000_003, count_only=count_only)
    rs = _run(torch_cuda, oracle, case, capi.OPT_LDS_GENERIC_MEMBER if member == "generic" else 0)
    if rs.report.kernel_name.decode() == "k_groupby_lds":
        assert rs.report.variant == (4 if member == "generic" else 5), rs.report.variant


@pytest.mark.parametrize("baseline", [False, True], ids=["perfect", "baseline"])
def test_typed_members_under_a_compiled_filter_on_the_device(torch_cuda, oracle, baseline):
    from heavydb_amd.executor import Expr, Qual
    I32 = capi.INT32
    C_, L = Expr.col, lambda x: Expr.lit(I32, x)
    for e in (C_(2).cmp(capi.EX_LT, L(5)).logical(capi.EX_AND, C_(3).cmp(capi.EX_GT, L(30))).logical(capi.EX_OR, C_(4).is_null()),
              C_(2).cmp(capi.EX_LT, L(0)).logical(capi.EX_OR, C_(3).cmp(capi.EX_GT, L(60))).logical_not()):
        case = _filtered_lds_case(oracle, [Qual(5, capi.EQ, 1)], exprs=[e], baseline=baseline, nullable_flt=True, n=4_000_003)
        rs = _run(torch_cuda, oracle, case)
        assert rs.report.kernel_name.decode() == "k_groupby_lds" and rs.report.variant == 5, (rs.report.kernel_name, rs.report.variant)
        # the interpreter pass (MI355Q_OPT_NO_COMPILED_FILTER) agrees
        _run(torch_cuda, oracle, case, capi.OPT_NO_COMPILED_FILTER)


# ---- round 6: program atoms (regprog.h) on the device — the shapes of test_program_atoms_in_compiled_filters at 2 M rows
from tests.test_hostsim_real_kernels import _prog_atom_case, _prog_atom_shapes  # noqa: E402


@pytest.mark.parametrize("consumer", ["typed_lds", "generic_lds", "scan_agg"])
@pytest.mark.parametrize("name", list(_prog_atom_shapes()))
def test_program_atoms_on_the_device(torch_cuda, oracle, name, consumer):
    from heavydb_amd.executor import Executor, FetchResult
    case = _prog_atom_case(name, grouped=consumer != "scan_agg", typed=consumer == "typed_lds", n=2_000_003)
    route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags])
    assert "filter compiled" in route and "k_project" not in route, route
    if case.expect_error is not None:
        q, want, code = oracle.execute(case.ra.to_plan(), case.frags, n_threads=8)
        assert code == case.expect_error
        frags = [[torch_cuda.from_numpy(np.ascontiguousarray(a)).cuda() for a in cols] for cols in case.frags]
        fr = FetchResult([[int(t.data_ptr()) for t in cols] for cols in frags], [len(cols[0]) for cols in case.frags], keepalive=[frags])
        with pytest.raises(capi.Mi355qError) as ei:
            Executor(0).executeWorkUnit(case.ra, fr, allow_retry=False)
        assert ei.value.code == case.expect_error, ei.value.code
        return
    rs = _run(torch_cuda, oracle, case)
    kn = rs.report.kernel_name.decode()
    assert kn == "k_scan_agg" if consumer == "scan_agg" else kn in ("k_groupby_lds", "k_perfect_lds"), kn
    _run(torch_cuda, oracle, case, capi.OPT_NO_COMPILED_FILTER)   # the interpreter pass agrees


@pytest.mark.parametrize("family,variant", [("k_part_scatter", 2), ("k_baseline_direct", 1), ("k_perfect_lds", 0)])
@pytest.mark.parametrize("name", ["guarded_div", "affine", "double_arithmetic"])
def test_program_atoms_through_the_mask_in_the_large_table_families_on_the_device(torch_cuda, oracle, name, family, variant):
    """the row mask in front of the partitioned GROUP BY (the headline family, 150 K groups over 3 M rows), the direct baseline
    member and the perfect-hash LDS member"""
    from tests.test_hostsim_real_kernels import _mask_large_case
    from heavydb_amd.executor import Executor, FetchResult
    case = _mask_large_case(name, family, n=3_000_003, n_groups=150_000)
    q, want, code = oracle.execute(case.ra.to_plan(), case.frags, n_threads=8)
    assert code == 0
    frags = [[torch_cuda.from_numpy(np.ascontiguousarray(a)).cuda() for a in cols] for cols in case.frags]
    fr = FetchResult([[int(t.data_ptr()) for t in cols] for cols in frags], [len(cols[0]) for cols in case.frags], keepalive=[frags])
    rs = Executor(0).executeWorkUnit(case.ra, fr, allow_retry=False, kernel_variant=variant, flags=capi.OPT_FILTER_PREPASS)
    qmd_equal(q, rs.getQueryMemDesc())
    compare_buffers(q, want, rs.getStorage(), 1e-9)
    compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), 1e-9)
    assert rs.report.kernel_name.decode() == family, rs.report.kernel_name
