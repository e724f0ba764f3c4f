"""CPU parity of the PRODUCT's plan-time layout decisions (csrc/plan.cpp) and per-row logic
(csrc/rowfunc.h, host-emulated with -DMQ_EMU) against the oracle, over the whole query-shape
matrix.  The GPU tests (test_gpu_parity.py) run the same matrix through the HIP library."""
import ctypes as C

import numpy as np
import pytest

from heavydb_amd import capi
from tests import cases as cases_mod
from tests.helpers import compare_buffers, emu_lib, qmd_equal

CASES = cases_mod.build_cases()


def _oracle_join(oracle, case):
    if case.join_keys is None:
        return None
    r = case.join_range
    return oracle.OracleJoin(case.join_keys, case.join_key_type, r.min, r.max,
                             nullable=case.join_key_nullable, prefer_baseline=case.join_prefer_baseline,
                             one_to_many=case.join_one_to_many)


def _emu_execute(case, plan, oj):
    lib = emu_lib()
    n_frags, n_cols = len(case.frags), plan.n_cols
    flat = (C.c_void_p * max(1, n_frags * n_cols))()
    rows = (C.c_int64 * max(1, n_frags))()
    for f, cols in enumerate(case.frags):
        for c, a in enumerate(cols):
            flat[f * n_cols + c] = a.ctypes.data
        rows[f] = len(cols[0])
    inner = (C.c_void_p * max(1, len(case.inner)))()
    for c, a in enumerate(case.inner):
        inner[c] = a.ctypes.data
    inp = capi.Inputs()
    inp.n_frags = n_frags
    inp.col_buffers = C.cast(flat, C.POINTER(C.c_void_p))
    inp.num_rows = C.cast(rows, C.POINTER(C.c_int64))
    inp.inner_col_buffers = C.cast(inner, C.POINTER(C.c_void_p))
    inp.inner_num_rows = len(case.inner[0]) if case.inner else 0
    q = capi.QMD()
    assert lib.emu_qmd_init(C.byref(plan), C.byref(q)) == 0
    buf = (np.zeros(lib.emu_buffer_bytes(C.byref(q)) // 8, dtype=np.int64) if q.output_columnar
           else np.empty((q.entry_count, q.row_size // 8), dtype=np.int64))
    jt, jbuf, jmin, jmax, jn, jk, jw = 0, None, 0, 0, 0, 1, 8
    if oj is not None:
        info = oj.info()
        sh = oj.shape()
        jt = info["hash_type"]
        jb = oj.raw()
        jbuf = jb.ctypes.data
        jn = info["entry_count"]
        jk, jw = sh["key_components"], sh["component_width"]
        jmin, jmax = case.join_range.min, case.join_range.max
        keep = jb  # noqa: F841
    out_q = capi.QMD()
    code = lib.emu_execute(C.byref(plan), C.byref(inp), jt, jbuf, jmin, jmax, jn, jk, jw, buf.ctypes.data,
                           C.byref(out_q))
    return out_q, buf, code


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_emulated_rowlogic_matches_oracle(oracle, case):
    plan = case.ra.to_plan()
    oj = _oracle_join(oracle, case)
    q, want, code = oracle.execute(plan, case.frags, case.inner, oj, n_threads=3)
    eq, got, ecode = _emu_execute(case, plan, oj)
    if case.expect_error is not None:
        if case.expect_error > 0:   # a persistent error code (enums.h:30-51), e.g. 7 = OVERFLOW_OR_UNDERFLOW
            assert code == ecode == case.expect_error, (code, ecode)
        else:                       # < 0: ran out of group slots
            assert code < 0 and ecode < 0
        return
    assert code == 0 and ecode == 0, (code, ecode)
    qmd_equal(q, eq)
    compare_buffers(q, want, got, case.fp_rtol)


def test_oracle_thread_count_invariance(oracle):
    """kernel-per-fragment + pairwise reduce == single kernel (ints exact)."""
    for case in CASES:
        if case.expect_error is not None or not case.frags:
            continue
        plan = case.ra.to_plan()
        oj = _oracle_join(oracle, case)
        q1, b1, c1 = oracle.execute(plan, case.frags, case.inner, oj, n_threads=1)
        q4, b4, c4 = oracle.execute(plan, case.frags, case.inner, oj, n_threads=4)
        assert c1 == 0 and c4 == 0
        compare_buffers(q1, b1, b4, 1e-12)


def test_emulated_reduce_matches_oracle(oracle):
    """this += that through the product's reduce code == ResultSetStorage::reduce restated."""
    lib = emu_lib()
    for case in CASES:
        if case.expect_error is not None or len(case.frags) < 2:
            continue
        plan = case.ra.to_plan()
        oj = _oracle_join(oracle, case)
        half = len(case.frags) // 2
        q, a, ca = oracle.execute(plan, case.frags[:half], case.inner, oj)
        _, b, cb = oracle.execute(plan, case.frags[half:], case.inner, oj)
        assert ca == 0 and cb == 0
        want = a.copy()
        assert oracle.reduce(q, want, b) == 0
        got = a.copy()
        assert lib.emu_reduce(C.byref(q), got.ctypes.data, b.ctypes.data, q.entry_count) == 0
        compare_buffers(q, want, got, case.fp_rtol)
        # and the merged halves equal the single pass
        _, full, cf = oracle.execute(plan, case.frags, case.inner, oj)
        assert cf == 0
        compare_buffers(q, full, want, case.fp_rtol)
