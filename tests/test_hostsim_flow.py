"""The library's HOST code on the CPU: api.cpp (route selection, derived plans, retries, workspaces, result layouts),
plan.cpp and the kernels of kernels_generic.hip run here against a stand-in HIP runtime (tests/hostsim: "device"
memory is host memory poisoned with 0xA5, a launch runs the kernel block by block on a pool of host threads with
working barriers / shuffles / atomics); the fast kernel families are replaced by row-function stand-ins that keep
their eligibility rules and their ways of handing a step back (tests/hostsim/kernels_host.cpp).  Results are held
against the oracle exactly as the gpu tests do.  What this covers that no other CPU test does: execute_projected,
execute_packed_multi, execute_multi_value, the 4-byte-slot and columnar twins, the LDS retry chain, the spill
re-run, async steps, workspace reservation — the code a GPU box otherwise sees first."""
import ctypes as C
import os

import numpy as np
import pytest

from heavydb_amd import capi
from tests import cases as cases_mod
from tests.helpers import compare_buffers, compare_rows, hostsim_lib, qmd_equal

CASES = cases_mod.build_cases()
ALL_ROUTES = 0xFFFFFFFF
F_SCAN_COUNT, F_SCAN_AGG, F_PERFECT_LDS, F_LDS_GROUPBY, F_BASELINE_DIRECT, F_BASELINE_PART, F_JOIN_SUM = range(7)


@pytest.fixture(scope="module")
def sim():
    """the simulated library bound into heavydb_amd.capi for the duration of this module"""
    lib = capi.load_library(hostsim_lib())
    lib.hostsim_configure.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
    lib.hostsim_configure.restype = None
    lib.hostsim_launches.argtypes = [C.c_int32]
    lib.hostsim_launches.restype = C.c_int32
    lib.hostsim_live_allocations.restype = C.c_int
    lib.hostsim_fail_allocs_larger_than.argtypes = [C.c_size_t]
    lib.hostsim_fail_allocs_larger_than.restype = None
    saved = capi._lib
    capi._lib = lib
    lib.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    yield lib
    capi._lib = saved


def _aligned(a):
    """a copy of the array on a 64-byte boundary (the fast families ask for 16-byte aligned chunks)"""
    a = np.ascontiguousarray(a)
    raw = np.empty(a.nbytes + 64, np.uint8)
    off = (-raw.ctypes.data) % 64
    out = raw[off:off + a.nbytes].view(a.dtype)
    out[...] = a
    return out


def _fetch_result(case):
    from heavydb_amd.executor import FetchResult
    frags = [[_aligned(a) for a in cols] for cols in case.frags]
    inner = [_aligned(a) for a in case.inner]
    return FetchResult([[a.ctypes.data for a in cols] for cols in frags], [len(cols[0]) for cols in frags],
                       [a.ctypes.data for a in inner], len(inner[0]) if inner else 0, 0, [frags, inner])


def _build_join(case):
    from heavydb_amd.executor import HashJoin
    if case.join_keys is None:
        return None, None
    multi = isinstance(case.join_keys, (list, tuple))
    keys = [_aligned(k) for k in (case.join_keys if multi else [case.join_keys])]
    hj = HashJoin.getInstance([k.ctypes.data for k in keys] if multi else keys[0].ctypes.data, len(keys[0]),
                              case.join_key_type, case.join_range, key_nullable=case.join_key_nullable,
                              prefer_baseline=case.join_prefer_baseline, one_to_many=case.join_one_to_many)
    return hj, keys


def _oracle_join(oracle, case):
    if case.join_keys is None:
        return None
    r = case.join_range
    return oracle.OracleJoin(case.join_keys, case.join_key_type, r.min, r.max, nullable=case.join_key_nullable,
                             prefer_baseline=case.join_prefer_baseline, one_to_many=case.join_one_to_many)


def _check(oracle, case, **opts):
    from heavydb_amd.executor import Executor
    q, want, code = oracle.execute(case.ra.to_plan(), case.frags, case.inner, _oracle_join(oracle, case), n_threads=2)
    hj, keep = _build_join(case)
    case.ra.join_table = hj
    try:
        ex = Executor(0)
        fr = _fetch_result(case)
        if case.expect_error is not None:
            with pytest.raises(capi.Mi355qError) as ei:
                ex.executeWorkUnit(case.ra, fr, allow_retry=False, **opts)
            if case.expect_error > 0:
                assert code == case.expect_error and ei.value.code == case.expect_error, (code, ei.value.code)
            else:
                assert code < 0 and (ei.value.code < 0 or ei.value.code == capi.ERR_OUT_OF_SLOTS)
            return None
        assert code == 0
        rs = ex.executeWorkUnit(case.ra, fr, allow_retry=False, **opts)
        qmd_equal(q, rs.getQueryMemDesc())
        compare_buffers(q, want, rs.getStorage(), case.fp_rtol)
        assert rs.rowCount() == oracle.row_count(q, want)
        compare_rows(q, oracle.fetch_rows(q, want), rs.fetch(), case.fp_rtol)
        return rs
    finally:
        case.ra.join_table = None


@pytest.mark.parametrize("variant", [0, 2], ids=["planned", "large_input_members"])
@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_case_matrix_through_the_host_library(sim, oracle, case, variant):
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    before = sim.hostsim_live_allocations()
    rs = _check(oracle, case, kernel_variant=variant)
    del rs
    # every step returns what it took (join tables and results are freed with their handles)
    import gc
    gc.collect()
    assert sim.hostsim_live_allocations() <= before + 8, (before, sim.hostsim_live_allocations())


@pytest.mark.parametrize("case", [c for c in CASES if c.expect_error is None][::3],
                         ids=[c.name for c in CASES if c.expect_error is None][::3])
def test_case_matrix_row_kernel_only(sim, oracle, case):
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    _check(oracle, case, force_generic=True)


# ---- the reference's benchmark queries (tools/refbench.py): projected keys / arguments, several value columns, packed
# 4-byte-width keys — the derived-plan routes of api.cpp, forced with kernel_variant 2 on a small table
import sys  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import refbench  # noqa: E402

QUERIES = refbench.queries()


def _refbench_case(oracle, name, n_rows, card_cap):
    names, descs, gens = refbench.schema(card_cap)
    cols = [oracle.generate_column(n_rows, g[0], g[1], g[2], g[3], g[4], g[5]) for g in gens]
    cut = (n_rows // 3) // 4 * 4
    ra, _ = refbench.build_unit(QUERIES[name], names, descs, n_rows)
    return cases_mod.Case(name, ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]])


@pytest.mark.parametrize("name", list(QUERIES), ids=list(QUERIES))
def test_refbench_queries_large_input_routes(sim, oracle, name):
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    rs = _check(oracle, _refbench_case(oracle, name, 6000, 600), kernel_variant=2)
    assert rs is not None


@pytest.mark.parametrize("name", ["NGA03", "PHS002", "PHM002", "BH002", "BH008", "MSBS002", "MSBS006", "MSPHM003", "MSPHS011"])
def test_refbench_queries_planned_routes(sim, oracle, name):
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    _check(oracle, _refbench_case(oracle, name, 6000, 600), kernel_variant=0)


def test_several_value_columns_are_zipped(sim, oracle):
    """MSBS002: GROUP BY CAST(x AS FLOAT) with aggregates over three value columns: one run per value column
    (execute_multi_value), zipped into a table that must have been initialised first — under this runtime fresh memory
    is 0xA5 everywhere, so a zip into an uninitialised table cannot pass by luck (the round-3 regression)."""
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    case = _refbench_case(oracle, "MSBS002", 6000, 600)
    rs = _check(oracle, case, kernel_variant=2)
    assert rs.report.n_launches >= 3, rs.report.n_launches


# ---- the ways a family hands a step back
def _baseline_case(oracle, n_groups, n_rows=5000, seed=3):
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(seed)
    key = (rng.integers(0, n_groups, n_rows) * 1000003).astype(np.int64)
    val = rng.integers(-50, 50, n_rows).astype(np.int64)
    ra = RelAlgExecutionUnit([InputColDescriptor(capi.INT64, False, ExpressionRange(False)),
                              InputColDescriptor(capi.INT64, False, ExpressionRange(True, -50, 49))],
                             [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1),
                              TargetExpr(capi.MIN, 1)], [], [0], max_groups_buffer_entry_guess=4096, num_tuples=n_rows)
    return cases_mod.Case("b", ra, [[key[:n_rows // 2], val[:n_rows // 2]], [key[n_rows // 2:], val[n_rows // 2:]]])


@pytest.mark.parametrize("groups,attempts,final", [(20, 1, "k_groupby_lds"), (120, 2, "k_groupby_lds"),
                                                   (1000, 3, "k_groupby_lds"), (1500, 3, "k_baseline_direct")])
def test_lds_replica_overflow_retry_chain(sim, oracle, groups, attempts, final):
    """a baseline table's group count is only known afterwards: small replicas, then the largest, then eight windows of
    the largest, then another family"""
    sim.hostsim_configure(ALL_ROUTES, 64, 150, 0)
    rs = _check(oracle, _baseline_case(oracle, groups))
    assert sim.hostsim_launches(F_LDS_GROUPBY) == attempts
    assert rs.report.kernel_name.decode() == final


def test_partitioned_spill_overflow_reruns_with_the_direct_member(sim, oracle):
    sim.hostsim_configure(ALL_ROUTES & ~(1 << F_LDS_GROUPBY), 0, 0, 1)
    rs = _check(oracle, _baseline_case(oracle, 700), kernel_variant=2)
    assert sim.hostsim_launches(F_BASELINE_PART) == 1 and sim.hostsim_launches(F_BASELINE_DIRECT) == 1
    assert rs.report.kernel_name.decode() == "k_baseline_direct"


def test_async_step_and_reserved_workspace(sim, oracle):
    from heavydb_amd.executor import Executor
    sim.hostsim_configure(ALL_ROUTES & ~(1 << F_LDS_GROUPBY), 0, 0, 0)
    case = _baseline_case(oracle, 300)
    q, want, code = oracle.execute(case.ra.to_plan(), case.frags, n_threads=2)
    ex = Executor(0)
    fr = _fetch_result(case)
    held = ex.reserveWorkspace(case.ra, fr, kernel_variant=2)
    assert held >= sum(fr.num_rows) * 16          # the partitioned member's scratch, sized before the first step
    rs, pend = ex.executeWorkUnitAsync(case.ra, fr, kernel_variant=2)
    rs2 = ex.executeWorkUnit(case.ra, fr, allow_retry=False)   # the next call on the device finishes the pending step first
    done = pend.wait()
    compare_buffers(q, want, done.getStorage())
    compare_buffers(q, want, rs2.getStorage())
    # and an error surfaces at wait(), not before
    small = _baseline_case(oracle, 3000)
    small.ra.max_groups_buffer_entry_guess = 1024
    rs3, pend3 = ex.executeWorkUnitAsync(small.ra, _fetch_result(small), kernel_variant=1)
    with pytest.raises(capi.Mi355qError):
        pend3.wait()


def test_out_of_device_memory_is_an_error_code(sim, oracle):
    """hipMalloc failing for the result table / the scratch: ERR_OUT_OF_GPU_MEM, nothing leaked, the next step works"""
    from heavydb_amd.executor import Executor
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    case = _baseline_case(oracle, 300)
    fr = _fetch_result(case)
    ex = Executor(0)
    ex.executeWorkUnit(case.ra, fr, allow_retry=False)
    import gc
    gc.collect()
    before = sim.hostsim_live_allocations()
    sim.hostsim_fail_allocs_larger_than(64 * 1024)
    try:
        with pytest.raises(capi.Mi355qError) as ei:
            ex.executeWorkUnit(case.ra, fr, allow_retry=False)
        assert ei.value.code == 2   # MI355Q_ERR_OUT_OF_GPU_MEM
    finally:
        sim.hostsim_fail_allocs_larger_than(2**62)
    gc.collect()
    assert sim.hostsim_live_allocations() <= before
    _check(oracle, case)


# ---- Arrow C Data Interface export (mi355q_result_export_arrow) on the host runtime
def _arrow_case():
    return next(c for c in CASES if c.name == "baseline_nullable_args")


def test_arrow_export_matches_oracle_rows(sim, oracle):
    import pyarrow as pa
    from heavydb_amd.executor import Executor
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    case = _arrow_case()
    rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case), allow_retry=False)
    q, buf, code = oracle.execute(case.ra.to_plan(), case.frags, case.inner, None, n_threads=2)
    assert code == 0
    ival, dval, nul = oracle.fetch_rows(q, buf)
    names = [f"c{i}" for i in range(q.n_targets)]
    got = rs.to_arrow_native(names)
    assert got.schema.names == names and got.num_rows == ival.shape[0]
    key = lambda row: tuple((0, 0) if v is None else (1, v) for v in row)  # noqa: E731
    want_rows = sorted((tuple(None if nul[r, t] else (float(dval[r, t]) if q.target_is_fp[t] else int(ival[r, t]))
                              for t in range(q.n_targets)) for r in range(ival.shape[0])), key=key)
    got_rows = sorted(zip(*[got.column(n).to_pylist() for n in names]), key=key)
    for a, b in zip(want_rows, got_rows):
        for x, y in zip(a, b):
            assert (x is None) == (y is None) and (x is None or x == y or abs(x - y) <= 1e-9 * max(abs(x), abs(y))), (a, b)
    for t, n in enumerate(names):
        assert got.column(n).type == (pa.float64() if q.target_is_fp[t] else pa.int64())


def test_arrow_child_moved_out_survives_its_parent(sim, oracle):
    """Arrow C Data Interface, "moving child arrays": a consumer copies a child struct out of the parent, marks the
    source released and releases the parent FIRST; the moved child (its name, its buffers) must stay valid until its own
    release runs (the names used to live in the parent's private data)."""
    from heavydb_amd.executor import Executor
    sim.hostsim_configure(ALL_ROUTES, 0, 0, 0)
    case = _arrow_case()
    rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case), allow_retry=False)

    class Schema(C.Structure):
        pass
    Schema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_void_p), ("flags", C.c_int64),
                       ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(Schema))), ("dictionary", C.c_void_p),
                       ("release", C.CFUNCTYPE(None, C.POINTER(Schema))), ("private_data", C.c_void_p)]

    class Array(C.Structure):
        pass
    Array._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                      ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)), ("children", C.POINTER(C.POINTER(Array))),
                      ("dictionary", C.c_void_p), ("release", C.CFUNCTYPE(None, C.POINTER(Array))), ("private_data", C.c_void_p)]
    assert C.sizeof(Schema) == 72 and C.sizeof(Array) == 80
    sch, arr = Schema(), Array()
    names = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta"][:rs.getQueryMemDesc().n_targets]
    c_names = (C.c_char_p * len(names))(*names)
    rc = sim.mi355q_result_export_arrow(rs.handle, c_names, C.addressof(sch), C.addressof(arr), None)
    assert rc == 0 and sch.n_children == len(names) and arr.n_children == len(names)
    # move child 1 of both out: copy the struct, mark the source released
    moved_s, moved_a = Schema(), Array()
    C.memmove(C.addressof(moved_s), C.addressof(sch.children[1].contents), 72)
    C.memmove(C.addressof(moved_a), C.addressof(arr.children[1].contents), 80)
    C.memset(C.addressof(sch.children[1].contents) + Schema.release.offset, 0, 8)
    C.memset(C.addressof(arr.children[1].contents) + Array.release.offset, 0, 8)
    n_rows = arr.length
    sch.release(C.byref(sch))
    arr.release(C.byref(arr))
    assert not sch.release and not arr.release
    import gc
    gc.collect()
    junk = [bytes(200) for _ in range(2000)]   # (give a dangling name every chance to be overwritten)
    assert moved_s.name == b"beta" and moved_s.format in (b"l", b"g")
    assert moved_a.length == n_rows and moved_a.n_buffers == 2
    vals = (C.c_int64 * max(int(n_rows), 1)).from_address(moved_a.buffers[1])
    assert n_rows == 0 or isinstance(vals[int(n_rows) - 1], int)
    moved_s.release(C.byref(moved_s))
    moved_a.release(C.byref(moved_a))
    assert not moved_s.release and not moved_a.release
    del junk


def test_the_fragment_table_is_uploaded_once_for_repeated_steps(sim, oracle):
    """a prepared step over resident columns sends the same column table every time: the second call skips the copy (and its
    DMA command ahead of the kernel); a changed row count, or error words a step left non-zero, send it again"""
    from heavydb_amd.executor import Executor, FetchResult
    sim.hostsim_h2d_async_copies.restype = C.c_int
    sim.hostsim_configure(ALL_ROUTES, 64, 150, 0)
    case = _baseline_case(oracle, 20)
    q, want, code = oracle.execute(case.ra.to_plan(), case.frags, n_threads=2)
    ex = Executor(0)
    fr = _fetch_result(case)
    rs = ex.executeWorkUnit(case.ra, fr, allow_retry=False)
    compare_buffers(q, want, rs.getStorage(), case.fp_rtol)
    c0 = sim.hostsim_h2d_async_copies()
    for _ in range(3):
        rs = ex.executeWorkUnit(case.ra, fr, allow_retry=False)
        compare_buffers(q, want, rs.getStorage(), case.fp_rtol)
    assert sim.hostsim_h2d_async_copies() == c0, "the same table was uploaded again"
    # fewer rows in the last fragment, the same pointers: a different table
    n_last = len(case.frags[1][0]) - 7
    short = [case.frags[0], [a[:n_last] for a in case.frags[1]]]
    q2, want2, _ = oracle.execute(case.ra.to_plan(), short, n_threads=2)
    fr2 = FetchResult(fr.col_buffers, [fr.num_rows[0], n_last], keepalive=[fr])
    rs = ex.executeWorkUnit(case.ra, fr2, allow_retry=False)
    compare_buffers(q2, want2, rs.getStorage(), case.fp_rtol)
    assert sim.hostsim_h2d_async_copies() > c0
    # a step whose LDS replicas overflow leaves a non-zero error word: the next call must lay the zeros down again
    big = _baseline_case(oracle, 120)
    qb, wantb, _ = oracle.execute(big.ra.to_plan(), big.frags, n_threads=2)
    frb = _fetch_result(big)
    for _ in range(2):
        rs = ex.executeWorkUnit(big.ra, frb, allow_retry=False)
        compare_buffers(qb, wantb, rs.getStorage(), big.fp_rtol)
