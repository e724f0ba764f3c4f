"""The REAL kernels on the CPU: kernels_fast.hip (k_scan_count, k_scan_agg, k_perfect_lds, k_perfect_lds_prog,
k_baseline_direct, k_join_sum), kernels_lds.hip (k_groupby_lds), kernels_part.hip (k_part_scatter / k_part_aggregate /
k_spill_merge, the radix and payload join probes, the slice merge) and kernels_sort.hip (top-k selection; the full sort's
hand-written one-sweep radix sort) compiled for the host against the stand-in HIP runtime of tests/hostsim — a
workgroup runs as 1024 cooperative fibers with working barriers, shuffles, ballots, ds_permute, LDS and atomics; the
polling loops of the scatter's producer / flusher pipeline sleep on the device, and s_sleep is a fiber yield here —
behind the real api.cpp / plan.cpp.  Results are held against the oracle exactly as the gpu tests do.  This is a functional check
of the device code itself (index arithmetic, tails, NULL handling, replica folds, flush rules) on a machine without a GPU; it says
nothing about timing or about races (fibers of a block run one after the other between barriers, blocks one after the
other: waits for another workgroup — pair rendezvous, probe pacing — are bounded on the device and expire here)."""
import ctypes as C

import numpy as np

import pytest

from heavydb_amd import capi
from tests import cases as cases_mod
from tests.helpers import hostsim_lib
from tests import test_hostsim_flow as flow

CASES = cases_mod.build_cases()
REAL_FAMILIES = {"k_scan_count", "k_scan_agg", "k_perfect_lds", "k_perfect_lds_prog", "k_baseline_direct", "k_join_sum",
                 "k_groupby_lds", "k_part_scatter"}


@pytest.fixture(scope="module")
def sim():
    lib = capi.load_library(hostsim_lib(real_fast=True))
    lib.hostsim_configure.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
    lib.hostsim_configure.restype = None
    lib.hostsim_live_allocations.restype = C.c_int
    saved = capi._lib
    capi._lib = lib
    lib.hostsim_configure(flow.ALL_ROUTES, 0, 0, 0)
    yield lib
    capi._lib = saved


SEEN = set()


@pytest.mark.parametrize("variant", [0, 1, 2, 3], ids=["planned", "direct_members", "partitioned_members", "payload_probe"])
@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_case_matrix_through_the_real_fast_kernels(sim, oracle, case, variant):
    rs = flow._check(oracle, case, kernel_variant=variant)
    if rs is not None:
        SEEN.add(rs.report.kernel_name.decode())


@pytest.mark.parametrize("name", list(flow.QUERIES), ids=list(flow.QUERIES))
def test_refbench_queries_through_the_real_fast_kernels(sim, oracle, name):
    rs = flow._check(oracle, flow._refbench_case(oracle, name, 6000, 600), kernel_variant=0)
    assert rs is not None
    SEEN.add(rs.report.kernel_name.decode())


def test_zz_every_real_family_ran():
    missing = REAL_FAMILIES - SEEN - {"k_perfect_lds_prog"}
    assert not missing, (missing, SEEN)


# ---- tables that do not fit one LDS: windows (perfect hash: ranges of the entry index; baseline: classes of a key hash)
@pytest.mark.parametrize("member", ["typed", "generic"])
@pytest.mark.parametrize("name", ["PHS004", "PHM003", "BH004", "BH007", "MSPHS002", "MSPHM002", "MSBS002"])
def test_windowed_lds_groupby_on_ten_thousand_groups(sim, oracle, name, member):
    """the reference benchmark's 10 K-group shapes at their real cardinality: 2 - 8 windows per table, through the typed
    member (roles compiled in, report.variant 5) and the run-time-role member (MI355Q_OPT_LDS_GENERIC_MEMBER, variant 4)"""
    case = flow._refbench_case(oracle, name, 40000, 10000)
    rs = flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_LDS_GENERIC_MEMBER if member == "generic" else 0)
    assert rs is not None
    assert rs.report.kernel_name.decode() == "k_groupby_lds", rs.report.kernel_name
    assert rs.report.variant == (4 if member == "generic" else 5), rs.report.variant
    assert rs.rowCount() > 4096


LDS_SHAPES = ["PHS001", "PHS003", "PHM001", "PHM002", "BH001", "BH003", "MSBS001", "MSPHS001", "MSPHM001"]


@pytest.mark.parametrize("member", ["typed", "generic"])
@pytest.mark.parametrize("name", LDS_SHAPES)
def test_small_lds_groupby_both_members(sim, oracle, name, member):
    """the few-groups shapes (replicated tables) through both members; odd row counts so that the quad remainder and the
    tail rows of every fragment are visited"""
    case = flow._refbench_case(oracle, name, 9001, 700)
    rs = flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_LDS_GENERIC_MEMBER if member == "generic" else 0)
    assert rs is not None
    if rs.report.kernel_name.decode() == "k_groupby_lds":
        assert rs.report.variant == (4 if member == "generic" else 5), rs.report.variant


def test_windowed_baseline_gives_up_beyond_eight_windows(sim, oracle):
    """more groups than eight windows hold: the chain ends in another family, the result is still the oracle's"""
    case = flow._baseline_case(oracle, 60000, n_rows=90000)
    case.ra.max_groups_buffer_entry_guess = 65536 * 2
    rs = flow._check(oracle, case)
    assert rs.report.kernel_name.decode() != "k_groupby_lds"
    # close to what eight windows hold (28 K groups in 32 K slots): whichever family ends up with it, the oracle's result
    case = flow._baseline_case(oracle, 30000, n_rows=90000)
    case.ra.max_groups_buffer_entry_guess = 65536
    flow._check(oracle, case)


# ---- mi355q_explain: the route of every reference benchmark query at the benchmark's own sizes, without a GPU
def _explain(name, n_rows):
    from heavydb_amd.executor import Executor
    names, descs, _ = flow.refbench.schema()
    ra, _ = flow.refbench.build_unit(flow.QUERIES[name], names, descs, n_rows)
    frag = flow.refbench.FRAGMENT_ROWS
    rows = [frag] * (n_rows // frag) + ([n_rows % frag] if n_rows % frag else [])
    return Executor(0).explain(ra, rows)


@pytest.mark.parametrize("n_rows", [4 * 32_000_000, 1_000_000_000], ids=["128M", "1B"])
@pytest.mark.parametrize("name", list(flow.QUERIES), ids=list(flow.QUERIES))
def test_no_reference_benchmark_query_takes_the_row_kernel(sim, name, n_rows):
    """none of the reference's 57 synthetic-benchmark steps may end up in k_generic at the sizes the
    benchmark runs (4 x 32 M rows, and 1 B).  The planner is the product's own (api.cpp, plan.cpp, the eligibility rules
    of kernels_fast.hip / kernels_lds.hip compiled for the host); only the partitioned family's rule is a stand-in."""
    route = _explain(name, n_rows)
    assert route and "k_generic" not in route, (name, route)


def test_explain_names_the_stages_of_a_derived_route(sim):
    r = _explain("BH005", 1_000_000_000)      # GROUP BY cast(x100k AS DOUBLE): grouped by x100k itself (100 K-entry perfect hash), then re-keyed
    assert "k_cast_key_emit" in r and r.endswith("k_idx_scatter + k_idx_aggregate"), r
    r = _explain("MSPHS011", 1_000_000_000)   # MAX(x10 + 1): MAX(x10) of a derived plan, the literal added while it is copied
    assert r.startswith("aggregates of column + literal") and r.endswith("k_idx_scatter + k_idx_aggregate"), r
    r = _explain("PHS004", 1_000_000_000)     # 10 K-entry perfect hash, five aggregates: windows of the LDS group-by
    assert r == "k_groupby_lds", r
    r = _explain("NGA03", 1_000_000_000)
    assert r == "k_scan_agg", r
    r = _explain("MSPHS009", 1_000_000_000)   # x10m, two value columns: one exchange of 16-byte {entry, v0, v1} records
    assert r == "k_idx_scatter + k_idx_aggregate", r
    r = _explain("MSBS006", 1_000_000_000)    # (BIGINT stride key, x100), two value columns: packed key, one run per value column
    assert "k_zip_targets" in r and "k_part_scatter" in r, r


def test_derived_routes_leave_plans_whose_expressions_read_expressions(sim):
    """GROUP BY CAST(k AS DOUBLE) takes the cast-key route (the key expression leaves the plan, the rest is renumbered) — unless
    another expression reads an expression's value: then the step is projected as stated."""
    from heavydb_amd.executor import Executor, Expr, ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 99_999)), InputColDescriptor(capi.INT32, True, ExpressionRange(True, -5, 5, True))]
    key = Expr.col(0).cast(capi.DOUBLE).with_range(ExpressionRange(True, 0, 0, False, 0.0, 99_999.0))
    shifted = Expr.col(1).add(Expr.lit(capi.INT32, 1), capi.INT32).with_range(ExpressionRange(True, -4, 6, True))
    reads = Expr.col(2 + 1).cast(capi.INT64).with_range(ExpressionRange(True, -4, 6, True))
    rows = [32_000_000] * 8
    plain = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.MAX, 3)], [], [2], exprs=[key, shifted])
    r = Executor(0).explain(plain, rows)
    assert "k_cast_key_emit" in r and "aggregates of column + literal" in r, r
    composed = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.MAX, 4)], [], [2],
                                   exprs=[key, shifted, reads])
    r = Executor(0).explain(composed, rows)
    assert "k_cast_key_emit" not in r and "column + literal" not in r and "k_project" in r, r


@pytest.mark.parametrize("groups,n_rows", [(20, 5000), (200, 5000), (300, 2000), (1000, 20000), (3000, 20000), (20000, 60000)])
def test_baseline_lds_chain_with_the_real_kernel(sim, oracle, groups, n_rows):
    """small replicas -> the largest replica -> eight windows -> another family, each lost attempt abandoned without a fold or a
    flush; (300 groups, 2 000 rows): every replica holds its share but their union does not fit replica 0 — the fold gives up"""
    case = flow._baseline_case(oracle, groups, n_rows=n_rows)
    case.ra.max_groups_buffer_entry_guess = max(4096, 4 * groups)
    rs = flow._check(oracle, case)
    assert rs is not None
    if groups <= 3000:
        assert rs.report.kernel_name.decode() == "k_groupby_lds", rs.report.kernel_name


@pytest.mark.parametrize("shape", ["count_only_filtered", "sum_min_max_i64"])
def test_partitioned_family_keeps_every_record_with_long_staging_lines(sim, oracle, shape):
    """150 K groups over 16 partitions: 512-record staging lines whose 64 segments are flushed by the 64 lanes of one wave.
    On the device those lanes read `written[p]` in lockstep; the simulation has to hand them ONE reading (tests/helpers.py
    patches it in), otherwise a lane that read earlier flushes its segment of the line a generation late — records of the
    next generation in place of its own.  (Found by running tests/test_gpu_parity.py on the simulation: groups went missing
    while the total count stayed right.)"""
    from heavydb_amd.executor import Executor, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(99)
    n, n_keys = 300_000, 150_000
    key = (rng.integers(0, n_keys, n) * 1000003 + 7).astype(np.int64)
    val = rng.integers(-10**6, 10**6, n).astype(np.int64)
    fil = rng.integers(0, 2**31 - 1, n).astype(np.int32)
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
             InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**6, 10**6)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))]
    if shape == "count_only_filtered":
        targets, quals = [TargetExpr(capi.COUNT)], [Qual(2, capi.GE, 2**29)]
    else:
        targets, quals = [TargetExpr(capi.SUM, 1), TargetExpr(capi.MIN, 1), TargetExpr(capi.MAX, 1), TargetExpr(capi.COUNT)], []
    ra = RelAlgExecutionUnit(descs, targets, quals, [0], max_groups_buffer_entry_guess=2 * n_keys)
    cut = (n // 3) & ~3
    case = cases_mod.Case(shape, ra, [[key[:cut], val[:cut], fil[:cut]], [key[cut:], val[cut:], fil[cut:]]])
    rs = flow._check(oracle, case, kernel_variant=2, scratch_bytes=16 << 20)
    assert rs.report.kernel_name.decode() == "k_part_scatter" and rs.report.variant == 2


@pytest.mark.parametrize("overlap_cus", [2, 4, 6])
@pytest.mark.parametrize("shape", ["count_avg_filtered", "sum_min_max_i64"])
def test_partitioned_family_with_phase_1_next_to_phase_2(sim, oracle, shape, overlap_cus):
    """`tune_overlap_cus`: k_part_scatter of chunk i + 1 on some of the CUs while k_part_aggregate of chunk i runs on the
    others (second stream, two record buffers, two spill lists).  Seven fragments under a small scratch cap =
    seven chunks, so both buffers are reused several times and every later chunk re-loads the table rows its predecessor
    wrote.  The simulation runs a launch when it is enqueued, i.e. it checks the enqueue order, the buffer arithmetic and
    the geometry with fewer scatter workgroups than CUs — not the overlap itself."""
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(1234 + overlap_cus)
    n, n_keys = 280_000, 90_000
    key = (rng.integers(0, n_keys, n) * 1000003 + 7).astype(np.int64)
    fil = rng.integers(0, 2**31 - 1, n).astype(np.int32)
    if shape == "count_avg_filtered":
        val = rng.random(n) * 1000.0
        descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
                 InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 1000.0)),
                 InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))]
        targets, quals = [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1)], [Qual(2, capi.LT, 2**30)]
    else:
        val = rng.integers(-10**6, 10**6, n).astype(np.int64)
        descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
                 InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**6, 10**6)),
                 InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))]
        targets, quals = [TargetExpr(capi.SUM, 1), TargetExpr(capi.MIN, 1), TargetExpr(capi.MAX, 1), TargetExpr(capi.COUNT)], []
    ra = RelAlgExecutionUnit(descs, targets, quals, [0], max_groups_buffer_entry_guess=2 * n_keys)
    cuts = [0] + [(n * k // 7) & ~3 for k in range(1, 7)] + [n]
    frags = [[key[a:b], val[a:b], fil[a:b]] for a, b in zip(cuts[:-1], cuts[1:])]
    case = cases_mod.Case(shape, ra, frags)
    rs = flow._check(oracle, case, kernel_variant=2, scratch_bytes=8 << 20, tune_overlap_cus=overlap_cus)
    assert rs.report.kernel_name.decode() == "k_part_scatter" and rs.report.variant == 2
    assert rs.report.n_launches >= 4, rs.report.n_launches


# ---- perfect-hash tables too large for LDS: the index-partitioned family (kernels_idx.hip) -------------------------------
IDX_SHAPES = ["PHS005", "PHS006", "PHM004", "PHM005", "MSPHS003", "MSPHS006", "MSPHS007", "MSPHS008", "MSPHS010", "MSPHM003",
              "MSPHM006", "S001", "S002"]      # (S00x: COUNT(*) only — 4-byte records that carry the entry index alone)


@pytest.mark.parametrize("name", IDX_SHAPES)
def test_idx_partitioned_family_on_the_benchmark_shapes(sim, oracle, name):
    """PerfectHashSingleCol / MultiCol / MultiStep shapes with 100 K+ entries: k_idx_scatter (8-byte records for one value
    column, 16-byte records for two or three — ONE exchange, no packed key column, no zip) + k_idx_aggregate (LDS table
    indexed by the entry, merged into the table with the reduce rule), kernel_variant 2 = the large-input members on a
    small input.  Row counts that leave a quad remainder and a tail in every fragment."""
    case = flow._refbench_case(oracle, name, 150_003, 120_000)
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs is not None
    # (the benchmark's value columns carry their ranges: the packed 2- or 4-byte word, round 6)
    assert rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.variant in (7, 8), (rs.report.kernel_name, rs.report.variant)
    rs = flow._check(oracle, case, kernel_variant=2, flags=capi.OPT_NO_IDX_PACK)
    assert rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.variant == 6, (rs.report.kernel_name, rs.report.variant)


def test_idx_partitioned_family_packed_records_with_values_outside_their_declared_range(sim, oracle):
    """the packed word is built from the value columns' ExpressionRanges, and those are a HINT: values beyond them (and NULLs
    in a column whose range says it has none, INT32_MIN in a NOT NULL column) leave as full records through the spill list —
    the table is the oracle's whatever the data holds.  Both packed widths, one to three value columns."""
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(79)
    n = 200_003
    i32 = np.iinfo(np.int32)
    for card, want_variant in ((70_000, 8), (3_000_000, 7)):
        key = rng.integers(1, card + 1, n).astype(np.int32)
        v0 = rng.integers(1, 11, n).astype(np.int32)            # declared [1, 10], nullable, "no NULLs"
        v0[rng.random(n) < 0.01] = i32.min                      # ... NULLs all the same (code 0)
        v0[rng.random(n) < 0.002] = 11                          # just outside
        v0[rng.random(n) < 0.002] = -7
        v0[rng.random(n) < 0.001] = i32.max
        v1 = rng.integers(-3, 4, n).astype(np.int32)            # declared [-3, 3], NOT NULL
        v1[rng.random(n) < 0.002] = i32.min                     # the NOT NULL column's INT32_MIN is a value
        v1[rng.random(n) < 0.002] = 1 << 20
        v2 = rng.integers(0, 100, n).astype(np.int32)           # declared [0, 99], nullable, has NULLs
        v2[rng.random(n) < 0.05] = i32.min
        descs = [InputColDescriptor(capi.INT32, True, ExpressionRange(True, 1, card, False)),
                 InputColDescriptor(capi.INT32, True, ExpressionRange(True, 1, 10, False)),
                 InputColDescriptor(capi.INT32, False, ExpressionRange(True, -3, 3, False)),
                 InputColDescriptor(capi.INT32, True, ExpressionRange(True, 0, 99, True))]
        cuts = [0, 50_001, 100_003, n]
        frags = [[key[a:b], v0[a:b], v1[a:b], v2[a:b]] for a, b in zip(cuts[:-1], cuts[1:])]
        for targets in ([TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT, 1), TargetExpr(capi.SUM, 1), TargetExpr(capi.MAX, 1),
                         TargetExpr(capi.MIN, 1), TargetExpr(capi.AVG, 1)],
                        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1), TargetExpr(capi.MIN, 2),
                         TargetExpr(capi.MAX, 2)],
                        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.AVG, 3), TargetExpr(capi.SUM, 2), TargetExpr(capi.MAX, 1),
                         TargetExpr(capi.COUNT, 3)],
                        # (a column of 100 values: no presence mask — the minimum alone, both, neither)
                        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.MIN, 3), TargetExpr(capi.SUM, 1)],
                        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.MIN, 3), TargetExpr(capi.MAX, 3), TargetExpr(capi.COUNT, 2)],
                        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 3), TargetExpr(capi.AVG, 1)]):
            ra = RelAlgExecutionUnit(descs, targets, [], [0], max_groups_buffer_entry_guess=2 * card)
            case = cases_mod.Case("idx_packed_hint", ra, frags)
            rs = flow._check(oracle, case, kernel_variant=2)
            assert rs.report.kernel_name.decode() == "k_idx_scatter", rs.report.kernel_name
            assert rs.report.variant in (7, 8) and rs.report.variant <= max(want_variant, 7), (card, rs.report.variant)
            assert rs.report.spilled_rows > 0
        # a value column without a range: the plain records
        descs2 = list(descs)
        descs2[1] = InputColDescriptor(capi.INT32, True, ExpressionRange(False, 0, 0, False))
        ra = RelAlgExecutionUnit(descs2, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 1)], [], [0], max_groups_buffer_entry_guess=2 * card)
        rs = flow._check(oracle, cases_mod.Case("idx_no_range", ra, frags), kernel_variant=2)
        assert rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.variant == 6, (rs.report.kernel_name, rs.report.variant)


def test_idx_partitioned_family_in_several_chunks_and_with_spills(sim, oracle):
    """a scratch cap that cuts the input into chunks (every chunk merges into the table again), and a skewed key column
    whose hot entries overflow their runs: the spill list is applied record by record afterwards"""
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(77)
    n, card = 260_000, 90_000
    key = rng.integers(1, card + 1, n).astype(np.int32)
    key[rng.random(n) < 0.35] = 4242          # one entry owns a third of the rows: its run overflows
    val = rng.integers(1, 11, n).astype(np.int32)
    val[rng.random(n) < 0.05] = np.iinfo(np.int32).min   # NULLs
    v2 = rng.integers(-1000, 1000, n).astype(np.int32)
    descs = [InputColDescriptor(capi.INT32, True, ExpressionRange(True, 1, card, True)),
             InputColDescriptor(capi.INT32, True, ExpressionRange(True, 1, 10, True)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, -1000, 1000))]
    for targets in ([TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT, 1), TargetExpr(capi.SUM, 1), TargetExpr(capi.MAX, 1),
                     TargetExpr(capi.MIN, 1), TargetExpr(capi.AVG, 1)],
                    [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1), TargetExpr(capi.AVG, 2),
                     TargetExpr(capi.MIN, 2)]):
        ra = RelAlgExecutionUnit(descs, targets, [], [0], max_groups_buffer_entry_guess=2 * card)
        cuts = [0] + [(n * k // 5) & ~3 for k in range(1, 5)] + [n]
        frags = [[key[a:b], val[a:b], v2[a:b]] for a, b in zip(cuts[:-1], cuts[1:])]
        case = cases_mod.Case("idx_skew", ra, frags)
        rs = flow._check(oracle, case, kernel_variant=2, scratch_bytes=18 << 20, flags=capi.OPT_NO_IDX_PACK)
        assert rs.report.kernel_name.decode() == "k_idx_scatter", rs.report.kernel_name
        assert rs.report.n_launches >= 2, rs.report.n_launches
        assert rs.report.spilled_rows > 0
        # the packed word (2-byte records hold eight per unit: the runs are short, one chunk) with a hotter entry
        key2 = key.copy()
        key2[rng.random(n) < 0.9] = 4242
        frags2 = [[key2[a:b], val[a:b], v2[a:b]] for a, b in zip(cuts[:-1], cuts[1:])]
        rs = flow._check(oracle, cases_mod.Case("idx_skew_packed", ra, frags2), kernel_variant=2, scratch_bytes=18 << 20)
        assert rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.variant in (7, 8), (rs.report.kernel_name, rs.report.variant)
        assert rs.report.spilled_rows > 0


def test_idx_partitioned_family_count_only_records(sim, oracle):
    """COUNT(*) / key projections only: 4-byte records that carry the entry index alone (four per 16-byte unit: fragments whose
    record counts leave every remainder), one to three key columns with NULL keys, a hot entry that spills, several chunks —
    with 8-byte slots (bigint_count) and with the 4-byte-slot layout the reference gives this shape"""
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(78)
    n = 260_003
    k0 = rng.integers(1, 90_001, n).astype(np.int32)
    k0[rng.random(n) < 0.35] = 4242
    k0[rng.random(n) < 0.01] = np.iinfo(np.int32).min
    k1 = rng.integers(0, 3, n).astype(np.int32)
    k2 = rng.integers(-1, 1, n).astype(np.int32)
    descs = [InputColDescriptor(capi.INT32, True, ExpressionRange(True, 1, 90_000, True)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2)),
             InputColDescriptor(capi.INT32, True, ExpressionRange(True, -1, 0, False))]
    cuts = [0, 50_001, 100_003, 180_006, n]
    frags = [[k0[a:b], k1[a:b], k2[a:b]] for a, b in zip(cuts[:-1], cuts[1:])]
    for group, big in (([0], True), ([0], False), ([0, 1], True), ([1, 0, 2], True)):
        targets = [TargetExpr(capi.PROJECT_KEY, g) for g in range(len(group))] + [TargetExpr(capi.COUNT)]
        ra = RelAlgExecutionUnit(descs, targets, [], group, max_groups_buffer_entry_guess=600_000, bigint_count=big, num_tuples=n)
        case = cases_mod.Case("idx_count_only", ra, frags)
        rs = flow._check(oracle, case, kernel_variant=2, scratch_bytes=6 << 20, flags=capi.OPT_NO_IDX_PACK)
        assert rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.variant == 6, (group, big, rs.report.kernel_name)
        assert rs.report.n_launches >= 2, rs.report.n_launches
        assert len(group) > 1 or rs.report.spilled_rows > 0, rs.report.spilled_rows      # (one key: the hot entry's runs overflow)
        # the 2-byte word (the entry's index inside its partition alone) where a partition has <= 65 536 entries
        rs = flow._check(oracle, case, kernel_variant=2, scratch_bytes=6 << 20)
        assert rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.variant in (6, 8), (group, big, rs.report.variant)
        assert len(group) > 1 or rs.report.variant == 8


def test_idx_partitioned_family_reports_a_key_outside_its_range(sim, oracle):
    from heavydb_amd.executor import Executor, ExpressionRange, FetchResult, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    n, card = 100_000, 80_000
    key = (np.arange(n) % card + 1).astype(np.int32)
    key[777] = card + 5      # outside the declared range: the reference would write past the table; here error 3
    val = np.ones(n, np.int32)
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 1, card)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 1, 1))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 1)], [], [0])
    case = cases_mod.Case("idx_bad_key", ra, [[key, val]])
    with pytest.raises(capi.Mi355qError) as ei:
        Executor(0).executeWorkUnit(ra, flow._fetch_result(case), allow_retry=False, kernel_variant=2)
    assert ei.value.code == capi.ERR_OUT_OF_SLOTS


# ---- GROUP BY CAST(int column AS DOUBLE | FLOAT): the step on the integer column + k_cast_key_emit -------------------------
@pytest.mark.parametrize("name", ["BH001", "BH002", "BH004", "BH005", "MSBS001", "MSBS002", "MSBS003"])
def test_cast_key_route_on_the_benchmark_shapes(sim, oracle, name):
    """BaselineHash / MultiStep-BaselineHash shapes: the key is CAST(x AS DOUBLE | FLOAT) of a plain integer column.  The
    step runs grouped by the integer column (perfect hash: whatever family that shape takes) and its entries are re-keyed
    with the cast value into the baseline table of the stated plan — same groups, same slots as the oracle's walk over the
    stated plan.  kernel_variant 2 = the large-input members on a small input."""
    case = flow._refbench_case(oracle, name, 150_003, 120_000)
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs is not None and rs.report.n_launches >= 2, rs.report.n_launches     # the inner step + the emit
    planned = flow._check(oracle, case, kernel_variant=0)                         # 150 K rows: below the route's threshold
    assert np.array_equal(planned.getStorage(), rs.getStorage()) or True          # (slot order may differ: both held to the oracle)


def test_cast_key_route_is_named_by_explain(sim):
    r = _explain("BH003", 1_000_000_000)
    assert "k_cast_key_emit" in r and "k_project" not in r, r
    r = _explain("MSBS004", 1_000_000_000)    # the other expression (x10 + 1): from the column's aggregates, no projection either
    assert "k_cast_key_emit" in r and "column + literal" in r and "k_project" not in r, r


def test_cast_key_route_with_null_keys_and_float_collisions(sim, oracle):
    """NULL keys (-> the NULL of the cast's type as the group key) and integers beyond 2^24 that round to the same FLOAT
    (two entries of the integer-keyed table merge into one group through the reduce rule)."""
    from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(5)
    n = 40_000
    for to, base in ((capi.DOUBLE, 0), (capi.FLOAT, (1 << 24) - 40)):
        key = (base + rng.integers(0, 90, n)).astype(np.int32)
        key[rng.random(n) < 0.05] = np.iinfo(np.int32).min
        val = rng.integers(-1000, 1000, n).astype(np.int64)
        val[rng.random(n) < 0.1] = -2**63
        descs = [InputColDescriptor(capi.INT32, True, ExpressionRange(True, base, base + 89, True)),
                 InputColDescriptor(capi.INT64, True, ExpressionRange(True, -1000, 999, True))]
        e = Expr.col(0).cast(to).with_range(ExpressionRange(True, 0, 0, True, float(base), float(base + 89)))
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1),
                                         TargetExpr(capi.MIN, 1), TargetExpr(capi.AVG, 1), TargetExpr(capi.COUNT, 1)],
                                 [], [2], max_groups_buffer_entry_guess=4096, exprs=[e], num_tuples=n)
        case = cases_mod.Case("cast_key_nulls", ra, [[key[:15_001], val[:15_001]], [key[15_001:], val[15_001:]]])
        rs = flow._check(oracle, case, kernel_variant=2)
        assert rs is not None and rs.report.n_launches >= 2, rs.report.n_launches


# ---- baseline steps over several ranged INT keys: a perfect-hash twin (index-partitioned family) + k_perfect_twin_emit --------
@pytest.mark.parametrize("name", ["PHM006", "MSPHM005", "MSPHM007"])
def test_perfect_twin_route_on_the_benchmark_shapes(sim, oracle, name):
    """PerfectHashMultiCol / MultiStep shapes whose key combinations exceed g_baseline_groupby_threshold (baseline layout in
    the reference, GroupByAndAggregate.cpp:232-365): the step runs on a library-owned perfect-hash table over the product of
    the key ranges and its live entries are re-keyed into the baseline table of the stated plan.  kernel_variant 2 = the
    large-input members on a small input."""
    case = flow._refbench_case(oracle, name, 150_003, 120_000)
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs is not None
    assert rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.n_launches >= 2, (rs.report.kernel_name, rs.report.n_launches)


def test_perfect_twin_route_is_named_by_explain(sim):
    for name in ("PHM006", "MSPHM005", "MSPHM007"):
        r = _explain(name, 1_000_000_000)
        assert "k_perfect_twin_emit" in r and r.endswith("k_idx_scatter + k_idx_aggregate"), (name, r)


def test_perfect_twin_route_with_null_keys(sim, oracle):
    """NULL keys in both key columns (translated to max + 1 in the twin, back to the column's NULL in the baseline key),
    nullable values, a table too small for the groups (out of slots, as the plain route reports it)"""
    from heavydb_amd.executor import Executor, ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(8)
    n = 60_000
    k0 = rng.integers(-700, 700, n).astype(np.int32)
    k1 = rng.integers(0, 900, n).astype(np.int32)
    k0[rng.random(n) < 0.03] = np.iinfo(np.int32).min
    k1[rng.random(n) < 0.03] = np.iinfo(np.int32).min
    val = rng.integers(-1000, 1000, n).astype(np.int32)
    val[rng.random(n) < 0.1] = np.iinfo(np.int32).min
    descs = [InputColDescriptor(capi.INT32, True, ExpressionRange(True, -700, 699, True)),
             InputColDescriptor(capi.INT32, True, ExpressionRange(True, 0, 899, True)),
             InputColDescriptor(capi.INT32, True, ExpressionRange(True, -1000, 999, True))]
    targets = [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.PROJECT_KEY, 1), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 2),
               TargetExpr(capi.MIN, 2), TargetExpr(capi.MAX, 2), TargetExpr(capi.AVG, 2), TargetExpr(capi.COUNT, 2)]
    ra = RelAlgExecutionUnit(descs, targets, [], [0, 1], max_groups_buffer_entry_guess=131072, num_tuples=n)   # 1401 x 901 > 1 M
    case = cases_mod.Case("twin_nulls", ra, [[k0[:20_001], k1[:20_001], val[:20_001]], [k0[20_001:], k1[20_001:], val[20_001:]]])
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs is not None and rs.report.kernel_name.decode() == "k_idx_scatter", rs.report.kernel_name
    ra.max_groups_buffer_entry_guess = 16384       # ~58 K groups do not fit
    with pytest.raises(Exception) as ei:
        Executor(0).executeWorkUnit(ra, flow._fetch_result(case), allow_retry=False, kernel_variant=2)
    assert "-" in str(ei.value) or "slots" in str(ei.value).lower(), str(ei.value)


# ---- aggregates of `column + literal` from the column's aggregates (execute_shifted_args) -----------------------------------
@pytest.mark.parametrize("name", ["MSPHS001", "MSPHS003", "MSPHS005", "MSPHS010", "MSPHS012", "MSPHM001", "MSPHM003", "MSPHM005",
                                  "MSBS001", "MSBS003"])
def test_shifted_argument_route_on_the_benchmark_shapes(sim, oracle, name):
    """MultiStep shapes: MAX(x10 + 1) / SUM(x10 + 1) next to MAX(x10).  The derived plan aggregates x10 once, adds COUNT(x10)
    for the sum, and the literal is added while the derived table is copied into the stated layout — no projected column.
    kernel_variant 2 = the large-input members on a small input."""
    case = flow._refbench_case(oracle, name, 150_003, 120_000)
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs is not None and rs.report.n_launches >= 2, rs.report.n_launches


def test_shifted_argument_route_is_named_by_explain(sim):
    for name in ("MSPHS001", "MSPHS004", "MSPHS011", "MSPHM002", "MSPHM005", "MSBS002"):
        r = _explain(name, 1_000_000_000)
        assert "column + literal" in r and "k_project" not in r, (name, r)


def test_shifted_arguments_with_nulls_and_a_range_that_could_overflow(sim, oracle):
    from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(21)
    n = 50_000
    key = rng.integers(0, 70_000, n).astype(np.int32)
    v32 = rng.integers(-500, 500, n).astype(np.int32)
    v32[rng.random(n) < 0.2] = np.iinfo(np.int32).min
    v64 = rng.integers(-10**12, 10**12, n).astype(np.int64)
    v64[rng.random(n) < 0.2] = np.iinfo(np.int64).min
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 69_999)),
             InputColDescriptor(capi.INT32, True, ExpressionRange(True, -500, 499, True)),
             InputColDescriptor(capi.INT64, True, ExpressionRange(True, -10**12, 10**12, True))]
    e0 = Expr.col(1).add(Expr.lit(capi.INT32, 7), capi.INT32).with_range(ExpressionRange(True, -493, 506, True))
    e1 = Expr.col(2).sub(Expr.lit(capi.INT64, 1000), capi.INT64).with_range(ExpressionRange(True, -10**12 - 1000, 10**12 - 1000, True))
    for targets in ([TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 3), TargetExpr(capi.MIN, 3), TargetExpr(capi.MAX, 1),
                     TargetExpr(capi.AVG, 4), TargetExpr(capi.COUNT, 3)],
                    [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.MAX, 4), TargetExpr(capi.SUM, 4),
                     TargetExpr(capi.AVG, 3), TargetExpr(capi.MIN, 2)]):
        ra = RelAlgExecutionUnit(descs, targets, [], [0], exprs=[e0, e1], num_tuples=n)
        case = cases_mod.Case("shifted_nulls", ra, [[key[:17_003], v32[:17_003], v64[:17_003]], [key[17_003:], v32[17_003:], v64[17_003:]]])
        rs = flow._check(oracle, case, kernel_variant=2)
        assert rs is not None and rs.report.n_launches >= 2, rs.report.n_launches
    # INT32 values up to 2^31 - 3: + 7 may overflow, the expression stays projected (and raises the reference's error 7)
    big = v32.copy()
    big[5] = 2**31 - 3
    descs2 = [descs[0], InputColDescriptor(capi.INT32, True, ExpressionRange(True, -500, 2**31 - 3, True)), descs[2]]
    ra = RelAlgExecutionUnit(descs2, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 3)], [], [0],
                             exprs=[Expr.col(1).add(Expr.lit(capi.INT32, 7), capi.INT32).with_range(ExpressionRange(True, -493, 2**31 - 1, True)), e1],
                             num_tuples=n)
    case = cases_mod.Case("shifted_overflow", ra, [[key, big, v64]], expect_error=7)
    flow._check(oracle, case, kernel_variant=2)     # the oracle's code (7) and the product's must agree


# ---- grouped joins on one-to-one tables: k_join_gather + the step without a join (execute_join_gather) -------------------------
def test_grouped_join_gather_route_in_several_passes(sim, oracle):
    """SELECT f.g, COUNT(*), SUM(d.w), MIN(d.x), AVG(d.x) FROM f [LEFT] JOIN d ON f.k = d.k GROUP BY f.g: the probe is its own pass
    (inner columns + a matched flag as dense outer columns), the derived step has no join and the stated layout.  Nullable inner
    column with NULLs, keys that miss, NULL join keys, pass_rows that cuts the input into three passes (folded with the reduce rule)."""
    from heavydb_amd.executor import Executor, ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(77)
    m, n = 5_000, 90_000
    dim_k = rng.permutation(m).astype(np.int64)
    dim_w = rng.integers(-1000, 1000, m).astype(np.int64)
    dim_x = rng.integers(-50, 50, m).astype(np.int32)
    dim_x[rng.random(m) < 0.2] = np.iinfo(np.int32).min
    fk = rng.integers(-200, m + 200, n).astype(np.int64)
    fk[rng.random(n) < 0.03] = np.iinfo(np.int64).min
    fg = rng.integers(0, 40, n).astype(np.int32)
    fv = rng.integers(-10**6, 10**6, n).astype(np.int64)
    fdescs = [InputColDescriptor(capi.INT64, True, ExpressionRange(True, -200, m + 199, True)),
              InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 39)),
              InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**6, 10**6))]
    idescs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, m - 1)),
              InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, 999)),
              InputColDescriptor(capi.INT32, True, ExpressionRange(True, -50, 49, True))]
    cuts = [0, 30_001, 60_002, n]
    frags = [[fk[a:b], fg[a:b], fv[a:b]] for a, b in zip(cuts[:-1], cuts[1:])]
    for kind in (capi.JOIN_INNER, capi.JOIN_LEFT):
        for pb in (False, True):
            ra = RelAlgExecutionUnit(fdescs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1, 1),
                                              TargetExpr(capi.MIN, 2, 1), TargetExpr(capi.AVG, 2, 1), TargetExpr(capi.SUM, 2),
                                              TargetExpr(capi.COUNT, 2, 1)], [], [1],
                                     inner_col_descs=idescs, join_outer_col=0, join_kind=kind)
            case = cases_mod.Case("grouped_join_gather", ra, frags, [dim_k, dim_w, dim_x], dim_k, capi.INT64,
                                  ExpressionRange(True, 0, m - 1), pb)
            rs = flow._check(oracle, case, kernel_variant=2, pass_rows=31_000)
            assert rs is not None and rs.report.n_launches >= 6, rs.report.n_launches     # three passes: gather + step each


def test_grouped_join_gather_route_is_named_by_explain(sim, oracle):
    """taken for one-to-one tables when the row kernel would not have its per-workgroup LDS copy (a table beyond 64 KB); not for
    one-to-many tables (row multiplicity), not for small tables (probe and update in one pass there: measured equal or faster)"""
    import copy
    from heavydb_amd.executor import Executor, ExpressionRange
    for name, groups, taken in (("join_left_perfect_1to1_groupby", 20_000, True), ("join_keyed_groupby", 20_000, True),
                                ("join_keyed_groupby", 30, False), ("join_1n_perfect_groupby_int32_key", 20_000, False)):
        case = next(c for c in CASES if c.name == name)
        ra = copy.copy(case.ra)
        ra.input_col_descs = list(ra.input_col_descs)
        g = ra.groupby_exprs[0]
        d = copy.copy(ra.input_col_descs[g])
        d.range = ExpressionRange(True, 0, groups - 1, d.range.has_nulls)
        ra.input_col_descs[g] = d
        hj, keep = flow._build_join(case)
        ra.join_table = hj
        r = Executor(0).explain(ra, [250_000_000] * 4)
        assert r.startswith("k_join_gather") == taken, (name, groups, r)


# ---- whole steps over random expressions (comparisons, CASE, / %, casts) through the real kernels ------------------------------
def test_random_expression_steps_through_the_real_kernels(sim, oracle):
    """Random grouped / non-grouped steps whose aggregate arguments, filter and (sometimes) group key are random well-typed
    expression programs (tests/test_expr._random_expr: every micro-op, nested CASE, NULL literals), planned (variant 0) or
    through the large-input routes (variant 2: shifted arguments, cast keys, the projection passes): table or error code must
    equal the oracle's walk over the stated plan."""
    import os
    from heavydb_amd.executor import Executor, Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    from tests.test_expr import _random_bool, _random_expr, _stack_depth
    from tests.cases import expr_range
    rng = np.random.default_rng(int(os.environ.get("MI355Q_FUZZ_SEED", "4711")))
    iters = int(os.environ.get("MI355Q_FUZZ_ITERS", "60"))
    NPT = {capi.INT32: np.int32, capi.INT64: np.int64, capi.DOUBLE: np.float64}
    ran = errors = 0
    for it in range(iters):
        n = int(rng.integers(40, 4000))
        descs, cols = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 49))], [rng.integers(0, 50, n).astype(np.int32)]
        for t in (capi.INT32, capi.INT32, capi.INT64, capi.INT64, capi.DOUBLE):
            nullable = bool(rng.integers(0, 2))
            if t == capi.DOUBLE:
                v = rng.uniform(-50, 50, n)
                v[rng.random(n) < 0.1] = 0.0
                if nullable:
                    v[rng.random(n) < 0.15] = np.finfo(np.float64).tiny
                r = ExpressionRange(True, 0, 0, nullable, -50.0, 50.0)
            else:
                lo, hi = (-1000, 1000) if rng.integers(0, 2) else (0, 5)
                v = rng.integers(lo, hi + 1, n).astype(NPT[t])
                if nullable:
                    v[rng.random(n) < 0.15] = np.iinfo(NPT[t]).min
                r = ExpressionRange(True, lo, hi, nullable)
            descs.append(InputColDescriptor(t, nullable, r))
            cols.append(np.ascontiguousarray(v.astype(NPT[t])))
        exprs = []
        for _ in range(int(rng.integers(1, 4))):
            e = _random_expr(rng, descs, int(rng.choice([capi.INT32, capi.INT64, capi.DOUBLE])), int(rng.integers(1, 4)), big=False)
            if e is not None and len(e.nodes) <= capi.MAX_EXPR_NODES and _stack_depth(e) <= capi.MAX_EXPR_STACK and any(nd.op == capi.EX_COL for nd in e.nodes):
                exprs.append(e)
        if not exprs:
            continue
        nc = len(descs)
        # ... an expression that reads the value of an earlier one, and a BOOLEAN program used as the filter
        if len(exprs) < capi.MAX_EXPRS and rng.integers(0, 3) == 0:
            j = int(rng.integers(0, len(exprs)))
            tj = exprs[j].result(descs, exprs[:j])[0]
            ref = Expr.col(nc + j)
            exprs.append(ref.is_null().logical_not().cast(capi.INT32) if rng.integers(0, 2) else
                         ref.add(Expr.lit(tj, 1), tj) if tj == capi.DOUBLE else ref.cast(capi.DOUBLE).mul(Expr.lit(capi.DOUBLE, 0.5), capi.DOUBLE))
        filt = None
        if len(exprs) < capi.MAX_EXPRS and rng.integers(0, 3) == 0:
            b = _random_bool(rng, descs, int(rng.integers(1, 3)), big=False)
            if b is not None and len(b.nodes) <= capi.MAX_EXPR_NODES and _stack_depth(b) <= capi.MAX_EXPR_STACK:
                filt = len(exprs)
                exprs.append(b)
        try:
            exprs = [e.with_range(expr_range(e, descs, [cols], exprs[:i])) for i, e in enumerate(exprs)]
        except Exception:
            continue      # (the numpy range helper does not model every program: inf / nan corners)
        grouped = bool(rng.integers(0, 4))
        targets = [TargetExpr(capi.PROJECT_KEY)] if grouped else []
        targets.append(TargetExpr(capi.COUNT))
        for k in range(len(exprs)):
            if k != filt:
                targets.append(TargetExpr(int(rng.choice([capi.SUM, capi.MIN, capi.MAX, capi.AVG, capi.COUNT])), nc + k))
        targets = targets[:6]
        quals = [Qual(2, capi.GE, -900)] if rng.integers(0, 3) == 0 else []
        if filt is not None:
            quals.append(Qual(nc + filt, capi.EQ, 1))
        ra = RelAlgExecutionUnit(descs, targets, quals, [0] if grouped else [], exprs=exprs, num_tuples=n)
        cut = (n // 2) & ~3
        case = cases_mod.Case("fuzz_expr_step", ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]])
        try:
            q, want, code = oracle.execute(ra.to_plan(), case.frags, n_threads=1)
        except capi.Mi355qError:
            continue
        if code:
            case.expect_error = code
            errors += 1
        try:
            flow._check(oracle, case, kernel_variant=int(rng.choice([0, 2])))
        except AssertionError:
            print("iteration", it, "exprs", [[(nd.op, nd.type, nd.arg, nd.ilit, nd.flit, nd.null_lit) for nd in e.nodes] for e in exprs],
                  "targets", [(t.agg, t.col) for t in targets], "grouped", grouped)
            raise
        ran += 1
    assert ran > iters // 2, (ran, errors)


def test_random_perfect_twin_steps(sim, oracle):
    """Random baseline steps over 2 - 3 ranged INT32 keys whose combinations exceed the perfect-hash threshold (1.0 - 2.2 M),
    random nullability / NULLs in keys and values, 1 - 3 INT32 value columns, random aggregates: the twin route (variant 2)
    against the oracle's walk over the stated (baseline) plan."""
    import os
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(int(os.environ.get("MI355Q_FUZZ_SEED", "99")))
    taken = 0
    for it in range(int(os.environ.get("MI355Q_FUZZ_ITERS", "3"))):
        n = int(rng.integers(2_000, 40_000))
        nk = int(rng.integers(2, 4))
        cards = [int(rng.integers(900, 1500)), int(rng.integers(900, 1500))] if nk == 2 else \
                [int(rng.integers(90, 130)), int(rng.integers(90, 130)), int(rng.integers(100, 130))]
        descs, cols = [], []
        for c in cards:
            lo = int(rng.integers(-500, 500))
            nullable = bool(rng.integers(0, 2))
            has_nulls = nullable and bool(rng.integers(0, 2))
            v = rng.integers(lo, lo + c, n).astype(np.int32)
            if has_nulls:
                v[rng.random(n) < 0.05] = np.iinfo(np.int32).min
            descs.append(InputColDescriptor(capi.INT32, nullable, ExpressionRange(True, lo, lo + c - 1, has_nulls)))
            cols.append(v)
        nv = int(rng.integers(1, 4))
        for _ in range(nv):
            nullable = bool(rng.integers(0, 2))
            v = rng.integers(-1000, 1000, n).astype(np.int32)
            if nullable:
                v[rng.random(n) < 0.1] = np.iinfo(np.int32).min
            descs.append(InputColDescriptor(capi.INT32, nullable, ExpressionRange(True, -1000, 999, nullable)))
            cols.append(v)
        targets = [TargetExpr(capi.PROJECT_KEY, g) for g in range(nk) if rng.integers(0, 4)] + [TargetExpr(capi.COUNT)]
        for j in range(nv):
            targets.append(TargetExpr(int(rng.choice([capi.SUM, capi.MIN, capi.MAX, capi.AVG, capi.COUNT])), nk + j))
        targets = targets[:capi.MAX_TARGETS] if hasattr(capi, "MAX_TARGETS") else targets[:8]
        ra = RelAlgExecutionUnit(descs, targets, [], list(range(nk)), max_groups_buffer_entry_guess=131072, num_tuples=n)
        cut = (n // 3) & ~3
        case = cases_mod.Case("fuzz_twin", ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]])
        rs = flow._check(oracle, case, kernel_variant=2)
        assert rs is not None
        taken += rs.report.kernel_name.decode() == "k_idx_scatter"
    assert taken >= 1, taken


# ---- baseline keys on a lattice (key = min + stride x i): stride from the first fragment, verified lattice indices, a perfect twin ----
LATTICE_SHAPES = ["BH008", "BH009", "BH010", "MSBS006", "MSBS007"]


@pytest.mark.parametrize("name", LATTICE_SHAPES)
def test_lattice_key_route_on_the_benchmark_shapes(sim, oracle, name):
    """BaselineHash / MultiStep shapes over the BIGINT stride columns (multiples of 10 000): grouped by the lattice index on a
    perfect-hash twin (k_idx_scatter: one exchange), re-keyed into the stated baseline table.  kernel_variant 2 = the
    large-input members on a small input."""
    case = flow._refbench_case(oracle, name, 150_003, 120_000)
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs is not None and rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.n_launches >= 3, \
        (rs.report.kernel_name, rs.report.n_launches)


def test_lattice_key_route_nulls_offsets_passes_and_a_key_off_the_lattice(sim, oracle):
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(31)
    n = 90_000
    k0 = (7 + 10_000 * rng.integers(0, 70_000, n)).astype(np.int64)             # BIGINT, stride 10 000, offset 7
    k0[rng.random(n) < 0.02] = np.iinfo(np.int64).min
    k1 = (-3_000_000 + 250_000 * rng.integers(0, 9, n)).astype(np.int32)         # INT over a wide range, stride 250 000
    v0 = rng.integers(-1000, 1000, n).astype(np.int32)
    v0[rng.random(n) < 0.1] = np.iinfo(np.int32).min
    v1 = rng.integers(0, 10, n).astype(np.int32)
    descs = [InputColDescriptor(capi.INT64, True, ExpressionRange(True, 7, 7 + 10_000 * 69_999, True)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, -3_000_000, -1_000_000)),
             InputColDescriptor(capi.INT32, True, ExpressionRange(True, -1000, 999, True)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 9))]
    cuts = [0, 30_001, 60_002, n]
    frags = [[k0[a:b], k1[a:b], v0[a:b], v1[a:b]] for a, b in zip(cuts[:-1], cuts[1:])]
    for group, targets in (([0], [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 2), TargetExpr(capi.MIN, 2)]),
                           ([0, 1], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.PROJECT_KEY, 1), TargetExpr(capi.COUNT),
                                     TargetExpr(capi.AVG, 2), TargetExpr(capi.MAX, 3), TargetExpr(capi.COUNT, 2)]),
                           ([1, 0], [TargetExpr(capi.PROJECT_KEY, 1), TargetExpr(capi.COUNT)])):
        ra = RelAlgExecutionUnit(descs, targets, [], group, max_groups_buffer_entry_guess=262_144, num_tuples=n)
        case = cases_mod.Case("lattice", ra, frags)
        rs = flow._check(oracle, case, kernel_variant=2, pass_rows=31_000)
        assert rs is not None and rs.report.kernel_name.decode() == "k_idx_scatter" and rs.report.n_launches >= 7, \
            (group, rs.report.kernel_name, rs.report.n_launches)      # three passes (lattice indices + twin step each) + the emit
    # one key off the lattice, in the LAST fragment: the route is given up after its check, the plain route answers
    bad = k0.copy()
    bad[n - 5] = 7 + 10_000 * 123 + 1
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 2)], [], [0],
                             max_groups_buffer_entry_guess=262_144, num_tuples=n)
    case = cases_mod.Case("off_lattice", ra, [[bad[a:b], k1[a:b], v0[a:b], v1[a:b]] for a, b in zip(cuts[:-1], cuts[1:])])
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs is not None and rs.report.kernel_name.decode() != "k_idx_scatter", rs.report.kernel_name


# ---- filters compiled at plan time (boolfilter.h): atoms + truth table evaluated by the consuming kernel
def _bool_filter_cases(oracle, n_rows=30_000, null_every=7):
    """the shapes of tools/bool_filter_bench.py (+ a few of the reference's own WHERE clauses) over a small table:
    g (1 000 groups), v, a, b uniform in [0, 1 M), c nullable"""
    from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    from tests.cases import Case
    rng = np.random.default_rng(77)
    g = rng.integers(0, 1000, n_rows).astype(np.int32)
    v = rng.integers(0, 1_000_000, n_rows).astype(np.int32)
    a = rng.integers(0, 1_000_000, n_rows).astype(np.int32)
    b = rng.integers(0, 1_000_000, n_rows).astype(np.int64)   # (an INT64 filter column beside the INT32 ones)
    c = rng.integers(0, 1_000_000, n_rows).astype(np.int32)
    c[::null_every] = -2**31
    I32, I64 = capi.INT32, capi.INT64
    descs = [InputColDescriptor(I32, False, ExpressionRange(True, 0, 999)), InputColDescriptor(I32, False, ExpressionRange(True, 0, 999_999)),
             InputColDescriptor(I32, False, ExpressionRange(True, 0, 999_999)), InputColDescriptor(I64, False, ExpressionRange(True, 0, 999_999)),
             InputColDescriptor(I32, True, ExpressionRange(True, 0, 999_999, True))]
    C_, L = Expr.col, Expr.lit
    a_lt = C_(2).cmp(capi.EX_LT, L(I32, 500_000))
    b_gt = C_(3).cmp(capi.EX_GT, L(I64, 250_000))
    c_lt = C_(4).cmp(capi.EX_LT, L(I32, 100_000))
    band = lambda col, t, lo, hi: C_(col).cmp(capi.EX_GT, L(t, lo)).logical(capi.EX_AND, C_(col).cmp(capi.EX_LT, L(t, hi)))
    nc = 5
    shapes = [
        ("and_in_or", [a_lt.logical(capi.EX_AND, b_gt).logical(capi.EX_OR, C_(4).is_null())], [Qual(nc, capi.EQ, 1)]),
        ("not_or", [a_lt.logical(capi.EX_OR, b_gt).logical_not()], [Qual(nc, capi.EQ, 1)]),
        ("composed", [band(2, I32, 100_000, 200_000), band(3, I64, 300_000, 400_000),
                      band(4, I32, 500_000, 600_000).logical(capi.EX_OR, C_(nc)).logical(capi.EX_OR, C_(nc + 1))], [Qual(nc + 2, capi.EQ, 1)]),
        # NULL-aware: NOT over a comparison of the nullable column (NULL stays NULL: not TRUE), mirrored literal, a plain qual beside it
        ("not_null_cmp_and_plain_qual", [c_lt.logical_not().logical(capi.EX_OR, L(I32, 900_000).cmp(capi.EX_LT, C_(2)))],
         [Qual(nc, capi.EQ, 1), Qual(1, capi.GE, 1000)]),
        # the short-circuit forms differ from the plain ones on NULL: NULL AND FALSE = NULL (not TRUE either way), NULL OR TRUE
        ("short_circuit_or", [c_lt.logical(capi.EX_OR, a_lt, True)], [Qual(nc, capi.EQ, 1)]),
        ("is_not_null_and_in_list", [C_(2).cmp(capi.EX_EQ, L(I32, int(a[5]))).logical(capi.EX_OR, C_(2).cmp(capi.EX_EQ, L(I32, int(a[9]))))
                                     .logical(capi.EX_OR, C_(4).is_null().logical_not().logical(capi.EX_AND, c_lt))], [Qual(nc, capi.EQ, 1)]),
    ]
    out = []
    for name, exprs, quals in shapes:
        xs = [e.with_range(ExpressionRange(True, 0, 1, True)) for e in exprs]
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1)], quals, [0], exprs=xs,
                                 num_tuples=n_rows)
        half = n_rows // 2 // 4 * 4
        out.append(Case(name, ra, [[x[:half] for x in (g, v, a, b, c)], [x[half:] for x in (g, v, a, b, c)]]))
    return out


@pytest.mark.parametrize("idx", range(6), ids=["and_in_or", "not_or", "composed", "not_null_cmp_and_plain_qual", "short_circuit_or",
                                                "is_not_null_and_in_list"])
def test_compiled_bool_filters_in_the_lds_groupby(sim, oracle, idx):
    """BOOLEAN filters of comparisons with literals no longer take the interpreter pass: the route is the LDS group-by alone,
    the result the oracle's (which evaluates the expression programs node by node)"""
    from heavydb_amd.executor import Executor
    case = _bool_filter_cases(oracle)[idx]
    rs = flow._check(oracle, case)
    assert rs is not None and rs.report.kernel_name.decode() == "k_groupby_lds", rs.report.kernel_name
    route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags])
    assert "filter compiled" in route and "k_project" not in route and "k_groupby_lds" in route, route


# ---- round 6: PROGRAM atoms (regprog.h) — leaves with arithmetic, two columns, DOUBLE operands compiled into two-register
# programs of typed steps; the atom's fourth state (ERROR) and the "which atom raises" table
def _prog_atom_table(n=30_011, seed=5, zero_b=True, nulls=True):
    """g (50 groups), v INT32, a INT32 in [-1000, 1000), b INT32 in [-5, 6) (zeros!), c nullable INT32, w INT64, d DOUBLE (nullable)"""
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 50, n).astype(np.int32)
    v = rng.integers(-1000, 1000, n).astype(np.int32)
    a = rng.integers(-1000, 1000, n).astype(np.int32)
    b = rng.integers(-5, 6, n).astype(np.int32)
    if not zero_b:
        b[b == 0] = 3
    c = rng.integers(0, 100, n).astype(np.int32)
    w = rng.integers(-10**12, 10**12, n).astype(np.int64)
    d = rng.uniform(-10, 10, n)
    if nulls:
        c[rng.random(n) < 0.15] = -2**31
        d[rng.random(n) < 0.1] = np.finfo(np.float64).tiny
    I32, I64, F64 = capi.INT32, capi.INT64, capi.DOUBLE
    R = ExpressionRange
    descs = [InputColDescriptor(I32, False, R(True, 0, 49)), InputColDescriptor(I32, False, R(True, -1000, 999)),
             InputColDescriptor(I32, False, R(True, -1000, 999)), InputColDescriptor(I32, False, R(True, -5, 5)),
             InputColDescriptor(I32, nulls, R(True, 0, 99, nulls)), InputColDescriptor(I64, False, R(True, -10**12, 10**12)),
             InputColDescriptor(F64, nulls, R(True, 0, 0, nulls, -10.0, 10.0))]
    return descs, [g, v, a, b, c, w, d]


def _prog_atom_shapes():
    from heavydb_amd.executor import Expr
    I32, I64, F64 = capi.INT32, capi.INT64, capi.DOUBLE
    C_, L = Expr.col, Expr.lit
    A, B, Cn, W, D, V = 2, 3, 4, 5, 6, 1
    sc = True
    return {
        # the reference's deferred qual: the division is never evaluated where b = 0 (short-circuit AND)
        "guarded_div": (C_(B).cmp(capi.EX_NE, L(I32, 0)).logical(capi.EX_AND, C_(A).div(C_(B), I32).cmp(capi.EX_GT, L(I32, 3)), sc), None),
        # unguarded: rows with b = 0 raise error 1
        "unguarded_div": (C_(A).div(C_(B), I32).cmp(capi.EX_GT, L(I32, 3)), capi.ERR_DIV_BY_ZERO),
        # the plain AND evaluates both sides: the guard does not help
        "plain_and_does_not_guard": (C_(B).cmp(capi.EX_NE, L(I32, 0)).logical(capi.EX_AND, C_(A).div(C_(B), I32).cmp(capi.EX_GT, L(I32, 3))),
                                     capi.ERR_DIV_BY_ZERO),
        "sum_of_two_columns": (C_(A).add(C_(V), I32).cmp(capi.EX_GT, L(I32, 100)), None),
        "column_vs_column": (C_(A).cmp(capi.EX_LT, C_(V)), None),
        "nullable_column_vs_column": (C_(Cn).cmp(capi.EX_GE, C_(B)).logical(capi.EX_OR, C_(Cn).is_null()), None),
        "affine": (C_(A).mul(L(I32, 3), I32).sub(L(I32, 7), I32).cmp(capi.EX_LE, L(I32, 500)), None),
        "overflow_raises": (C_(A).mul(L(I32, 5_000_000), I32).cmp(capi.EX_GT, L(I32, 0)), capi.ERR_OVERFLOW_OR_UNDERFLOW),
        "widened_product": (C_(A).cast(I64).mul(L(I64, 5_000_000), I64).cmp(capi.EX_GT, C_(W)), None),
        "modulo": (C_(A).mod(L(I32, 7), I32).cmp(capi.EX_EQ, L(I32, 3)).logical(capi.EX_OR, C_(B).cmp(capi.EX_GT, L(I32, 3))), None),
        "double_column": (C_(D).cmp(capi.EX_LT, L(F64, 2.5)).logical(capi.EX_AND, C_(A).cmp(capi.EX_GT, L(I32, -500))), None),
        "double_arithmetic": (C_(D).mul(L(F64, 2.0), F64).cmp(capi.EX_GT, C_(A).cast(F64)), None),
        "double_division_guarded": (C_(D).cmp(capi.EX_NE, L(F64, 0.0)).logical(capi.EX_AND, L(F64, 1.0).div(C_(D), F64).cmp(capi.EX_LT, L(F64, 0.5)), sc), None),
        "is_null_of_a_sum": (C_(Cn).add(C_(A), I32).is_null().logical_not(), None),
        "not_over_program_atom": (C_(A).add(C_(V), I32).cmp(capi.EX_GT, L(I32, 100)).logical_not().logical(capi.EX_OR, C_(Cn).cmp(capi.EX_LT, L(I32, 10))), None),
        "uminus": (C_(A).neg(I32).cmp(capi.EX_GT, C_(V)), None),
        "two_program_atoms_short_circuit_or": (C_(B).cmp(capi.EX_EQ, L(I32, 0)).logical(capi.EX_OR, L(I32, 100).div(C_(B), I32).cmp(capi.EX_GT, C_(A)), sc)
                                               .logical(capi.EX_AND, C_(A).add(C_(V), I32).cmp(capi.EX_LT, L(I32, 900))), None),
    }


def _prog_atom_case(name, grouped=True, typed=True, quals=(), n=30_011):
    from heavydb_amd.executor import ExpressionRange, Qual, RelAlgExecutionUnit, TargetExpr
    from tests.cases import Case
    descs, cols = _prog_atom_table(n)
    e, err = _prog_atom_shapes()[name]
    nc = len(descs)
    targets = ([TargetExpr(capi.PROJECT_KEY)] if grouped else []) + [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1)] + \
              ([] if typed else [TargetExpr(capi.MIN, 5)])   # (an INT64 value column: the run-time-role member)
    ra = RelAlgExecutionUnit(descs, targets, [Qual(nc, capi.EQ, 1)] + list(quals), [0] if grouped else [],
                             exprs=[e.with_range(ExpressionRange(True, 0, 1, True))], max_groups_buffer_entry_guess=256, num_tuples=n)
    h = n // 2 // 4 * 4 + 4
    case = Case(name, ra, [[x[:h] for x in cols], [x[h:] for x in cols]])
    case.expect_error = err
    return case


@pytest.mark.parametrize("consumer", ["typed_lds", "generic_lds", "scan_agg"])
@pytest.mark.parametrize("name", list(_prog_atom_shapes()))
def test_program_atoms_in_compiled_filters(sim, oracle, name, consumer):
    """filters whose leaves hold arithmetic, two columns or DOUBLE operands are compiled into program atoms: the row-mask pre-pass
    (k_filter_mask) evaluates them, the consumer filters on one byte per row — no interpreter pass; values AND error codes as
    the oracle's node-by-node evaluation gives them"""
    from heavydb_amd.executor import Executor
    case = _prog_atom_case(name, grouped=consumer != "scan_agg", typed=consumer == "typed_lds")
    route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags])
    assert "filter compiled" in route and "k_project" not in route, route
    rs = flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_LDS_GENERIC_MEMBER if consumer == "generic_lds" else 0)
    if case.expect_error is None:
        kn = rs.report.kernel_name.decode()
        # (behind the row mask the step has ONE 1-byte qual: the planner may give a perfect-hash few-groups step to k_perfect_lds)
        assert kn == "k_scan_agg" if consumer == "scan_agg" else kn in ("k_groupby_lds", "k_perfect_lds"), kn
        lean = name in ("guarded_div", "sum_of_two_columns", "column_vs_column", "nullable_column_vs_column", "modulo",
                        "not_over_program_atom", "affine")
        if consumer == "typed_lds":
            assert rs.report.variant == 5 or kn == "k_perfect_lds", rs.report.variant
            # lean atoms (INT32 operands, one operation) are evaluated by the typed member itself; the rest by the row-mask pre-pass
            assert ("k_filter_mask" in route) == (not lean), (name, route)
            # ... and the pre-pass (both of its members) agrees
            flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_FILTER_PREPASS)
            flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_FILTER_PREPASS | capi.OPT_LDS_GENERIC_MEMBER)
        else:
            assert "k_filter_mask" in route, route   # the consumer filters on `mask = 1` (one byte per row)
        # the interpreter pass agrees
        flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_NO_COMPILED_FILTER)


@pytest.mark.parametrize("member", ["fused", "lean", "general"])
@pytest.mark.parametrize("nullable", [False, True], ids=["notnull", "nullable"])
@pytest.mark.parametrize("op", ["cmp2", "add", "sub", "mul", "div", "mod", "add_lit", "mul_lit", "div_lit", "mod_lit", "chain_mul_sub", "chain_add_div",
                                "chain_cols_mod"])
def test_pair_atoms_on_edge_values(sim, oracle, op, nullable, member):
    """the LEAN form of a program atom (boolfilter.h pair_eval: 32-bit values) against the program's own steps (the general
    member of the pre-pass, MI355Q_OPT_LDS_GENERIC_MEMBER) and the oracle, on every pair of INT32 edge values: first with the
    rows that raise (the code must be the oracle's), then without them (the groups must be the oracle's)"""
    from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    from tests.cases import Case
    I32 = capi.INT32
    edge = np.array([-2**31, -2**31 + 1, -65536, -7, -2, -1, 0, 1, 2, 3, 7, 46341, 65536, 2**31 - 2, 2**31 - 1], dtype=np.int64)
    a, b = [x.ravel() for x in np.meshgrid(edge, edge)]
    C_, L = Expr.col, lambda v: Expr.lit(I32, v)
    lit = {"add_lit": 5, "mul_lit": 65536, "div_lit": -1, "mod_lit": 7}.get(op)
    rhs = L(lit) if lit is not None else C_(2)
    base = op.split("_")[0]
    if base == "chain":   # ((a <op> b) <op2> literal) <cmp> literal: which rows raise is the oracle's to say (one code per shape)
        val = {"chain_mul_sub": C_(1).mul(L(3), I32).sub(L(7), I32), "chain_add_div": C_(1).add(C_(2), I32).div(L(3), I32),
               "chain_cols_mod": C_(1).sub(C_(2), I32).mod(L(-1), I32)}[op]
        from heavydb_amd.executor import ExpressionRange as _R, InputColDescriptor as _D, RelAlgExecutionUnit as _U, TargetExpr as _T
        probe = _U([_D(I32, nullable, _R(True, -2**31 + 1, 2**31 - 1, nullable))] * 2, [_T(capi.PROJECT, 2)],
                   exprs=[Expr(val.nodes[:]).with_range(_R(True, -2**31, 2**31 - 1, True))], max_groups_buffer_entry_guess=8)
        probe.exprs[0].nodes = [type(n)(n.op, n.type, n.arg - 1 if n.op == capi.EX_COL else n.arg, n.ilit, n.flit, n.null_lit) for n in val.nodes]
        raises = np.zeros(len(a), bool)
        for i in range(len(a)):   # one row at a time through the oracle: does the value expression raise?
            _, _, code_i = oracle.execute(probe.to_plan(), [[np.array([a[i]], np.int32), np.array([b[i]], np.int32)]])
            raises[i] = code_i > 0
    else:
        val = {"add": C_(1).add(rhs, I32), "sub": C_(1).sub(rhs, I32), "mul": C_(1).mul(rhs, I32), "div": C_(1).div(rhs, I32),
               "mod": C_(1).mod(rhs, I32)}.get(base)
    bb = np.full_like(b, lit) if lit is not None else b
    with np.errstate(all="ignore"):
        exact = {"add": a + bb, "sub": a - bb, "mul": a * bb}.get(base)
    nul = -2**31
    is_null = ((a == nul) | ((bb == nul) & (lit is None))) if nullable else np.zeros(len(a), bool)
    if base == "chain":
        pass
    elif base in ("add", "sub", "mul"):
        raises = ~is_null & ((exact > 2**31 - 1) | (exact < -2**31))
    elif base == "div":
        skip = (np.full(len(a), nullable) & ((a == nul) | (bb == nul)))
        raises = ~skip & (bb == 0)
    elif base == "mod":
        raises = bb == 0
    else:
        raises = np.zeros(len(a), bool)
    for cmp_op, k in ((capi.EX_GT, 3), (capi.EX_LE, -1), (capi.EX_EQ, nul)):
        e = C_(1).cmp(cmp_op, C_(2)) if op == "cmp2" else val.cmp(cmp_op, L(k))
        for keep, expect_err in ((np.ones(len(a), bool), None), (~raises, 0)):
            aa, b2 = a[keep].astype(np.int32), b[keep].astype(np.int32)
            if len(aa) < 8:
                continue
            rep = 5   # (a few quads per lane, a ragged end)
            aa, b2 = np.tile(aa, rep)[:-3], np.tile(b2, rep)[:-3]
            g = (np.arange(len(aa)) % 13).astype(np.int32)
            descs = [InputColDescriptor(I32, False, ExpressionRange(True, 0, 12)),
                     InputColDescriptor(I32, nullable, ExpressionRange(True, -2**31 + 1, 2**31 - 1, nullable)),
                     InputColDescriptor(I32, nullable, ExpressionRange(True, -2**31 + 1, 2**31 - 1, nullable))]
            ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT)], [Qual(3, capi.EQ, 1)], [0],
                                     exprs=[e.with_range(ExpressionRange(True, 0, 1, True))], max_groups_buffer_entry_guess=64, num_tuples=len(aa))
            h = len(aa) // 2 // 4 * 4
            case = Case(f"{op}", ra, [[x[:h] for x in (g, aa, b2)], [x[h:] for x in (g, aa, b2)]])
            q, want, code = oracle.execute(ra.to_plan(), case.frags, n_threads=2)
            if expect_err == 0:
                assert code == 0, (op, cmp_op, code)
            case.expect_error = code if code else None
            flow._check(oracle, case, kernel_variant=0, flags={"general": capi.OPT_LDS_GENERIC_MEMBER | capi.OPT_FILTER_PREPASS,
                                                               "lean": capi.OPT_FILTER_PREPASS, "fused": 0}[member])


@pytest.mark.parametrize("family,variant", [("k_part_scatter", 2), ("k_baseline_direct", 1), ("k_perfect_lds", 0)])
@pytest.mark.parametrize("name", ["guarded_div", "affine", "double_arithmetic", "unguarded_div"])
def test_program_atoms_through_the_mask_in_the_large_table_families(sim, oracle, name, family, variant):
    """the row mask (1 B/row, `mask = 1`) is a filter column the FastShape families take too (round 6: Quad<int8_t>): a filter
    with program atoms in front of the partitioned GROUP BY (the headline family), the direct baseline member and the
    perfect-hash LDS member — no interpreter pass, no 4-byte temporary column"""
    from heavydb_amd.executor import Executor
    case = _mask_large_case(name, family)
    err = case.expect_error
    rs = flow._check(oracle, case, kernel_variant=variant, flags=capi.OPT_FILTER_PREPASS)
    if err is None:
        assert rs.report.kernel_name.decode() == family, rs.report.kernel_name
        route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags], kernel_variant=variant, flags=capi.OPT_FILTER_PREPASS)
        assert "k_filter_mask" in route and "k_project" not in route, route


def _mask_large_case(name, family, n=40_003, n_groups=3000):
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    from tests.cases import Case
    descs, cols = _prog_atom_table(n, seed=9)
    e, err = _prog_atom_shapes()[name]
    rng = np.random.default_rng(1)
    if family == "k_perfect_lds":   # SELECT g, SUM(w) ... GROUP BY g: one INT64 value column, 50 groups
        key_desc, key = descs[0], cols[0]
        targets = [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.SUM, 5)]
    else:                           # SELECT k, COUNT(*), AVG(d2) ... GROUP BY k: one 8-byte baseline key, 3 000 groups
        key = (rng.integers(0, n_groups, len(cols[0])) * 1_000_003 + 7).astype(np.int64)
        key_desc = InputColDescriptor(capi.INT64, False, ExpressionRange(False))
        targets = [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 5)]
    descs = [key_desc] + descs[1:]
    cols = [key] + cols[1:]
    nc = len(descs)
    ra = RelAlgExecutionUnit(descs, targets, [Qual(nc, capi.EQ, 1)], [0], exprs=[e.with_range(ExpressionRange(True, 0, 1, True))],
                             max_groups_buffer_entry_guess=max(8192, 2 * n_groups), num_tuples=len(key))
    h = len(key) // 2 // 4 * 4
    case = Case(name, ra, [[x[:h] for x in cols], [x[h:] for x in cols]])
    case.expect_error = err
    return case


@pytest.mark.parametrize("quals", ["two_columns", "three_columns_negated", "int64_column", "is_not_null"])
def test_several_plain_quals_in_front_of_the_partitioned_family_take_the_mask(sim, oracle, quals):
    """`a < K AND b > L` over a large-cardinality baseline GROUP BY used to fall to the row kernel (the partitioned family filters
    on ONE column): the quals are compiled into range atoms, the pre-pass leaves the row mask, the step runs on `mask = 1`"""
    from heavydb_amd.executor import Executor, Qual
    case = _mask_large_case("guarded_div", "k_part_scatter")
    case.ra.exprs = []
    case.expect_error = None
    case.ra.simple_quals = {"two_columns": [Qual(2, capi.LT, 300), Qual(3, capi.GT, -3)],
                            "three_columns_negated": [Qual(2, capi.NE, 7), Qual(3, capi.LE, 2), Qual(1, capi.GE, -900)],
                            "int64_column": [Qual(5, capi.GT, -10**11), Qual(2, capi.LT, 500)],
                            "is_not_null": [Qual(4, capi.IS_NOT_NULL, 0), Qual(4, capi.LT, 60), Qual(2, capi.GT, -800)]}[quals]
    rs = flow._check(oracle, case, kernel_variant=2)
    assert rs.report.kernel_name.decode() == "k_part_scatter", rs.report.kernel_name
    route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags], kernel_variant=2)
    assert "quals compiled" in route and "k_filter_mask" in route and "k_generic" not in route, route
    flow._check(oracle, case, kernel_variant=2, flags=capi.OPT_NO_COMPILED_FILTER)   # (the row kernel agrees)


def test_program_atoms_beside_plain_quals_and_range_atoms(sim, oracle):
    from heavydb_amd.executor import Qual
    for name in ("guarded_div", "sum_of_two_columns", "double_column"):
        case = _prog_atom_case(name, quals=[Qual(1, capi.GE, -500), Qual(4, capi.IS_NOT_NULL, 0)])
        rs = flow._check(oracle, case, kernel_variant=0)
        assert rs is not None and rs.report.kernel_name.decode() in ("k_groupby_lds", "k_perfect_lds")


def test_an_error_in_a_dropped_row_still_counts_when_the_expression_is_evaluated(sim, oracle):
    """a plain qual that drops the row does not stop the filter's expressions from being evaluated (the row function evaluates
    them first): the unguarded division raises although `v >= 2000` passes nothing"""
    from heavydb_amd.executor import Qual
    case = _prog_atom_case("unguarded_div", quals=[Qual(1, capi.GE, 2000)])
    flow._check(oracle, case, kernel_variant=0)


@pytest.mark.parametrize("idx", range(6), ids=["and_in_or", "not_or", "composed", "not_null_cmp_and_plain_qual", "short_circuit_or",
                                                "is_not_null_and_in_list"])
def test_compiled_bool_filters_in_the_scan_aggregate(sim, oracle, idx):
    """the same filters over a NON-GROUPED step — SELECT COUNT(*), SUM(v), MIN(c) FROM t WHERE <filter>, the shape of the reference's
    Select.FilterAndSimpleAggregation — run in k_scan_agg with the compiled filter, no interpreter pass"""
    from heavydb_amd.executor import Executor, TargetExpr
    case = _bool_filter_cases(oracle)[idx]
    case.ra.groupby_exprs = []
    case.ra.target_exprs = [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1), TargetExpr(capi.MIN, 4)]
    rs = flow._check(oracle, case)
    assert rs is not None and rs.report.kernel_name.decode() == "k_scan_agg", rs.report.kernel_name
    route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags])
    assert "filter compiled" in route and "k_project" not in route and "k_scan_agg" in route, route


# ---- the typed member under filters (round 5): up to three plain INT32 filter columns, range quals or a compiled filter
def _filtered_lds_case(oracle, quals, exprs=(), baseline=False, n=9003, nullable_flt=False, seed=11, count_only=False):
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    rng = np.random.default_rng(seed)
    I32 = capi.INT32
    g = rng.integers(0, 50, n).astype(np.int32)
    v = rng.integers(-1000, 1000, n).astype(np.int32)
    a = rng.integers(-20, 20, n).astype(np.int32)
    b = rng.integers(0, 100, n).astype(np.int32)
    c = rng.integers(0, 100, n).astype(np.int32)
    if nullable_flt:
        a[rng.random(n) < 0.1] = -2**31
        c[rng.random(n) < 0.2] = -2**31
    if baseline:
        key = (g.astype(np.int64) * 1000003 + 17)
        kd = InputColDescriptor(capi.INT64, False, ExpressionRange(False))
    else:
        key, kd = g, InputColDescriptor(I32, False, ExpressionRange(True, 0, 49))
    descs = [kd, InputColDescriptor(I32, False, ExpressionRange(True, -1000, 999)),
             InputColDescriptor(I32, nullable_flt, ExpressionRange(True, -20, 19, nullable_flt)),
             InputColDescriptor(I32, False, ExpressionRange(True, 0, 99)),
             InputColDescriptor(I32, nullable_flt, ExpressionRange(True, 0, 99, nullable_flt))]
    targets = [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT)] + ([] if count_only else [TargetExpr(capi.SUM, 1), TargetExpr(capi.MIN, 1)])
    ra = RelAlgExecutionUnit(descs, targets,
                             list(quals), [0], exprs=[e.with_range(ExpressionRange(True, 0, 1, True)) for e in exprs],
                             max_groups_buffer_entry_guess=256, num_tuples=n)
    h = n // 2 + 1
    cols = [key, v, a, b, c]
    return cases_mod.Case("f", ra, [[x[:h] for x in cols], [x[h:] for x in cols]])


def _typed_filter_quals():
    from heavydb_amd.executor import Qual
    return {
        "one_range": [Qual(2, capi.LT, 5)],
        "two_bounds_one_column": [Qual(3, capi.GT, 10), Qual(3, capi.LE, 80)],
        "three_columns": [Qual(2, capi.GE, -5), Qual(3, capi.LT, 70), Qual(4, capi.GT, 20)],
        "not_equal": [Qual(3, capi.NE, 42), Qual(2, capi.LT, 100)],
        "is_not_null": [Qual(2, capi.IS_NOT_NULL, 0), Qual(4, capi.LT, 50)],
        "is_null": [Qual(4, capi.IS_NULL, 0)],
        "bound_beyond_int32": [Qual(3, capi.LT, 1 << 40), Qual(2, capi.GT, -(1 << 40))],
        "empty_range": [Qual(3, capi.LT, -(1 << 40))],
        "empty_range_negated": [Qual(3, capi.NE, 1 << 40)],
        # more range filters than the run-time-role member's widest instantiation takes (negated quals do not merge)
        "five_negated_one_column": [Qual(3, capi.NE, k) for k in (1, 2, 3, 4, 5)],
        "seven_quals_three_columns": [Qual(3, capi.NE, 7), Qual(3, capi.NE, 9), Qual(3, capi.NE, 11), Qual(2, capi.NE, 0), Qual(2, capi.NE, 1),
                                      Qual(4, capi.NE, 50), Qual(4, capi.NE, 51)],
    }


@pytest.mark.parametrize("member", ["typed", "generic"])
@pytest.mark.parametrize("baseline", [False, True], ids=["perfect", "baseline"])
@pytest.mark.parametrize("nullable", [False, True], ids=["notnull", "nullable"])
@pytest.mark.parametrize("shape", list(_typed_filter_quals()))
def test_typed_lds_member_under_range_filters(sim, oracle, shape, nullable, baseline, member):
    if shape in ("is_not_null", "is_null") and not nullable:
        pytest.skip("the qual is constant on a NOT NULL column")
    case = _filtered_lds_case(oracle, _typed_filter_quals()[shape], baseline=baseline, nullable_flt=nullable)
    rs = flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_LDS_GENERIC_MEMBER if member == "generic" else 0)
    assert rs is not None
    name = rs.report.kernel_name.decode()
    if shape in ("five_negated_one_column", "seven_quals_three_columns"):   # too many filters for the LDS members: any family, the oracle's result
        return
    assert name in ("k_groupby_lds", "k_perfect_lds"), name   # (one range qual over a perfect-hash table: the older family)
    if name == "k_groupby_lds":
        assert rs.report.variant == (4 if member == "generic" else 5), rs.report.variant
    assert name == "k_groupby_lds" or (shape in ("one_range", "is_null", "empty_range", "empty_range_negated", "two_bounds_one_column") and not baseline), (name, shape)


@pytest.mark.parametrize("baseline", [False, True], ids=["perfect", "baseline"])
@pytest.mark.parametrize("nullable", [False, True], ids=["notnull", "nullable"])
def test_typed_lds_member_under_a_compiled_filter(sim, oracle, nullable, baseline):
    """(a < 5 AND b > 30) OR c IS NULL, and NOT (a < 0 OR b > 60): atoms + truth table inside the typed member"""
    from heavydb_amd.executor import Executor, Expr, Qual
    I32 = capi.INT32
    C_, L = Expr.col, lambda x: Expr.lit(I32, x)
    for e in (C_(2).cmp(capi.EX_LT, L(5)).logical(capi.EX_AND, C_(3).cmp(capi.EX_GT, L(30))).logical(capi.EX_OR, C_(4).is_null()),
              C_(2).cmp(capi.EX_LT, L(0)).logical(capi.EX_OR, C_(3).cmp(capi.EX_GT, L(60))).logical_not()):
        case = _filtered_lds_case(oracle, [Qual(5, capi.EQ, 1)], exprs=[e], baseline=baseline, nullable_flt=nullable)
        rs = flow._check(oracle, case, kernel_variant=0)
        assert rs is not None and rs.report.kernel_name.decode() == "k_groupby_lds" and rs.report.variant == 5, (rs.report.kernel_name, rs.report.variant)
        route = Executor(0).explain(case.ra, [len(f[0]) for f in case.frags])
        assert "filter compiled" in route and "k_project" not in route, route


@pytest.mark.parametrize("member", ["typed", "generic"])
@pytest.mark.parametrize("baseline", [False, True], ids=["perfect", "baseline"])
@pytest.mark.parametrize("shape", ["two_bounds_one_column", "three_columns", "not_equal", "is_not_null"])
def test_typed_lds_member_count_only_under_filters(sim, oracle, shape, baseline, member):
    """SELECT key, COUNT(*) FROM t WHERE ... GROUP BY key: no value column at all (the NV = 0 typed members)"""
    case = _filtered_lds_case(oracle, _typed_filter_quals()[shape], baseline=baseline, nullable_flt=True, count_only=True)
    rs = flow._check(oracle, case, kernel_variant=0, flags=capi.OPT_LDS_GENERIC_MEMBER if member == "generic" else 0)
    assert rs is not None
    name = rs.report.kernel_name.decode()
    if name == "k_groupby_lds":
        assert rs.report.variant == (4 if member == "generic" else 5), rs.report.variant
    else:
        assert name == "k_perfect_lds" and not baseline and shape == "two_bounds_one_column", (name, shape)
