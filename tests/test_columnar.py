"""Columnar output buffers (output_columnar_): layout formulas against literals worked
out from the reference (QueryMemoryDescriptor.cpp getPrependedGroupColOffInBytes :962-975,
getColOffInBytes :906-929, getBufferSizeBytes :1084-1111, initColumnarGroups
QueryMemoryInitializer.cpp:713-780), and the whole case matrix with the hint switched on:

  oracle   — runs the step NATIVELY on the columnar buffer with the reference's *_columnar runtime
             functions (get_columnar_group_bin_offset, set_matching_group_value_perfect_hash_columnar,
             get_group_value_columnar_slot) and column + bin * width slot addresses;
  product  — runs the step on the row-wise form of the same decisions and moves the finished
             entries into the columns (api.cpp / rowfunc.h entry_to_columns), here through the host
             emulation of that code;
  numpy    — tests/helpers.columnar_to_rows restates the layout a third time to read both.
"""
import copy
import ctypes as C

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
from tests import cases as cases_mod
from tests.helpers import columnar_to_rows, compare_buffers, emu_lib, qmd_equal, rowwise_qmd
from tests.test_rowlogic_emu import _emu_execute, _oracle_join

CASES = [c for c in cases_mod.build_cases() if c.expect_error is None]


def _ra(groups, targets, ranges, guess=16384, types=None, **kw):
    descs = [InputColDescriptor(types[i] if types else capi.INT64, False, r) for i, r in enumerate(ranges)]
    return RelAlgExecutionUnit(descs, targets, [], groups, max_groups_buffer_entry_guess=guess,
                               output_columnar_hint=capi.OUTPUT_COLUMNAR, **kw)


def _offsets(q, oracle):
    lib, emu = capi.load_library(), emu_lib()
    prod = ([lib.mi355q_qmd_group_col_offset(C.byref(q), g) for g in range(q.group_col_count)],
            [lib.mi355q_qmd_slot_col_offset(C.byref(q), s) for s in range(q.slot_count)],
            lib.mi355q_qmd_buffer_bytes(C.byref(q)))
    em = ([emu.emu_group_col_offset(C.byref(q), g) for g in range(q.group_col_count)],
          [emu.emu_slot_col_offset(C.byref(q), s) for s in range(q.slot_count)],
          emu.emu_buffer_bytes(C.byref(q)))
    assert prod == em
    orc = ([oracle.col_group_off(q, g) for g in range(q.group_col_count)],
           [oracle.col_slot_off(q, s) for s in range(q.slot_count)], oracle.buffer_bytes(q))
    return prod, orc


def test_layout_literals(oracle):
    V = ExpressionRange
    # two group columns (perfect hash 5 x 3 = 15 entries), targets MAX, AVG, SUM of a nullable
    # column (no keyless candidate): keys at 0 and 120, slots (max, avg sum, avg count, sum) at
    # 240, 360, 480, 600; 720 bytes
    ra = _ra([0, 1], [TargetExpr(capi.MAX, 2), TargetExpr(capi.AVG, 2), TargetExpr(capi.SUM, 2)],
             [V(True, 0, 4), V(True, 10, 12), V(True, -5, 5, True)])
    ra.input_col_descs[2].nullable = True
    q = oracle.qmd_init(ra.to_plan())
    assert (q.desc_type, q.entry_count, q.output_columnar, q.keyless) == (capi.GROUP_BY_PERFECT_HASH, 15, 1, 0)
    prod, orc = _offsets(q, oracle)
    assert prod == orc == ([0, 120], [240, 360, 480, 600], 720)
    # keyless (single column, COUNT(*) first): no group columns at all
    ra = _ra([0], [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1)], [V(True, 0, 6), V(True, 0, 9)], bigint_count=True)
    q = oracle.qmd_init(ra.to_plan())
    assert (q.keyless, q.entry_count, q.slot_width) == (1, 7, 8)
    prod, orc = _offsets(q, oracle)
    assert prod == ([-1], [0, 56], 112) and orc[1:] == ([0, 56], 112)
    # 4-byte slots, odd entry count: every slot column is padded to a multiple of 8 bytes
    # (align_to_int64(4 * 7) = 32); keyless through COUNT(*) ...
    ra = _ra([0], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.COUNT)], [V(True, 0, 6)], types=[capi.INT32],
             num_tuples=1000)
    q = oracle.qmd_init(ra.to_plan())
    assert (q.slot_width, q.entry_count, q.keyless) == (4, 7, 1)
    prod, orc = _offsets(q, oracle)
    assert prod == ([-1], [0, 32], 64) and orc[1:] == ([0, 32], 64)
    # ... and keyed when the range is bucketed (0..12 step 2: 7 entries): one 8-byte key column first
    ra = _ra([0], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.COUNT)], [V(True, 0, 12, bucket=2)],
             types=[capi.INT32], num_tuples=1000)
    q = oracle.qmd_init(ra.to_plan())
    assert (q.slot_width, q.entry_count, q.keyless) == (4, 7, 0)
    prod, orc = _offsets(q, oracle)
    assert prod == orc == ([0], [56, 88], 120)
    # baseline hash: 8-byte key components whatever the ranges say ("output_columnar ? 8 : pick_baseline_key_width")
    small = V(True, 0, 10**6)  # 10^12 combinations: beyond the perfect-hash threshold
    ra = _ra([0, 1], [TargetExpr(capi.SUM, 2)], [small, small, small], guess=100, types=[capi.INT32] * 3)
    q = oracle.qmd_init(ra.to_plan())
    assert (q.desc_type, q.key_width, q.key_bytes, q.entry_count) == (capi.GROUP_BY_BASELINE_HASH, 8, 16, 100)
    prod, orc = _offsets(q, oracle)
    assert prod == orc == ([0, 800], [1600], 2400)
    row = copy.copy(ra)
    row.output_columnar_hint = 0
    assert oracle.qmd_init(row.to_plan()).key_width == 4
    # non-grouped: one entry, every column one 8-byte value = the row-wise bytes
    ra = _ra([], [TargetExpr(capi.COUNT), TargetExpr(capi.MAX, 0)], [V(True, 0, 9)])
    q = oracle.qmd_init(ra.to_plan())
    prod, orc = _offsets(q, oracle)
    assert prod == orc == ([], [0, 8], 16)
    # keyless single-column perfect hash whose first slot starts at EMPTY_KEY_64 (MIN over a NOT NULL BIGINT):
    # get_columnar_group_bin_offset takes that slot's column for the key column (GroupByRuntime.cpp:228-239).
    # The outcome does not depend on the row order: every row of a group writes the same key
    ra = _ra([0], [TargetExpr(capi.MIN, 1), TargetExpr(capi.COUNT)], [V(True, 0, 6), V(True, 1, 9)])
    p = ra.to_plan()
    qe, qo = capi.QMD(), capi.QMD()
    assert emu_lib().emu_qmd_init(C.byref(p), C.byref(qe)) == 0 and qe.keyless and qe.init_vals[0] == 2**63 - 1
    assert oracle.lib().orc_qmd_init(C.byref(p), C.byref(qo)) == 0
    rab = _ra([0], [TargetExpr(capi.MIN, 1), TargetExpr(capi.COUNT)], [V(True, 0, 600, bucket=100), V(True, 1, 9)])
    pb = rab.to_plan()
    # (a bucketed key never gets the keyless layout, so the order-dependent variant cannot arise)
    assert emu_lib().emu_qmd_init(C.byref(pb), C.byref(qe)) == 0 and not qe.keyless
    assert oracle.lib().orc_qmd_init(C.byref(pb), C.byref(qo)) == 0 and not qo.keyless
    p.output_columnar_hint = 7
    assert emu_lib().emu_qmd_init(C.byref(p), C.byref(qe)) == capi.ERR_INVALID_PLAN
    assert oracle.lib().orc_qmd_init(C.byref(p), C.byref(qo)) == capi.ERR_INVALID_PLAN


def test_init_images_agree(oracle):
    """initColumnarGroups: every key column EMPTY_KEY_64, every slot column its init value."""
    V = ExpressionRange
    for ra in (_ra([0, 1], [TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 2), TargetExpr(capi.MIN, 2)],
                   [V(True, 0, 4), V(True, 10, 12), V(True, -5, 5)]),
               _ra([0, 1], [TargetExpr(capi.SUM, 2)], [V(True, 0, 4), V(True, 10, 12), V(False)]),
               _ra([0], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.COUNT)], [V(True, 0, 6)],
                   types=[capi.INT32], num_tuples=1000),
               _ra([0], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.COUNT)], [V(True, 0, 12, bucket=2)],
                   types=[capi.INT32], num_tuples=1000),
               _ra([0], [TargetExpr(capi.MAX, 1)], [V(False), V(False)], guess=33)):
        q = oracle.qmd_init(ra.to_plan())
        want = oracle.init_buffer(q)
        got = np.zeros_like(want)
        emu_lib().emu_init_buffer(C.byref(q), got.ctypes.data)
        assert np.array_equal(want, got)
        rows = columnar_to_rows(q, want)
        qr = rowwise_qmd(q)
        assert np.array_equal(rows, oracle.init_buffer(qr))
        assert oracle.row_count(q, want) == 0


def test_first_slot_taken_for_the_key_column(oracle):
    """The reference's columnar keyless quirk, literally: SELECT MIN(v), COUNT(*) GROUP BY k with k in [0, 6] and
    v in [3, 9] — in a columnar buffer slot 0 of group k comes out as MIN(k, MIN(v)), not MIN(v), because
    get_columnar_group_bin_offset writes the key into a first-slot entry that still equals EMPTY_KEY_64.  The
    row-wise step of the same query is unaffected."""
    from tests.cases import Case
    V = ExpressionRange
    rng = np.random.default_rng(3)
    k = rng.integers(0, 7, 500).astype(np.int64)
    v = rng.integers(3, 10, 500).astype(np.int64)
    frags = [[k[:250], v[:250]], [k[250:], v[250:]]]
    ra = _ra([0], [TargetExpr(capi.MIN, 1), TargetExpr(capi.COUNT)], [V(True, 0, 6), V(True, 3, 9)])
    plan = ra.to_plan()
    q, want, code = oracle.execute(plan, frags, n_threads=2)
    assert code == 0 and q.output_columnar == 1 and q.keyless == 1
    rows = columnar_to_rows(q, want).reshape(q.entry_count, -1)
    for key in range(7):
        assert rows[key, 0] == min(key, int(v[k == key].min())) and rows[key, 1] == int((k == key).sum())
    eq, got, ecode = _emu_execute(Case("quirk", ra, frags), plan, None)
    assert ecode == 0
    qmd_equal(q, eq)
    assert np.array_equal(np.asarray(want).view(np.int64), np.asarray(got).view(np.int64))
    row = copy.copy(ra)
    row.output_columnar_hint = 0
    qr, wr, code = oracle.execute(row.to_plan(), frags, n_threads=2)
    rr = np.asarray(wr).view(np.int64).reshape(qr.entry_count, -1)
    for key in range(7):
        assert rr[key, -2] == int(v[k == key].min())


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_columnar_step_matches_oracle(oracle, case):
    ra = copy.copy(case.ra)
    ra.output_columnar_hint = capi.OUTPUT_COLUMNAR
    plan = ra.to_plan()
    probe = capi.QMD()
    rc = oracle.lib().orc_qmd_init(C.byref(plan), C.byref(probe))
    if rc == capi.ERR_UNSUPPORTED:  # the refused keyless corner: both sides say so
        assert emu_lib().emu_qmd_init(C.byref(plan), C.byref(capi.QMD())) == capi.ERR_UNSUPPORTED
        pytest.skip("columnar keyless with an EMPTY_KEY_64 first slot is refused")
    oj = _oracle_join(oracle, case)
    q, want, code = oracle.execute(plan, case.frags, case.inner, oj, n_threads=3)
    eq, got, ecode = _emu_execute(case, plan, oj)
    assert code == 0 and ecode == 0, (code, ecode)
    assert q.output_columnar == 1
    qmd_equal(q, eq)
    assert want.nbytes == got.nbytes == oracle.buffer_bytes(q)
    qr = rowwise_qmd(q)
    w_rows, g_rows = columnar_to_rows(q, want), columnar_to_rows(q, got)
    compare_buffers(qr, w_rows, g_rows, case.fp_rtol)
    # the columnar step holds the same groups and values as the row-wise step of the same decisions
    ra2 = copy.copy(case.ra)
    ra2.output_columnar_hint = capi.OUTPUT_ROWWISE_COLUMNAR_DECISIONS
    q2, rows2, code2 = oracle.execute(ra2.to_plan(), case.frags, case.inner, oj, n_threads=3)
    assert code2 == 0 and q2.output_columnar == 0 and q2.row_size == q.row_size
    compare_buffers(qr, rows2, w_rows, case.fp_rtol)
    # iteration through the columnar accessors
    iv, dv, nu = oracle.fetch_rows(q, want)
    iv2, dv2, nu2 = oracle.fetch_rows(qr, w_rows)
    assert np.array_equal(iv, iv2) and np.array_equal(nu, nu2) and np.array_equal(dv, dv2, equal_nan=True)
    assert oracle.row_count(q, want) == oracle.row_count(qr, w_rows)


def test_columnar_reduce(oracle):
    """this (op)= that on columnar buffers: oracle vs the product's reduce code, perfect and baseline."""
    rng = np.random.default_rng(5)
    V = ExpressionRange
    n = 4000
    for baseline in (False, True):
        key = rng.integers(0, 50, n).astype(np.int64) * (1000003 if baseline else 1)
        val = rng.integers(-1000, 1000, n).astype(np.int64)
        kr = V(False) if baseline else V(True, 0, 49)
        ra = _ra([0], [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1),
                       TargetExpr(capi.MIN, 1)], [kr, V(True, -1000, 999)], guess=256)
        plan = ra.to_plan()
        halves = []
        for sl in (slice(0, n // 2), slice(n // 2, n)):
            q, buf, code = oracle.execute(plan, [[key[sl], val[sl]]])
            assert code == 0
            halves.append(buf)
        a_o, a_e = halves[0].copy(), halves[0].copy()
        assert oracle.reduce(q, a_o, halves[1]) == 0
        assert emu_lib().emu_reduce(C.byref(q), a_e.ctypes.data, halves[1].ctypes.data, q.entry_count) == 0
        qr = rowwise_qmd(q)
        compare_buffers(qr, columnar_to_rows(q, a_o), columnar_to_rows(q, a_e))
        qf, full, code = oracle.execute(plan, [[key, val]])
        assert code == 0
        compare_buffers(qr, columnar_to_rows(q, full), columnar_to_rows(q, a_o))


def test_layout_decisions_agree_with_hints(oracle):
    """plan.cpp vs oracle on random plans with a random output_columnar_hint (0 / 1 / 2)."""
    from tests.test_plan_fuzz import _random_plan
    rng, hrng = np.random.default_rng(4040), np.random.default_rng(41)
    emu = emu_lib()
    seen = set()
    for i in range(2500):
        ra = _random_plan(rng)
        ra.output_columnar_hint = int(hrng.integers(0, 3))
        plan = ra.to_plan()
        qe, qo = capi.QMD(), capi.QMD()
        ce = emu.emu_qmd_init(C.byref(plan), C.byref(qe))
        co = oracle.lib().orc_qmd_init(C.byref(plan), C.byref(qo))
        assert ce == co or ((ce == 0) == (co == 0) and ce != 0), (i, ce, co)
        if ce:
            continue
        de, do = qe.as_dict(), qo.as_dict()
        assert de == do, (i, {k: (de[k], do[k]) for k in de if de[k] != do[k]})
        assert qe.output_columnar == (ra.output_columnar_hint == 1)
        if ra.output_columnar_hint and qe.desc_type == capi.GROUP_BY_BASELINE_HASH:
            assert qe.key_width == 8
        assert emu.emu_buffer_bytes(C.byref(qe)) == oracle.buffer_bytes(qo)
        if qe.output_columnar:
            for s in range(qe.slot_count):
                assert emu.emu_slot_col_offset(C.byref(qe), s) == oracle.col_slot_off(qo, s)
            for g in range(0 if qe.keyless else qe.group_col_count):
                assert emu.emu_group_col_offset(C.byref(qe), g) == oracle.col_group_off(qo, g)
        seen.add((qe.desc_type, qe.keyless, qe.slot_width, qe.output_columnar))
    assert len([k for k in seen if k[3]]) >= 5, sorted(seen)


def test_columnar_row_logic_fuzz(oracle):
    """Random tables x random plans, columnar: the oracle's native columnar step against the
    product's row-wise step + entry_to_columns (host emulation)."""
    from tests.cases import Case
    from tests.test_plan_fuzz import _fuzz_row_plan, _fuzz_table
    rng = np.random.default_rng(7117)
    emu = emu_lib()
    ran = refused = errors = 0
    for i in range(100):   # (tools/soak_fuzz.py runs thousands with fresh seeds; the suite keeps a sample)
        n_rows = int(rng.integers(1, 400))
        descs, cols = _fuzz_table(rng, n_rows)
        ra = _fuzz_row_plan(rng, descs)
        ra.output_columnar_hint = capi.OUTPUT_COLUMNAR
        cut = n_rows // 2
        frags = [[c[:cut] for c in cols], [c[cut:] for c in cols]]
        plan = ra.to_plan()
        qo, qe = capi.QMD(), capi.QMD()
        co = oracle.lib().orc_qmd_init(C.byref(plan), C.byref(qo))
        ce = emu.emu_qmd_init(C.byref(plan), C.byref(qe))
        if co or ce:
            assert co == ce or (co != 0 and ce != 0), (i, co, ce)
            refused += 1
            continue
        q, want, code = oracle.execute(plan, frags, n_threads=2)
        eq, got, ecode = _emu_execute(Case(f"colfuzz{i}", ra, frags), plan, None)
        if code != 0 or ecode != 0:
            assert code != 0 and ecode != 0, (i, code, ecode)
            errors += 1
            continue
        qmd_equal(q, eq)
        qr = rowwise_qmd(q)
        compare_buffers(qr, columnar_to_rows(q, want), columnar_to_rows(q, got), 1e-9)
        ran += 1
    assert ran > 55, (ran, refused, errors)


def test_columnar_join_fuzz(oracle):
    from tests.test_plan_fuzz import _fuzz_join
    rng = np.random.default_rng(9119)
    for i in range(150):
        case = _fuzz_join(rng)
        case.ra.output_columnar_hint = capi.OUTPUT_COLUMNAR
        plan = case.ra.to_plan()
        if oracle.lib().orc_qmd_init(C.byref(plan), C.byref(capi.QMD())) == capi.ERR_UNSUPPORTED:
            assert emu_lib().emu_qmd_init(C.byref(plan), C.byref(capi.QMD())) == capi.ERR_UNSUPPORTED
            continue
        oj = _oracle_join(oracle, case)
        q, want, code = oracle.execute(plan, case.frags, case.inner, oj, n_threads=2)
        eq, got, ecode = _emu_execute(case, plan, oj)
        assert code == 0 and ecode == 0, (i, code, ecode)
        qmd_equal(q, eq)
        compare_buffers(rowwise_qmd(q), columnar_to_rows(q, want), columnar_to_rows(q, got), 1e-9)
