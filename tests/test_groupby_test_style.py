"""Port of the reference's Tests/GroupByTest.cpp (HighCardinalityStringEnv): the layout decision for a
single dictionary-encoded string group key, pinned to what the reference's own test expects rather than to
the oracle's reading of GroupByAndAggregate.cpp.

    CREATE TABLE high_cardinality_str (x INT, str TEXT ENCODING DICT (32));     -- :60-65
    INSERT ... (1, 'hi'), (2, 'bye');
    SELECT COUNT(*) FROM high_cardinality_str WHERE x = 1 GROUP BY str          -- the work unit of :100-130

  PerfectHashNoFallback (:73-155)   range of `str` from the table's metadata, max_groups_buffer_entry_guess = 1,
                                    no cardinality estimation: runs (perfect hash needs no estimate), 1 row, COUNT = 1
  BaselineFallbackTest  (:173-262)  the cached range of `str` replaced by [0, 134217728] ("1 additional value over
                                    the max buffer size"): WITHOUT an estimate the call throws
                                    CardinalityEstimationRequired — i.e. the baseline layout was chosen — and
                                    WITH one it returns 1 row, COUNT = 1
  BaselineNoFilters     (:264-338)  the same range but no filter: "no filters, so expect no throw w/out cardinality
                                    estimation" — the perfect hash is kept; 2 rows, COUNT = 1 each

The seam of this tier sits below the estimator (the caller passes the estimate as max_groups_buffer_entry_guess), so
"throws CardinalityEstimationRequired" reads here as "desc_type is GROUP_BY_BASELINE_HASH"."""
import ctypes as C

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
from tests.cases import Case

X = np.array([1, 2], dtype=np.int32)
STR = np.array([0, 1], dtype=np.int32)        # dictionary ids of 'hi', 'bye'
TOO_BIG = 134217728                           # GroupByTest.cpp:188


def _unit(str_max, with_filter, guess):
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, str_max), capi.ENC_DICT, 0),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 1, 2))]
    quals = [Qual(1, capi.EQ, 1)] if with_filter else []
    return RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT)], quals, [0], max_groups_buffer_entry_guess=guess,
                               num_tuples=2)


def _qmds(oracle, ra):
    from tests.test_rowlogic_emu import emu_lib
    plan = ra.to_plan()
    out = []
    for fn in (oracle.lib().orc_qmd_init, emu_lib().emu_qmd_init, capi.load_library().mi355q_qmd_init):
        q = capi.QMD()
        assert fn(C.byref(plan), C.byref(q)) == 0
        out.append(q)
    return plan, out


def _counts(oracle, q, buf):
    iv, dv, nu = oracle.fetch_rows(q, buf)
    return sorted(int(v) for v in iv[:, 0])


@pytest.mark.parametrize("name,str_max,with_filter,guess,want_type,want_counts", [
    ("PerfectHashNoFallback", 1, True, 1, capi.GROUP_BY_PERFECT_HASH, [1]),
    ("BaselineFallbackTest", TOO_BIG, True, 16384, capi.GROUP_BY_BASELINE_HASH, [1]),
])
def test_high_cardinality_string_env(oracle, name, str_max, with_filter, guess, want_type, want_counts):
    from tests.test_rowlogic_emu import _emu_execute
    ra = _unit(str_max, with_filter, guess)
    plan, qs = _qmds(oracle, ra)
    for q in qs:   # oracle, product row logic built for the host, product library
        assert q.desc_type == want_type, name
    frags = [[STR, X]]
    q, buf, code = oracle.execute(plan, frags, n_threads=1)
    assert code == 0 and _counts(oracle, q, buf) == want_counts
    eq, ebuf, ecode = _emu_execute(Case(name, ra, frags), plan, None)
    assert ecode == 0 and _counts(oracle, eq, ebuf) == want_counts


def test_baseline_no_filters_keeps_the_perfect_hash(oracle):
    """The 134 M-entry table itself is only built on the device (tests/test_zz_gpu_execute_style.py)."""
    plan, qs = _qmds(oracle, _unit(TOO_BIG, False, 1))
    for q in qs:
        assert q.desc_type == capi.GROUP_BY_PERFECT_HASH
        assert q.entry_count == TOO_BIG + 1 and q.min_val == 0 and q.max_val == TOO_BIG


def test_an_integer_key_of_that_range_takes_the_baseline_layout(oracle):
    """GroupByAndAggregate.cpp:344: what the string special case is an exception FROM."""
    descs = [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, TOO_BIG)),
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 1, 2))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT)], [], [0], max_groups_buffer_entry_guess=16384, num_tuples=2)
    _, qs = _qmds(oracle, ra)
    for q in qs:
        assert q.desc_type == capi.GROUP_BY_BASELINE_HASH
