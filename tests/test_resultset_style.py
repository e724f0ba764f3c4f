"""Ports of the reference's own ResultSet fill / reduce / iterate tests (Tests/ResultSetTest.cpp
Reduce.PerfectHashOneCol :1591, Reduce.PerfectHashOneColKeyless :1660, Reduce.BaselineHashOneCol,
Iterate.* :1398-1584) with its fillers restated (Tests/ResultSetTestUtils.cpp:104-160 fill_one_entry_no_
collisions, :322-351 fill_storage_buffer_perfect_hash_rowwise, :395-430 ..._baseline_rowwise,
EvenNumberGenerator ResultSetTestUtils.h:39-52): every second entry holds the value v = 0, 2, 4, ...
in every target (AVG as the pair (v, 1)); two such buffers are reduced and iterated; the
expectations are the reference test's own formulas (test_reduce :1026-1100: SUM / COUNT = step *
entry, everything else = entry).  Run against the oracle's reduce / iteration AND the product's
reduce logic (host emulation of k_reduce)."""
import ctypes as C

import numpy as np
import pytest

from heavydb_amd import capi
from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
from tests.helpers import emu_lib

EMPTY64 = 2**63 - 1
DEADBEEF = 0xdeadbeef


def _qmd(oracle, baseline: bool, keyless: bool):
    """key int64; targets: projected key (the reference's non-aggregate column), AVG(int), SUM(int) —
    generate_test_target_infos (:938-955) without the string column."""
    key_range = ExpressionRange(False) if baseline else ExpressionRange(True, 0, 99)
    descs = [InputColDescriptor(capi.INT64, False, key_range), InputColDescriptor(capi.INT32, True, ExpressionRange(False))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.AVG, 1), TargetExpr(capi.SUM, 1)],
                             groupby_exprs=[0], max_groups_buffer_entry_guess=200)
    q = oracle.qmd_init(ra.to_plan())
    if baseline:
        assert q.desc_type == capi.GROUP_BY_BASELINE_HASH and q.entry_count == 200 and q.key_bytes == 8
    else:
        assert q.desc_type == capi.GROUP_BY_PERFECT_HASH and q.entry_count == 100 and not q.keyless
        if keyless:  # setHasKeylessHash(true); setTargetIdxForKey(2): AVG's count slot marks live entries
            q.keyless, q.idx_target_as_key, q.key_bytes = 1, 2, 0
            q.row_size = 8 * q.slot_count
    return q


def _fill_perfect(q, step=2):
    """fill_storage_buffer_perfect_hash_rowwise + fill_one_entry_no_collisions"""
    rq, kq = q.row_size // 8, q.key_bytes // 8
    buf = np.zeros((q.entry_count, rq), dtype=np.int64)
    v = 0
    for i in range(q.entry_count):
        if i % step == 0:
            if kq:
                buf[i, 0] = v
            buf[i, kq:] = [v, v, 1, v]  # projected column, AVG (sum, count), SUM
            v += 2
        else:
            if kq:
                buf[i, 0] = EMPTY64
            buf[i, kq:] = [0, 0, 0, 0] if q.keyless else [DEADBEEF, DEADBEEF, 0, DEADBEEF]
    return buf


def _check_reduced(q, buf, fetch, step=2):
    iv, dv, nu = fetch(q, buf)
    assert iv.shape[0] == (q.entry_count + step - 1) // step if q.desc_type != capi.GROUP_BY_BASELINE_HASH else True
    for r in range(iv.shape[0]):
        entry = int(iv[r, 0])  # the projected column is the entry's own value
        assert entry % 2 == 0
        assert not nu[r].any()
        assert dv[r, 1] == float(entry)          # AVG = (v + v) / (1 + 1)
        assert iv[r, 2] == step * entry          # SUM = step * entry (two buffers, step 2)
    return iv[:, 0]


@pytest.mark.parametrize("keyless", [False, True], ids=["PerfectHashOneCol", "PerfectHashOneColKeyless"])
@pytest.mark.parametrize("impl", ["oracle", "product_reduce"])
def test_reduce_perfect_hash_one_col(oracle, keyless, impl):
    q = _qmd(oracle, False, keyless)
    a, b = _fill_perfect(q), _fill_perfect(q)
    if impl == "oracle":
        assert oracle.reduce(q, a, b) == 0
    else:
        assert emu_lib().emu_reduce(C.byref(q), a.ctypes.data, b.ctypes.data, q.entry_count) == 0
    keys = _check_reduced(q, a, oracle.fetch_rows)
    assert list(keys) == list(range(0, 100, 2))          # entry order, Iterate.* expectations
    assert oracle.row_count(q, a) == 50


@pytest.mark.parametrize("impl", ["oracle", "product_reduce"])
def test_reduce_baseline_hash_one_col(oracle, impl):
    """Reduce.BaselineHashOneCol: the filler places v = 0, 2, 4, ... with get_group_value
    (fill_storage_buffer_baseline_rowwise), the reduction re-hashes them."""
    q = _qmd(oracle, True, False)
    rq = q.row_size // 8

    def fill():
        buf = np.zeros((q.entry_count, rq), dtype=np.int64)
        buf[:, 0] = EMPTY64
        # the baseline layout projects the key from the key column itself: slots are AVG (sum, count), SUM
        buf[:, 1:] = [DEADBEEF, 0, DEADBEEF]  # "kCOUNT ? 0 : 0xdeadbeef" — AVG's count starts at 0
        flat = buf.reshape(-1)
        for v in range(0, 2 * (q.entry_count // 2), 2):
            s = oracle.lib().orc_get_group_value_slot(flat.ctypes.data, q.entry_count, v, 8, rq)
            assert s >= 0
            flat[s:s + 3] = [v, 1, v]
        return buf
    a, b = fill(), fill()
    if impl == "oracle":
        assert oracle.reduce(q, a, b) == 0
    else:
        assert emu_lib().emu_reduce(C.byref(q), a.ctypes.data, b.ctypes.data, q.entry_count) == 0
    keys = _check_reduced(q, a, oracle.fetch_rows)
    assert sorted(keys) == list(range(0, 200, 2))
    # the slots agree with the key column, and the table is still a valid probing image
    live = a[a[:, 0] != EMPTY64]
    assert (live[:, 3] == 2 * live[:, 0]).all() and (live[:, 2] == 2).all()
    from tests.helpers import check_probe_invariant
    check_probe_invariant(q, a.reshape(-1))


def test_iterate_perfect_hash_one_col(oracle):
    """Iterate.PerfectHashOneCol (:1398): rows come out in entry order with ref_val += 2."""
    q = _qmd(oracle, False, False)
    buf = _fill_perfect(q)
    iv, dv, nu = oracle.fetch_rows(q, buf)
    ref = 0
    for r in range(iv.shape[0]):
        assert iv[r, 0] == ref and dv[r, 1] == float(ref) and iv[r, 2] == ref and not nu[r].any()
        ref += 2
    assert ref == 100


# ---------------------------------------------------------------------------------------------
# Two key columns and COLUMNAR buffers: Reduce.PerfectHashTwoCol{,Keyless,Columnar,ColumnarKeyless}
# (:1735-1808), Reduce.PerfectHashOneColColumnar{,Keyless} (:1607, :1677), Reduce.BaselineHash{,Columnar}
# (:1811-1827) with the colwise fillers restated (fill_storage_buffer_perfect_hash_colwise
# ResultSetTestUtils.cpp:250-320, fill_storage_buffer_baseline_colwise :352-393: every column
# advance_to_next_columnar_*_buff = align_to_int64(width * entry_count) after the previous one).
def _qmd2(oracle, n_keys: int, baseline: bool, keyless: bool, columnar: bool, entries: int = 36):
    side = 6 if n_keys == 2 else entries
    kr = ExpressionRange(False) if baseline else ExpressionRange(True, 0, side - 1)
    descs = [InputColDescriptor(capi.INT64, False, kr) for _ in range(n_keys)] + \
        [InputColDescriptor(capi.INT32, True, ExpressionRange(False))]
    ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.PROJECT_KEY, 0), TargetExpr(capi.AVG, n_keys), TargetExpr(capi.SUM, n_keys)],
                             groupby_exprs=list(range(n_keys)), max_groups_buffer_entry_guess=entries,
                             output_columnar_hint=capi.OUTPUT_COLUMNAR if columnar else capi.OUTPUT_ROWWISE)
    q = oracle.qmd_init(ra.to_plan())
    assert q.entry_count == entries and q.output_columnar == int(columnar) and not q.keyless
    assert q.desc_type == (capi.GROUP_BY_BASELINE_HASH if baseline else capi.GROUP_BY_PERFECT_HASH)
    if keyless:  # setHasKeylessHash(true); setTargetIdxForKey(2)
        q.keyless, q.idx_target_as_key, q.key_bytes = 1, 2, 0
        q.row_size = 8 * q.slot_count
    return q


def _assemble(q, key_cols, slot_cols):
    """row-wise: [entry, quads]; columnar: the columns laid end to end, each align8(8 * entries)"""
    if q.output_columnar:
        return np.concatenate([np.asarray(c, dtype=np.int64) for c in key_cols + slot_cols])
    return np.stack([np.asarray(c, dtype=np.int64) for c in key_cols + slot_cols], axis=1).copy()


def _fill_perfect2(q, n_keys, step=2):
    E = q.entry_count
    live = np.arange(E) % step == 0
    v = np.where(live, 2 * (np.arange(E) // step), 0)
    keys = [] if q.keyless else [np.where(live, v, EMPTY64) for _ in range(n_keys)]
    dead = 0 if q.keyless else DEADBEEF
    slots = [np.where(live, v, dead) for _ in range(q.slot_count)]
    if q.target_slot[0] < 0:  # the baseline layout projects the key from the key column
        base = 0
    else:
        base = 1
    slots[base + 1] = np.where(live, 1, 0)  # AVG's count
    return _assemble(q, keys, slots)


def _reduce_both(oracle, q, a, b):
    a_o, a_e = a.copy(), a.copy()
    assert oracle.reduce(q, a_o, b) == 0
    assert emu_lib().emu_reduce(C.byref(q), a_e.ctypes.data, b.ctypes.data, q.entry_count) == 0
    return a_o, a_e


@pytest.mark.parametrize("n_keys", [1, 2])
@pytest.mark.parametrize("keyless", [False, True], ids=["keyed", "keyless"])
@pytest.mark.parametrize("columnar", [False, True], ids=["rowwise", "columnar"])
def test_reduce_perfect_hash_layouts(oracle, n_keys, keyless, columnar):
    q = _qmd2(oracle, n_keys, False, keyless, columnar)
    a, b = _fill_perfect2(q, n_keys), _fill_perfect2(q, n_keys)
    for red in _reduce_both(oracle, q, a, b):
        iv, dv, nu = oracle.fetch_rows(q, red)
        assert iv.shape[0] == 18 and oracle.row_count(q, red) == 18
        for r in range(18):
            entry = 2 * r  # EvenNumberGenerator, step 2: entry i holds v = i
            assert iv[r, 0] == entry and dv[r, 1] == float(entry) and iv[r, 2] == 2 * entry and not nu[r].any()
    if columnar:  # the columns are where getColOffInBytes says
        raw = a.view(np.int64)
        nk = 0 if keyless else n_keys
        assert raw.size == (nk + q.slot_count) * 36
        for k in range(nk):
            assert oracle.col_group_off(q, k) == 8 * 36 * k
        for s in range(q.slot_count):
            assert oracle.col_slot_off(q, s) == 8 * 36 * (nk + s)


@pytest.mark.parametrize("columnar", [False, True], ids=["rowwise", "columnar"])
def test_reduce_baseline_hash_two_col(oracle, columnar):
    """Reduce.BaselineHash / BaselineHashColumnar: EvenNumberGenerator against ReverseOddOrEven, step 1,
    keys (v, v) placed by get_group_value / get_group_value_columnar; every key appears once, so after
    the reduction and the sort on the first target row r holds r everywhere."""
    n, E = 16, 64
    q = _qmd2(oracle, 2, True, False, columnar, entries=E)
    assert q.key_width == 8 and q.key_bytes == 16 and q.target_slot[0] == -1

    def fill(values):
        buf = oracle.init_buffer(q)
        rows = np.zeros((E, q.row_size // 8), dtype=np.int64)
        for v in values:
            key = np.array([v, v], dtype=np.int64)
            if columnar:
                b = oracle.lib().orc_get_group_value_columnar_slot(buf.ctypes.data, E, key.ctypes.data, 2)
                assert b >= 0
                flat = buf.view(np.int64)
                for s, x in enumerate([v, 1, v]):  # AVG (sum, count), SUM
                    flat[oracle.col_slot_off(q, s) // 8 + b] = x
            else:
                flat = buf.reshape(-1)
                s0 = oracle.lib().orc_get_group_value_n_slot(flat.ctypes.data, E, key.ctypes.data, 2, 8, q.row_size // 8)
                assert s0 >= 0
                flat[s0:s0 + 3] = [v, 1, v]
        del rows
        return buf
    a = fill(range(0, 2 * n, 2))                 # EvenNumberGenerator
    b = fill(range(2 * n - 1, 0, -2))            # ReverseOddOrEvenNumberGenerator(2 * n - 1)
    for red in _reduce_both(oracle, q, a, b):
        iv, dv, nu = oracle.fetch_rows(q, red)
        order = np.argsort(iv[:, 0])
        iv, dv = iv[order], dv[order]
        assert iv.shape[0] == 2 * n
        for r in range(2 * n):
            assert iv[r, 0] == r and dv[r, 1] == float(r) and iv[r, 2] == r  # step 1: SUM = 1 * row_idx
