"""SURVEY §8 rows a7 / a8 pinned to values the reference itself computed: tests/golden/ref_layout_vectors.json holds, for
600 random plans, what the reference's own get_target_info_impl (Shared/TargetInfo.cpp), ColSlotContext
(QueryEngine/Descriptors/ColSlotContext.cpp) and init_agg_val_vec (QueryEngine/OutputBufferInitialization.cpp) produce —
those sources compiled unmodified into oracle/_ref/libref_layout.so and run by oracle/gen_golden_layout.py in the build
container.  Both restatements of the layout code are held to it: oracle/oracle.cpp (qmd_init) and the product's
heavydb_amd/csrc/plan.cpp (through the host emulation library)."""
import ctypes as C
import json
import os

import pytest

from heavydb_amd import capi
from heavydb_amd.executor import ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
from tests.helpers import emu_lib

kFLOAT, kDOUBLE = 8, 9
HERE = os.path.dirname(__file__)


def _vectors():
    with open(os.path.join(HERE, "golden", "ref_layout_vectors.json")) as f:
        return json.load(f)


def _unit(d) -> RelAlgExecutionUnit:
    cols = [InputColDescriptor(c["type"], c["nullable"], ExpressionRange(**c["range"]), c["encoding"], c["logical_type"])
            for c in d["input_col_descs"]]
    targets = [TargetExpr(t["agg"], t["col"], t["table"], Qual(**t["cond"]) if t["cond"] else None)
               for t in d["target_exprs"]]
    return RelAlgExecutionUnit(cols, targets, [Qual(**q) for q in d["simple_quals"]], d["groupby_exprs"],
                               max_groups_buffer_entry_guess=d["max_groups_buffer_entry_guess"],
                               bigint_count=d["bigint_count"], output_columnar_hint=d["output_columnar_hint"],
                               num_tuples=d["num_tuples"])


def _check(case, q, who):
    ref, ra = case["ref"], case["unit"]
    given = case["given"]
    # the decisions that were inputs to the reference run must still be the ones this descriptor makes
    assert (q.desc_type, q.keyless, q.slot_width, q.output_columnar, q.group_col_count) == (
        given["desc_type"], given["keyless"], given["slot_width"], given["output_columnar"],
        given["group_col_count"]), who
    live = [w for w in ref["slot_padded"] if w > 0]   # zero-width slots: targets read from the key columns
    assert q.slot_count == len(live), (who, q.slot_count, ref["slot_padded"])
    assert all(w == q.slot_width for w in live), (who, q.slot_width, ref["slot_padded"])
    if not q.output_columnar:
        # getRowSize (QueryMemoryDescriptor.cpp:848-860) = align8(align8(key bytes) + getColsSize())
        assert q.row_size - q.key_bytes == (ref["aligned_padded_size"] + 7) // 8 * 8, (who, q.row_size, q.key_bytes, ref)
    assert [int(v) for v in ref["init_vals"]] == list(q.init_vals[:q.slot_count]), (
        who, ref["init_vals"], list(q.init_vals[:q.slot_count]), ra["target_exprs"], ra["simple_quals"])
    # first slot of each target, from the reference's slot list (AVG owns two slots)
    slot, live_slot = 0, 0
    for i, t in enumerate(ra["target_exprs"]):
        n = 2 if t["agg"] == capi.AVG else 1
        if ref["slot_padded"][slot] > 0:
            assert q.target_slot[i] == live_slot, (who, i, q.target_slot[i], live_slot)
            live_slot += n
        else:
            assert q.target_slot[i] == -1, (who, i)
        slot += n
        if t["agg"] == capi.PROJECT_KEY:
            assert not ref["ti_is_agg"][i]
            continue
        assert ref["ti_is_agg"][i]
        # TargetInfo.skip_null_val, before constrained_not_null clears it for a qualified argument
        constrained = any(s["op"] == capi.IS_NOT_NULL and s["col"] == t["col"] for s in ra["simple_quals"])
        forced = q.desc_type == capi.NON_GROUPED_AGGREGATE and (t["col"] >= 0 or t["agg"] == capi.COUNT_IF)
        if forced:   # a non-grouped aggregate with an argument always skips NULLs (TargetExprBuilder.cpp:684-690)
            assert q.target_skip_null[i], (who, i, t)
        elif not constrained and t["agg"] != capi.COUNT_IF:
            assert bool(q.target_skip_null[i]) == bool(ref["ti_skip_null"][i]), (who, i, t)
        if t["agg"] in (capi.SUM, capi.MIN, capi.MAX, capi.SUM_IF):
            assert bool(q.target_is_fp[i]) == (ref["ti_sql_type"][i] in (kFLOAT, kDOUBLE)), (who, i, t)
        if t["agg"] != capi.COUNT_IF and t["col"] >= 0:
            assert bool(q.target_arg_is_f32[i]) == (ref["ti_arg_type"][i] == kFLOAT and t["agg"] != capi.COUNT), (who, i)


def test_oracle_descriptor_matches_reference_layout_code(oracle):
    v = _vectors()
    assert len(v["cases"]) >= 500
    for n, case in enumerate(v["cases"]):
        plan = _unit(case["unit"]).to_plan()
        q = capi.QMD()
        assert oracle.lib().orc_qmd_init(C.byref(plan), C.byref(q)) == 0, n
        _check(case, q, ("oracle", n))


def test_product_descriptor_matches_reference_layout_code():
    emu = emu_lib()
    for n, case in enumerate(_vectors()["cases"]):
        plan = _unit(case["unit"]).to_plan()
        q = capi.QMD()
        assert emu.emu_qmd_init(C.byref(plan), C.byref(q)) == 0, n
        _check(case, q, ("plan.cpp", n))
