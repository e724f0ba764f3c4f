"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every
symbol include/mi355q.h declares, and the ctypes mirror matches the compiled structs.  No
compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from heavydb_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mi355q.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355q_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_loads():
    from heavydb_amd import _build
    path = _build.build()
    assert os.path.exists(path)
    lib = capi.load_library()
    assert lib.mi355q_abi_version() == capi.ABI_VERSION


def test_every_declared_symbol_is_exported_and_bound():
    lib = capi.load_library()
    declared = _declared_symbols()
    bound = {name for name, _, _ in capi.SYMBOLS}
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mi355q.h but not exported"
        assert name in bound, f"{name} not bound in heavydb_amd/capi.py"
    assert bound <= set(declared)


def test_struct_mirrors_match_the_library():
    lib = capi.load_library()
    for which, st in ((1, capi.Plan), (2, capi.QMD), (3, capi.Inputs), (4, capi.ExecOptions),
                      (5, capi.ExecReport), (6, capi.JoinSpec)):
        assert lib.mi355q_abi_sizeof(which) == C.sizeof(st)


def test_qmd_init_through_the_abi_matches_oracle(oracle):
    """Plan-time layout decisions are host code: callable without a GPU."""
    from tests import cases
    from tests.helpers import qmd_equal
    lib = capi.load_library()
    for c in cases.build_cases():
        plan = c.ra.to_plan()
        q = capi.QMD()
        assert lib.mi355q_qmd_init(C.byref(plan), C.byref(q)) == 0
        qmd_equal(oracle.qmd_init(plan), q)
        assert lib.mi355q_qmd_buffer_bytes(C.byref(q)) == q.entry_count * q.row_size


def test_invalid_plans_are_rejected():
    from heavydb_amd.executor import InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    lib = capi.load_library()
    q = capi.QMD()
    ra = RelAlgExecutionUnit([InputColDescriptor(capi.INT32)], [TargetExpr(capi.SUM, -1)])
    assert lib.mi355q_qmd_init(C.byref(ra.to_plan()), C.byref(q)) == capi.ERR_INVALID_PLAN
    ra = RelAlgExecutionUnit([InputColDescriptor(capi.INT32)], [TargetExpr(capi.PROJECT_KEY)])
    assert lib.mi355q_qmd_init(C.byref(ra.to_plan()), C.byref(q)) == capi.ERR_INVALID_PLAN
    ra = RelAlgExecutionUnit([InputColDescriptor(capi.DOUBLE)], [TargetExpr(capi.COUNT)], groupby_exprs=[0])
    # a floating-point group key is the baseline layout with an 8-byte key, whatever the range says
    assert lib.mi355q_qmd_init(C.byref(ra.to_plan()), C.byref(q)) == 0
    assert (q.desc_type, q.key_width) == (capi.GROUP_BY_BASELINE_HASH, 8)
    ra = RelAlgExecutionUnit([InputColDescriptor(capi.INT32)], [TargetExpr(capi.COUNT)], groupby_exprs=[0], output_columnar_hint=9)
    assert lib.mi355q_qmd_init(C.byref(ra.to_plan()), C.byref(q)) == capi.ERR_INVALID_PLAN
    assert lib.mi355q_error_string(3) == b"Out of Slots"


def test_product_path_fails_loudly_without_the_library(tmp_path):
    """No CPU fallback: a missing HIP library is an error at load time, and nothing under heavydb_amd/
    imports the oracle (test infrastructure)."""
    import re
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load_library(str(tmp_path / "libmi355q.so"))
    pkg = os.path.join(ROOT, "heavydb_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src and "oracle/" not in src.replace("oracle/oracle.cpp)", ""), f


def test_execute_without_a_device_reports_an_error_code():
    """On a box without a GPU the entry points return a HIP error code (never a result)."""
    import ctypes as C
    lib = capi.load_library()
    if lib.mi355q_device_count() > 0:
        pytest.skip("a GPU is present")
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor, RelAlgExecutionUnit, TargetExpr
    ra = RelAlgExecutionUnit([InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 9))], [TargetExpr(capi.COUNT)])
    plan = ra.to_plan()
    inp = capi.Inputs()
    out = C.c_void_p()
    rc = lib.mi355q_execute(C.byref(plan), C.byref(inp), None, C.byref(out), None)
    assert rc == capi.ERR_HIP and not out.value
    q = capi.QMD()
    assert lib.mi355q_qmd_init(C.byref(plan), C.byref(q)) == 0
    h = C.c_void_p()
    assert lib.mi355q_result_create(C.byref(q), 0, None, C.byref(h)) in (capi.ERR_HIP, 2)  # 2 = ERR_OUT_OF_GPU_MEM
    assert not h.value
