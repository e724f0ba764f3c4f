"""GPU leg of tools/boundary_fuzz.py: columns, group keys and join keys at the extremes of their
types (inline NULL sentinels, EMPTY_KEY neighbours, the int32 boundary) through the HIP library —
device atomics on extreme values, device-built join tables — against the oracle (and, for joins,
SQLite)."""
import importlib.util
import os

import pytest

from heavydb_amd import capi
from tests.test_gpu_parity import _build_join, _fetch_result, _upload, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu


def _tool():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "boundary_fuzz.py")
    spec = importlib.util.spec_from_file_location("boundary_fuzz", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _hip_engine(torch):
    from heavydb_amd.executor import Executor

    def engine(case, plan, oj):
        frag_t, inner_t = _upload(torch, case)
        hj, keep = _build_join(torch, case)
        case.ra.join_table = hj
        try:
            rs = Executor(0).executeWorkUnit(case.ra, _fetch_result(case, frag_t, inner_t), allow_retry=False)
            return rs.getQueryMemDesc(), rs.getStorage(), 0
        except capi.Mi355qError as e:
            return None, None, e.code
        finally:
            case.ra.join_table = None
    return engine


def test_boundary_values_on_gpu(torch_cuda, oracle):
    mod = _tool()
    t = mod.run(7001, 120, _hip_engine(torch_cuda))
    assert t.get("ok", 0) > 100, t


def test_boundary_keys_on_gpu(torch_cuda, oracle):
    mod = _tool()
    t = mod.run_keys(7002, 120, _hip_engine(torch_cuda))
    assert sum(v for k, v in t.items() if k.startswith("ok_")) > 100, t


def test_boundary_joins_on_gpu(torch_cuda, oracle):
    mod = _tool()
    t = mod.run_joins(7003, 120, _hip_engine(torch_cuda))
    assert sum(v for k, v in t.items() if k.startswith("ok_")) > 100, t


def test_boundary_floats_on_gpu(torch_cuda, oracle):
    mod = _tool()
    t = mod.run_fp(7004, 150, False, _hip_engine(torch_cuda))
    assert t == {"ok": 150}, t


def test_boundary_encoded_columns_on_gpu(torch_cuda, oracle):
    mod = _tool()
    t = mod.run_enc(7005, 120, _hip_engine(torch_cuda))
    assert t.get("ok", 0) > 90 and set(t) <= {"ok", "rejected", "keyless-null-aware"}, t


def test_floating_point_group_keys_on_gpu(torch_cuda, oracle):
    mod = _tool()
    t = mod.run_fpkeys(7006, 120, _hip_engine(torch_cuda))
    assert t.get("ok_1", 0) > 70 and set(t) <= {"ok_0", "ok_1", "err", "keyless-null-aware_0"}, t
