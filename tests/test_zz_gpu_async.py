"""mi355q_execute_async / mi355q_wait (the stream-ordered step) and mi355q_reserve_workspace on the device."""
import ctypes as C

import numpy as np
import pytest

from heavydb_amd import capi
from tests.helpers import compare_buffers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    capi.load_library()
    return torch


def _unit(torch, n=12_000_004, n_keys=150_000, guess=None):
    from heavydb_amd.executor import (ExpressionRange, FetchResult, InputColDescriptor, Qual, RelAlgExecutionUnit,
                                      TargetExpr, generate_column)
    key = torch.empty(n, dtype=torch.int64, device="cuda:0")
    val = torch.empty(n, dtype=torch.float64, device="cuda:0")
    fil = torch.empty(n, dtype=torch.int32, device="cuda:0")
    generate_column(int(key.data_ptr()), n, capi.GEN_I64_MOD_MUL, 0xA5A50000, n_keys, 1000003, 7)
    generate_column(int(val.data_ptr()), n, capi.GEN_F64_UNIT, 0xA5A50001, a_f=1000.0)
    generate_column(int(fil.data_ptr()), n, capi.GEN_I32_UNIFORM31, 0xA5A50002)
    torch.cuda.synchronize()
    ra = RelAlgExecutionUnit(
        [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, (n_keys - 1) * 1000003 + 7)),
         InputColDescriptor(capi.DOUBLE, False, ExpressionRange(True, 0, 0, False, 0.0, 1000.0)),
         InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 2**31 - 1))],
        [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.AVG, 1)],
        [Qual(2, capi.LT, 2**30)], [0], max_groups_buffer_entry_guess=guess or 2 * n_keys)
    half = (n // 2) // 4 * 4
    fr = FetchResult([[int(key.data_ptr()), int(val.data_ptr()), int(fil.data_ptr())],
                      [int(key.data_ptr()) + half * 8, int(val.data_ptr()) + half * 8, int(fil.data_ptr()) + half * 4]],
                     [half, n - half], keepalive=[key, val, fil])
    return ra, fr


def test_async_step_equals_the_synchronous_one(torch_cuda):
    from heavydb_amd.executor import Executor
    ra, fr = _unit(torch_cuda)
    ex = Executor(0)
    sync = ex.executeWorkUnit(ra, fr, allow_retry=False)
    rs, pend = ex.executeWorkUnitAsync(ra, fr)
    assert rs is not None and pend.handle is not None
    got = pend.wait()
    assert got is rs and rs.report.kernel_name.decode() == sync.report.kernel_name.decode() == "k_part_scatter"
    assert rs.report.n_launches == sync.report.n_launches and rs.report.kernel_ms > 0
    q = rs.getQueryMemDesc()
    compare_buffers(q, sync.getStorage(), rs.getStorage(), 1e-9)
    assert pend.wait() is rs   # idempotent on the Python side (the handle is gone)


def test_a_second_call_finishes_the_step_in_flight(torch_cuda):
    """One step per device may be in flight: the next execute on the device drains it, and its status is
    still delivered by its own mi355q_wait."""
    from heavydb_amd.executor import Executor
    ra, fr = _unit(torch_cuda)
    ex = Executor(0)
    want = ex.executeWorkUnit(ra, fr, allow_retry=False).getStorage()
    rs1, p1 = ex.executeWorkUnitAsync(ra, fr)
    rs2, p2 = ex.executeWorkUnitAsync(ra, fr)       # drains p1 first
    third = ex.executeWorkUnit(ra, fr, allow_retry=False)   # drains p2
    q = third.getQueryMemDesc()
    p2.wait()
    p1.wait()
    for r in (rs1, rs2, third):
        compare_buffers(q, want, r.getStorage(), 1e-9)


def test_async_error_arrives_at_wait(torch_cuda):
    """A table too small for the groups: the step runs out of slots (a negative code / ERR_OUT_OF_SLOTS), which
    only the host can see — it is raised by wait(), and the result handle stays the caller's to free."""
    from heavydb_amd.executor import Executor
    ra, fr = _unit(torch_cuda, guess=40_000)
    ex = Executor(0)
    rs, pend = ex.executeWorkUnitAsync(ra, fr)
    with pytest.raises(capi.Mi355qError) as ei:
        pend.wait()
    assert ei.value.code < 0 or ei.value.code == capi.ERR_OUT_OF_SLOTS


def test_reserve_workspace(torch_cuda):
    from heavydb_amd.executor import Executor
    lib = capi.load_library()
    ra, fr = _unit(torch_cuda)
    ex = Executor(0)
    assert lib.mi355q_release_workspace(0) == 0
    free0, _ = torch_cuda.cuda.mem_get_info(0)
    got = ex.reserveWorkspace(ra, fr)
    free1, _ = torch_cuda.cuda.mem_get_info(0)
    assert got > 0 and free0 - free1 >= got // 2     # the partition scratch is really held now
    rs = ex.executeWorkUnit(ra, fr, allow_retry=False)
    free2, _ = torch_cuda.cuda.mem_get_info(0)
    assert rs.report.kernel_name.decode() == "k_part_scatter"
    assert free1 - free2 < got // 4 + (64 << 20)     # the step found its scratch in place (only the table is new)
    assert ex.reserveWorkspace(ra, fr) == got        # idempotent
