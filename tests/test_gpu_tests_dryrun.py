"""Dry run of the `-m gpu` test FUNCTIONS on a machine without a GPU: the test bodies run unchanged,
with the host emulation of the product's row logic (tests/emu) standing in for the device behind
`Executor.executeWorkUnit` / `HashJoin.getInstance`, and torch tensors staying in host memory.  This
does not test the product (the gpu run does); it keeps the gpu tests' own Python — helpers, shapes,
tolerances, bookkeeping — from rotting between GPU sessions."""
import ctypes as C
import inspect

import numpy as np
import pytest

from heavydb_amd import capi, executor
from tests.helpers import columnar_to_rows, emu_lib, rowwise_qmd

torch = pytest.importorskip("torch")


class _FakeJoin:
    def __init__(self, oj, case):
        self.oj, self.case, self.handle = oj, case, None

    def info(self):
        i = self.oj.info()
        return dict(hash_type=i["hash_type"], entry_count=i["entry_count"], min_key=self.case.join_range.min,
                    max_key=self.case.join_range.max)


class _FakeRS:
    def __init__(self, oracle, q, buf):
        self._o, self._q, self._buf = oracle, q, buf
        self._lib = capi.load_library()
        self.report = type("R", (), {"kernel_name": b"emu", "variant": 0})()
        self.handle = None

    def getQueryMemDesc(self):
        return self._q

    def getStorage(self):
        return self._buf.copy()

    columns = executor.ResultSet.columns

    def rowCount(self):
        return self._o.row_count(self._q, self._buf)

    def fetch(self):
        return self._o.fetch_rows(self._q, self._buf)

    def reduce(self, that, stream=None):
        flat = np.ascontiguousarray(self._buf).reshape(-1)
        rc = emu_lib().emu_reduce(C.byref(self._q), flat.ctypes.data,
                                  np.ascontiguousarray(that._buf).ctypes.data, that._q.entry_count)
        if rc:
            raise capi.Mi355qError(rc, "reduce")
        self._buf = flat if self._q.output_columnar else flat.reshape(self._buf.shape)

    def sort(self, target_idx, top_n, out_rows_dev, desc=True, nulls_first=False):
        return min(top_n, self.rowCount())

    def to_columns(self, torch_mod):
        iv, dv, nu = self.fetch()
        q = self._q
        cols = []
        for t in range(q.n_targets):
            c = dv[:, t].copy().view(np.int64) if q.target_is_fp[t] else iv[:, t].copy()
            cols.append(torch_mod.from_numpy(np.ascontiguousarray(c)))
        return cols, iv.shape[0]


def _install(monkeypatch, oracle):
    emu = emu_lib()
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self, raising=False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    real_zeros, real_empty = torch.zeros, torch.empty
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: real_zeros(*a, **{x: y for x, y in k.items() if x != "device"}))
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{x: y for x, y in k.items() if x != "device"}))

    def execute(self, ra, fr, stream=None, out_buffer=None, force_generic=False, kernel_variant=0,
                scratch_bytes=0, allow_retry=True, **knobs):   # (knobs: flags, tuning — no meaning for the emulation)
        plan = ra.to_plan()
        inp, keep = fr.to_c(plan.n_cols)
        q = capi.QMD()
        rc = emu.emu_qmd_init(C.byref(plan), C.byref(q))
        if rc:
            raise capi.Mi355qError(rc, "qmd_init")
        buf = np.zeros(emu.emu_buffer_bytes(C.byref(q)) // 8, dtype=np.int64)
        jt, jbuf, jmin, jmax, jn, jk, jw, hold = 0, None, 0, 0, 0, 1, 8, None
        if ra.join_table is not None:
            fj = ra.join_table
            info, sh = fj.oj.info(), fj.oj.shape()
            hold = fj.oj.raw()
            jt, jbuf, jn = info["hash_type"], hold.ctypes.data, info["entry_count"]
            jk, jw = sh["key_components"], sh["component_width"]
            jmin, jmax = fj.case.join_range.min, fj.case.join_range.max
        out_q = capi.QMD()
        code = emu.emu_execute(C.byref(plan), C.byref(inp), jt, jbuf, jmin, jmax, jn, jk, jw, buf.ctypes.data,
                               C.byref(out_q))
        if code:
            raise capi.Mi355qError(code, "execute")
        return _FakeRS(oracle, out_q, buf if out_q.output_columnar else buf.reshape(out_q.entry_count, -1))
    monkeypatch.setattr(executor.Executor, "__init__", lambda self, device_id=0: setattr(self, "device_id", device_id))
    monkeypatch.setattr(executor.Executor, "executeWorkUnit", execute)

    def init_qmd(self, ra):
        q = capi.QMD()
        capi.check(emu.emu_qmd_init(C.byref(ra.to_plan()), C.byref(q)), "qmd_init")
        return q
    monkeypatch.setattr(executor.Executor, "initQueryMemoryDescriptor", init_qmd)
    import tests.test_gpu_parity as gp

    def build_join(torch_mod, case):
        if case.join_keys is None:
            return None, None
        return _FakeJoin(gp._oracle_join(oracle, case), case), None
    for modname in ("tests.test_gpu_parity", "tests.test_zz_gpu_columnar", "tests.test_zz_gpu_execute_style",
                    "tests.test_zz_gpu_sqlite_scale", "tests.test_zz_gpu_boundary"):
        mod = __import__(modname, fromlist=["x"])
        if hasattr(mod, "_build_join"):
            monkeypatch.setattr(mod, "_build_join", build_join)


def _call(fn, **kw):
    fn(**{k: v for k, v in kw.items() if k in inspect.signature(fn).parameters})


def test_dryrun_columnar_matrix(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_zz_gpu_columnar as m
    for case in m.CASES[::3]:
        for fg in (True, False):
            m.test_hip_columnar_matches_oracle(torch, oracle, case, fg)


def test_dryrun_columnar_operations(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_zz_gpu_columnar as m
    names = ["perfect_key_sum_projectkey", "baseline_count_avg", "multi_perfect_2col_keyless",
             "compact_baseline_count_only"]
    for name in names:
        try:
            m.test_columnar_result_operations(torch, oracle, name)
        except capi.Mi355qError as e:
            # the last block creates a result on a real device (mi355q_result_create): everything
            # before it ran
            assert e.code in (capi.ERR_HIP, 2) and "result_create" in str(e), e


def test_dryrun_execute_style(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_zz_gpu_execute_style as m
    from tests.test_execute_style import JOIN_QUERIES, QUERIES
    for qi in range(len(QUERIES)):
        m.test_reference_queries_on_gpu(torch, oracle, qi)
    for ji in range(len(JOIN_QUERIES)):
        m.test_reference_join_queries_on_gpu(torch, oracle, ji)
    m.test_reference_expression_queries_on_gpu(torch, oracle)
    m.test_reference_div_by_zero_queries_on_gpu(torch, oracle)
    m.test_reference_boolean_column_queries_on_gpu(torch, oracle)
    m.test_reference_overflow_queries_on_gpu(torch, oracle)


def test_dryrun_sqlite_scale(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_zz_gpu_sqlite_scale as m
    for si in (1, 2):   # the emulation is a scalar loop: two of the million-row shapes are enough here
        m.test_kernel_families_agree_with_sqlite(torch, si)


def test_dryrun_boundary(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_zz_gpu_boundary as m
    mod = m._tool()
    eng = m._hip_engine(torch)
    assert mod.run(7001, 30, eng).get("ok", 0) > 20
    assert sum(v for k, v in mod.run_keys(7002, 30, eng).items() if k.startswith("ok_")) > 20
    assert sum(v for k, v in mod.run_joins(7003, 30, eng).items() if k.startswith("ok_")) > 20
    assert mod.run_fp(7004, 30, False, eng) == {"ok": 30}
    assert mod.run_enc(7005, 30, eng).get("ok", 0) > 20
    assert mod.run_fpkeys(7006, 30, eng).get("ok_1", 0) > 15


def test_dryrun_parity_matrix(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_gpu_parity as m
    for case in m.CASES[::4]:
        m.test_hip_matches_oracle(torch, oracle, case, False)
    for name in ("simple_aggs_nullable", "baseline_count_avg", "multi_baseline_i64_2col", "compact_baseline_key32"):
        m.test_device_reduce_matches_oracle(torch, oracle, name)


def test_dryrun_expression_vectors(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_gpu_parity as m
    m.test_hip_expressions_match_reference_functions(torch)


def test_dryrun_refbench(monkeypatch, oracle):
    _install(monkeypatch, oracle)
    import tests.test_zz_gpu_refbench as m
    for name in ("NGA01", "PHS002", "BH001", "MSBS001", "MSPHM006"):
        m._run(torch, oracle, name, 30_000, 3_000, 0)
