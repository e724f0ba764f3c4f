#!/usr/bin/env python
"""Boundary-value fuzz: columns drawn from the extremes of their types (the inline NULL sentinels
included, in NOT NULL columns too), every aggregate kind, grouped and not, row-wise and columnar —
the product's row logic (host emulation) against the oracle, bit for bit.  64-bit values stay small
enough that no SUM wraps.  usage: boundary_fuzz.py <seed> <iterations>"""
import sys

import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heavydb_amd import capi
from heavydb_amd.executor import ExpressionRange as V, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
from oracle import oracle
from tests.cases import Case, col_range
from tests.helpers import compare_buffers, qmd_equal
from tests.test_rowlogic_emu import _emu_execute
NP = {capi.INT8: np.int8, capi.INT16: np.int16, capi.INT32: np.int32, capi.INT64: np.int64}


def run(seed, iters, engine=None):
    engine = engine or _emu_execute
    rng = np.random.default_rng(seed)
    tally = {}
    for it in range(iters):
        n = int(rng.integers(1, 200))
        cols, descs = [], []
        key = rng.integers(0, 6, n).astype(np.int32)
        cols.append(key); descs.append(InputColDescriptor(capi.INT32, False, V(True, 0, 5)))
        for c in range(int(rng.integers(1, 4))):
            t = [capi.INT8, capi.INT16, capi.INT32, capi.INT64][int(rng.integers(0, 4))]
            info = np.iinfo(NP[t])
            nullable = bool(rng.integers(0, 2))
            if t == capi.INT64:  # keep 64-bit sums away from wrap-around (order-dependent in the reference too)
                pool = np.array(([info.min] if nullable else []) + [-2**50, -2**31 - 1, -2, -1, 0, 1, 2, 2**31, 2**50], dtype=np.int64)
            else:
                pool = np.array([info.min, info.min + 1, info.min + 2, -2, -1, 0, 1, 2, info.max - 2, info.max - 1, info.max], dtype=np.int64)
            a = pool[rng.integers(0, len(pool), n)].astype(NP[t])
            valid_range = bool(rng.integers(0, 2))
            r = col_range([a], t, nullable) if valid_range else V(False)
            cols.append(a); descs.append(InputColDescriptor(t, nullable, r))
        targets = []
        for _ in range(int(rng.integers(1, 5))):
            k = int(rng.integers(0, 7))
            col = int(rng.integers(1, len(cols)))
            if k == 0: targets.append(TargetExpr(capi.COUNT))
            elif k == 1: targets.append(TargetExpr(capi.COUNT, col))
            elif k == 6: targets.append(TargetExpr(capi.SUM_IF, col, cond=Qual(int(rng.integers(1, len(cols))), capi.GT, 0)))
            else: targets.append(TargetExpr([capi.SUM, capi.AVG, capi.MIN, capi.MAX][k - 2], col))
        grouped = bool(rng.integers(0, 3))
        if grouped and rng.integers(0, 2): targets = [TargetExpr(capi.PROJECT_KEY)] + targets
        quals = [Qual(int(rng.integers(1, len(cols))), [capi.LT, capi.GE, capi.NE, capi.EQ][int(rng.integers(0, 4))], int(rng.choice([-1, 0, 1, 127, -128, 2**31 - 1])))] if rng.integers(0, 2) else []
        ra = RelAlgExecutionUnit(descs, targets, quals, [0] if grouped else [], output_columnar_hint=int(rng.integers(0, 2)))
        cut = n // 2
        case = Case("b", ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]])
        plan = ra.to_plan()
        try:
            q, want, code = oracle.execute(plan, case.frags, n_threads=2)
        except capi.Mi355qError:
            tally["rejected"] = tally.get("rejected", 0) + 1; continue
        eq, got, ecode = engine(case, plan, None)
        assert (code == 0) == (ecode == 0), (seed, it, code, ecode)
        if code: tally["err"] = tally.get("err", 0) + 1; continue
        qmd_equal(q, eq)
        if q.output_columnar:
            from tests.helpers import columnar_to_rows, rowwise_qmd
            compare_buffers(rowwise_qmd(q), columnar_to_rows(q, want), columnar_to_rows(q, got), 1e-9)
        else:
            compare_buffers(q, want, got, 1e-9)
        tally["ok"] = tally.get("ok", 0) + 1
        # SQLite as the third opinion — where the reference follows SQL: grouped steps, and non-grouped
        # ones unless a NOT NULL column holds its own NULL pattern (non-grouped aggregates skip it)
        sentinel_in_notnull = any((not d.nullable) and (np.asarray(c).astype(np.int64) == np.iinfo(NP[d.type]).min).any()
                                  for d, c in zip(descs[1:], cols[1:]))
        if grouped or not sentinel_in_notnull:
            from tests.test_sqlite_semantics import _check_case
            case.ra.output_columnar_hint = 0
            r = _check_case(oracle, case)
            assert r in ("ok", "keyless-null-aware"), r
            tally["sql_" + r] = tally.get("sql_" + r, 0) + 1
        if engine is _emu_execute:
            # ResultSetStorage::reduce of the two per-fragment buffers: oracle vs the product's reduce code
            import ctypes as C
            from tests.helpers import columnar_to_rows, emu_lib, rowwise_qmd
            _, a, ca = oracle.execute(plan, case.frags[:1])
            _, b, cb = oracle.execute(plan, case.frags[1:])
            if ca == 0 and cb == 0:
                r_o, r_e = a.copy(), a.copy()
                co = oracle.reduce(q, r_o, b)
                ce = emu_lib().emu_reduce(C.byref(q), r_e.ctypes.data, b.ctypes.data, q.entry_count)
                assert (co == 0) == (ce == 0), (seed, it, co, ce)
                if co == 0:
                    if q.output_columnar:
                        compare_buffers(rowwise_qmd(q), columnar_to_rows(q, r_o), columnar_to_rows(q, r_e), 1e-9)
                    else:
                        compare_buffers(q, r_o, r_e, 1e-9)
                    tally["reduce_ok"] = tally.get("reduce_ok", 0) + 1
    return tally


def run_keys(seed, iters, engine=None):
    engine = engine or _emu_execute
    """group keys at the extremes of their types (EMPTY_KEY_32/64 neighbours, the int32 boundary for
    8-byte keys), 1-3 group columns, perfect and baseline layouts"""
    from tests.helpers import columnar_to_rows, rowwise_qmd
    rng = np.random.default_rng(seed); tally = {}
    for it in range(iters):
        n = int(rng.integers(1, 120))
        ng = int(rng.integers(1, 4))
        cols, descs = [], []
        for g in range(ng):
            t = [capi.INT8, capi.INT16, capi.INT32, capi.INT64][int(rng.integers(0, 4))]
            info = np.iinfo(NP[t])
            nullable = bool(rng.integers(0, 2))
            hi = [info.max - 3, info.max - 2] + ([info.max - 1, info.max] if t != capi.INT64 else [])  # INT64_MAX - 1: the product's lock value (DESIGN)
            lo = ([info.min] if nullable or t != capi.INT64 else []) + [info.min + 1, info.min + 2]
            near32 = [2**31 - 3, 2**31 - 2, 2**31 - 1, 2**31, -2**31, -2**31 + 1, -2**31 - 1] if t == capi.INT64 else []
            pool = np.array(lo + [-1, 0, 1] + hi + near32, dtype=np.int64)
            pick = pool[rng.integers(0, len(pool), int(rng.integers(1, 5)))]       # few distinct values
            a = pick[rng.integers(0, len(pick), n)].astype(NP[t])
            rk = int(rng.integers(0, 3))
            r = V(False) if rk == 0 else col_range([a], t, nullable)
            cols.append(a); descs.append(InputColDescriptor(t, nullable, r))
        val = rng.integers(-100, 100, n).astype(np.int64)
        cols.append(val); descs.append(InputColDescriptor(capi.INT64, False, V(True, -100, 99)))
        targets = [TargetExpr(capi.PROJECT_KEY, g) for g in range(ng) if rng.integers(0, 2)] + [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, ng)]
        ra = RelAlgExecutionUnit(descs, targets, [], list(range(ng)), max_groups_buffer_entry_guess=int(rng.choice([16, 64, 2048])),
                                 output_columnar_hint=int(rng.integers(0, 2)), num_tuples=n)
        cut = n // 2
        case = Case("k", ra, [[c[:cut] for c in cols], [c[cut:] for c in cols]])
        plan = ra.to_plan()
        try:
            q, want, code = oracle.execute(plan, case.frags, n_threads=2)
        except capi.Mi355qError as e:
            from tests.helpers import emu_lib
            import ctypes as C
            assert emu_lib().emu_qmd_init(C.byref(plan), C.byref(capi.QMD())) != 0
            tally["rejected"] = tally.get("rejected", 0) + 1; continue
        eq, got, ecode = engine(case, plan, None)
        assert (code == 0) == (ecode == 0), (seed, it, code, ecode)
        if code: tally["err"] = tally.get("err", 0) + 1; continue
        qmd_equal(q, eq)
        if q.output_columnar: compare_buffers(rowwise_qmd(q), columnar_to_rows(q, want), columnar_to_rows(q, got), 1e-9)
        else: compare_buffers(q, want, got, 1e-9)
        k = "ok_%d_w%d" % (q.desc_type, q.key_width)
        tally[k] = tally.get(k, 0) + 1
    return tally


def run_joins(seed, iters, engine=None):
    engine = engine or _emu_execute
    """join keys at the extremes of their types on both sides, NULL keys, duplicates, empty inner
    tables, all four table layouts, INNER / LEFT: product row logic vs oracle vs SQLite"""
    from tests.test_rowlogic_emu import _oracle_join
    from tests.test_sqlite_semantics import _check_case
    NPJ = {capi.INT16: np.int16, capi.INT32: np.int32, capi.INT64: np.int64}
    rng = np.random.default_rng(seed); tally = {}
    for it in range(iters):
        t = [capi.INT16, capi.INT32, capi.INT64][int(rng.integers(0, 3))]
        info = np.iinfo(NPJ[t])
        wide = bool(rng.integers(0, 2))
        if wide:   # extremes of the type: only a keyed table can hold them
            pool = np.array([info.min + 1, info.min + 2, -1, 0, 1, info.max - 3, info.max - 2], dtype=np.int64)
        else:
            base = int(rng.choice([0, -5, info.max - 40, info.min + 2]))
            pool = base + np.arange(0, 12, dtype=np.int64)
        m = int(rng.integers(0, 10))
        dim = pool[rng.integers(0, len(pool), m)] if rng.integers(0, 2) else rng.permutation(pool)[:m]
        dnull = bool(rng.integers(0, 2))
        dim = dim.astype(NPJ[t])
        m = len(dim)
        if dnull and m: dim[rng.random(m) < 0.2] = info.min
        n = int(rng.integers(1, 80))
        fact = np.concatenate([pool, [info.min, info.max if t != capi.INT64 else info.max - 2]])[rng.integers(0, len(pool) + 2, n)].astype(NPJ[t])
        fnull = bool(rng.integers(0, 2))
        if not fnull: fact[fact == info.min] = pool[0]
        v = rng.integers(-50, 50, n).astype(np.int64)
        g = rng.integers(0, 3, n).astype(np.int32)
        w = rng.integers(-9, 9, m).astype(np.int64)
        fd = [InputColDescriptor(t, fnull, col_range([fact], t, fnull)), InputColDescriptor(capi.INT64, False, V(True, -50, 49)),
              InputColDescriptor(capi.INT32, False, V(True, 0, 2))]
        idesc = [InputColDescriptor(t, dnull, col_range([dim], t, dnull)), InputColDescriptor(capi.INT64, False, V(True, -9, 8))]
        grouped = bool(rng.integers(0, 2))
        targets = ([TargetExpr(capi.PROJECT_KEY)] if grouped else []) + [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1), TargetExpr(capi.SUM, 1, 1), TargetExpr(capi.COUNT, 0, 1)]
        ra = RelAlgExecutionUnit(fd, targets, [], [2] if grouped else [], inner_col_descs=idesc, join_outer_col=0, join_kind=int(rng.integers(0, 2)))
        dr = col_range([dim], t, dnull)
        if dr.min > dr.max: dr = V(True, 0, 0)
        case = Case("bj", ra, [[fact, v, g]], [dim, w], dim, t, dr, bool(rng.integers(0, 2)) or wide, join_one_to_many=1, join_key_nullable=dnull)
        plan = ra.to_plan()
        try:
            oj = _oracle_join(oracle, case)
        except capi.Mi355qError as e:
            tally["build_" + str(e.code)] = tally.get("build_" + str(e.code), 0) + 1; continue
        q, want, code = oracle.execute(plan, case.frags, case.inner, oj)
        eq, got, ecode = engine(case, plan, oj)
        assert code == 0 and ecode == 0, (seed, it, code, ecode)
        qmd_equal(q, eq); compare_buffers(q, want, got, 1e-9)
        r = _check_case(oracle, case)
        k = f"{r}_ht{oj.info()['hash_type']}"
        tally[k] = tally.get(k, 0) + 1
    return tally


def run_fp(seed, iters, with_nan=False, engine=None):
    """MIN / MAX / COUNT over FLOAT and DOUBLE columns of +-inf, +-0.0, the NULL sentinels (FLT_MIN / DBL_MIN),
    +-MAX and eps, bit for bit.  NaN is left out by default: the reference's MIN / MAX are comparison based,
    so their result with NaN inputs depends on the order of the rows (in the reference as well)."""
    rng = np.random.default_rng(seed); tally = {}
    for it in range(iters):
        n = int(rng.integers(1, 60))
        key = rng.integers(0, 4, n).astype(np.int32)
        f64 = bool(rng.integers(0, 2))
        dt = np.float64 if f64 else np.float32
        fi = np.finfo(dt)
        pool = [np.inf, -np.inf, -0.0, 0.0, fi.tiny, fi.max, -fi.max, 1.5, -2.25, fi.eps] + ([np.nan] if with_nan else [])
        a = np.array(pool, dtype=dt)[rng.integers(0, len(pool), n)]
        nullable = bool(rng.integers(0, 2))
        d = [InputColDescriptor(capi.INT32, False, V(True, 0, 3)), InputColDescriptor(capi.DOUBLE if f64 else capi.FLOAT, nullable, V(False))]
        grouped = bool(rng.integers(0, 2))
        tg = [TargetExpr([capi.MIN, capi.MAX, capi.COUNT][int(rng.integers(0, 3))], 1) for _ in range(int(rng.integers(1, 4)))]
        ra = RelAlgExecutionUnit(d, tg, [], [0] if grouped else [])
        case = Case("fp", ra, [[key, a]])
        plan = ra.to_plan()
        q, want, code = oracle.execute(plan, case.frags)
        eq, got, ecode = (engine or _emu_execute)(case, plan, None)
        assert code == 0 and ecode == 0
        qmd_equal(q, eq)
        if engine is not None:
            # concurrent execution: +0.0 and -0.0 compare equal, so which one a MIN / MAX keeps depends
            # on the order of the rows (std::max keeps the first) — compared as values, not as bits
            compare_buffers(q, want, got, 0.0)
            tally["ok"] = tally.get("ok", 0) + 1
        elif not np.array_equal(want, got):
            tally["bad"] = tally.get("bad", 0) + 1
        else: tally["ok"] = tally.get("ok", 0) + 1
    return tally


def _enc_col(rng, n):
    from tests.test_sqlite_semantics import _decoded_values
    kind = int(rng.integers(0, 3))
    if kind == 0:      # kENCODING_FIXED: storage narrower than the logical type
        st = [capi.INT8, capi.INT16, capi.INT32][int(rng.integers(0, 3))]
        lg = [t for t in (capi.INT16, capi.INT32, capi.INT64) if t > st][int(rng.integers(0, 3 - [capi.INT8, capi.INT16, capi.INT32].index(st)))]
        enc = capi.ENC_FIXED
    elif kind == 1:    # dictionary ids: unsigned 1 / 2 bytes, signed 4 bytes
        st, lg, enc = [capi.INT8, capi.INT16, capi.INT32][int(rng.integers(0, 3))], 0, capi.ENC_DICT
    else:              # DATE in days, 2 or 4 bytes
        st, lg, enc = [capi.INT16, capi.INT32][int(rng.integers(0, 2))], 0, capi.ENC_DATE_IN_DAYS
    info = np.iinfo(NP[st])
    nullable = bool(rng.integers(0, 2))
    pool = [info.min + 1, info.min + 2, -2, -1, 0, 1, 2, info.max - 2, info.max - 1]
    if enc == capi.ENC_DATE_IN_DAYS and st == capi.INT32:
        pool = [-30000, -1, 0, 1, 18000, 18001, 40000]
    if not (enc == capi.ENC_DICT and st != capi.INT32): pool.append(info.max)   # unsigned ids: -1 is the NULL id (255 / 65535)
    if enc == capi.ENC_DICT and st != capi.INT32 and not nullable: pool = [p for p in pool if p != -1]
    if nullable: pool.append(info.min if not (enc == capi.ENC_DICT and st != capi.INT32) else -1)
    if enc == capi.ENC_DICT and st != capi.INT32 and not nullable: pass
    a = np.array(pool, dtype=np.int64)[rng.integers(0, len(pool), n)].astype(NP[st])
    d = InputColDescriptor(st, nullable, V(False), enc, lg)
    vals = [v for v in _decoded_values(d, a) if v is not None]
    has_null = any(v is None for v in _decoded_values(d, a))
    if vals and rng.integers(0, 2):
        d.range = V(True, min(vals), max(vals), has_null, bucket=86400 if enc == capi.ENC_DATE_IN_DAYS and rng.integers(0, 2) else 0)
    return d, a
def run_enc(seed, iters, engine=None):
    """kENCODING_FIXED / dictionary-id / DATE-in-days columns at their storage extremes (and NULL
    patterns), as group keys and aggregate arguments: product row logic vs oracle vs SQLite (decoded with
    numpy)"""
    from tests.test_sqlite_semantics import _check_case
    rng = np.random.default_rng(seed); tally = {}
    for it in range(iters):
        n = int(rng.integers(1, 100))
        descs, cols = [], []
        for _ in range(int(rng.integers(1, 4))):
            d, a = _enc_col(rng, n); descs.append(d); cols.append(a)
        grouped = bool(rng.integers(0, 2))
        group = [int(rng.integers(0, len(cols)))] if grouped else []
        tg = ([TargetExpr(capi.PROJECT_KEY)] if grouped else []) + [TargetExpr(capi.COUNT)]
        for _ in range(int(rng.integers(1, 4))):
            tg.append(TargetExpr([capi.MIN, capi.MAX, capi.SUM, capi.COUNT, capi.AVG][int(rng.integers(0, 5))], int(rng.integers(0, len(cols)))))
        ra = RelAlgExecutionUnit(descs, tg, [], group, max_groups_buffer_entry_guess=64)
        case = Case("enc", ra, [[c[:n // 2] for c in cols], [c[n // 2:] for c in cols]])
        plan = ra.to_plan()
        try:
            q, want, code = oracle.execute(plan, case.frags, n_threads=2)
        except capi.Mi355qError:
            tally["rejected"] = tally.get("rejected", 0) + 1; continue
        eq, got, ecode = (engine or _emu_execute)(case, plan, None)
        assert (code == 0) == (ecode == 0), (seed, it, code, ecode)
        if code: tally["err"] = tally.get("err", 0) + 1; continue
        qmd_equal(q, eq); compare_buffers(q, want, got, 1e-9)
        r = _check_case(oracle, case)
        tally[r] = tally.get(r, 0) + 1
    return tally


def run_fpkeys(seed, iters, engine=None):
    """floating-point group keys (DOUBLE / FLOAT, nullable or not, alone or next to integer keys): always
    the baseline layout with the value's double bit pattern as the key; product vs oracle vs SQLite.
    (No signed-zero pairs and no NaN: bit patterns differ where SQL values compare equal.)"""
    from tests.test_sqlite_semantics import _check_case
    rng = np.random.default_rng(seed); tally = {}
    for it in range(iters):
        n = int(rng.integers(1, 150))
        cols, descs, group = [], [], []
        for g in range(int(rng.integers(1, 4))):
            kind = int(rng.integers(0, 3))
            nullable = bool(rng.integers(0, 2))
            if kind == 2:
                a = rng.integers(-3, 4, n).astype(np.int32)
                if nullable: a[rng.random(n) < 0.2] = -2**31
                d = InputColDescriptor(capi.INT32, nullable, col_range([a], capi.INT32, nullable))
            else:
                dt = np.float64 if kind == 0 else np.float32
                fi = np.finfo(dt)
                pool = np.array([np.inf, -np.inf, 0.0, 1.5, -2.25, fi.max, -fi.max, fi.eps, 2 * fi.tiny, 1e10, -1e-10], dtype=dt)
                pick = pool[rng.integers(0, len(pool), int(rng.integers(1, 6)))]
                a = pick[rng.integers(0, len(pick), n)].astype(dt)
                if nullable: a[rng.random(n) < 0.2] = fi.tiny
                d = InputColDescriptor(capi.DOUBLE if kind == 0 else capi.FLOAT, nullable, V(False))
            cols.append(a); descs.append(d); group.append(g)
        val = rng.integers(-100, 100, n).astype(np.int64)
        cols.append(val); descs.append(InputColDescriptor(capi.INT64, False, V(True, -100, 99)))
        targets = [TargetExpr(capi.PROJECT_KEY, g) for g in group if rng.integers(0, 3)] + [TargetExpr(capi.COUNT), TargetExpr(capi.SUM, len(group))]
        ra = RelAlgExecutionUnit(descs, targets, [], group, max_groups_buffer_entry_guess=int(rng.choice([64, 1024])),
                                 output_columnar_hint=int(rng.integers(0, 2)), num_tuples=n)
        case = Case("fpk", ra, [[c[:n // 2] for c in cols], [c[n // 2:] for c in cols]])
        plan = ra.to_plan()
        q, want, code = oracle.execute(plan, case.frags, n_threads=2)
        eq, got, ecode = (engine or _emu_execute)(case, plan, None)
        assert (code == 0) == (ecode == 0), (seed, it, code, ecode)
        if code: tally["err"] = tally.get("err", 0) + 1; continue
        qmd_equal(q, eq)
        if q.output_columnar:
            from tests.helpers import columnar_to_rows, rowwise_qmd
            compare_buffers(rowwise_qmd(q), columnar_to_rows(q, want), columnar_to_rows(q, got), 1e-9)
        else:
            compare_buffers(q, want, got, 1e-9)
        case.ra.output_columnar_hint = 0
        r = _check_case(oracle, case)
        k = f"{r}_{q.desc_type}"
        tally[k] = tally.get(k, 0) + 1
    return tally


if __name__ == "__main__":
    print(sys.argv[1], run(int(sys.argv[1]), int(sys.argv[2])), run_keys(int(sys.argv[1]), int(sys.argv[2])),
          run_joins(int(sys.argv[1]), int(sys.argv[2])), run_fp(int(sys.argv[1]), int(sys.argv[2])),
          run_enc(int(sys.argv[1]), int(sys.argv[2])), run_fpkeys(int(sys.argv[1]), int(sys.argv[2])))
