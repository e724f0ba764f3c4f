"""The C-ABI must answer malformed plans with an error code, never with a crash or an absurd
descriptor (the reference validates at the same boundary with CHECKs; a drop-in library cannot
abort its host).  Valid random plans (tests/test_plan_fuzz.py) get 1-3 hostile field mutations
and go through the REAL library's host-only entry points (mi355q_qmd_init,
mi355q_qmd_buffer_bytes, mi355q_join_key_shape) in a child process, so a fault shows up as a
non-zero exit."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes as C, sys
    import numpy as np
    sys.path.insert(0, %r)
    from heavydb_amd import capi
    from tests.test_plan_fuzz import _random_plan
    lib = capi.load_library()
    rng = np.random.default_rng(int(sys.argv[1]))
    HOSTILE32 = [-1, -2, 4, 5, 8, 9, 15, 16, 17, 31, 32, 33, 64, 127, 255, 1 << 20, -(1 << 31), (1 << 31) - 1]
    HOSTILE64 = [0, 1, -1, 2**63 - 1, -2**63, 2**62, -2**62, 2**32, 2**32 - 1, -2**32]
    def h32():
        return int(HOSTILE32[rng.integers(0, len(HOSTILE32))])
    def h64():
        return int(HOSTILE64[rng.integers(0, len(HOSTILE64))])
    def mutate(p):
        k = int(rng.integers(0, 17))
        i = int(rng.integers(0, capi.MAX_COLS))
        t = int(rng.integers(0, capi.MAX_TARGETS))
        g = int(rng.integers(0, capi.MAX_GROUP_COLS))
        if k == 0: p.n_cols = h32()
        elif k == 1: p.n_targets = h32()
        elif k == 2: p.n_group_cols = h32()
        elif k == 3: p.n_quals = h32()
        elif k == 4: p.group_cols[g] = h32()
        elif k == 5: p.targets[t].col = h32()
        elif k == 6: p.targets[t].agg = h32()
        elif k == 7: p.targets[t].table = h32()
        elif k == 8: p.targets[t].cond.col = h32()
        elif k == 9: p.cols[i].type = h32()
        elif k == 10: p.cols[i].encoding = h32(); p.cols[i].logical_type = h32()
        elif k == 11: p.col_ranges[i].min = h64(); p.col_ranges[i].max = h64(); p.col_ranges[i].valid = 1
        elif k == 12: p.col_ranges[i].bucket = h64()
        elif k == 13: p.max_groups_buffer_entry_guess = h64()
        elif k == 14: p.num_tuples = h64()
        elif k == 16: p.output_columnar_hint = int(rng.integers(-2, 5))
        else:
            p.join_outer_col = h32(); p.n_join_cols = h32(); p.n_inner_cols = h32(); p.join_kind = h32()
    ok = bad = 0
    for it in range(int(sys.argv[2])):
        p = _random_plan(rng).to_plan()
        for _ in range(int(rng.integers(1, 4))):
            mutate(p)
        q = capi.QMD()
        rc = lib.mi355q_qmd_init(C.byref(p), C.byref(q))
        if rc == 0:
            ok += 1
            assert 0 < q.row_size <= 8 * (capi.MAX_GROUP_COLS + capi.MAX_SLOTS), q.row_size
            assert q.entry_count > 0 and 0 <= q.slot_count <= capi.MAX_SLOTS and 0 <= q.n_targets <= capi.MAX_TARGETS
            assert q.slot_width in (4, 8) and q.key_bytes %% 8 == 0 and q.key_bytes <= 8 * capi.MAX_GROUP_COLS
            for j in range(q.n_targets):
                assert -1 <= q.target_slot[j] < q.slot_count
            if not q.output_columnar:
                assert lib.mi355q_qmd_buffer_bytes(C.byref(q)) == q.entry_count * q.row_size
            else:
                assert lib.mi355q_qmd_buffer_bytes(C.byref(q)) >= q.entry_count * q.slot_width * q.slot_count
                assert lib.mi355q_qmd_slot_col_offset(C.byref(q), q.slot_count) == -1
        else:
            bad += 1
            assert rc in (capi.ERR_INVALID_PLAN, capi.ERR_UNSUPPORTED), rc
    print(ok, bad)
""") % ROOT


def test_malformed_plans_are_rejected_not_crashed():
    env = dict(os.environ, PYTHONPATH=ROOT)
    tot_ok = tot_bad = 0
    for seed in (1, 2, 3):
        r = subprocess.run([sys.executable, "-c", CHILD, str(seed), "4000"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (seed, r.returncode, r.stdout[-500:], r.stderr[-2000:])
        ok, bad = (int(x) for x in r.stdout.split())
        tot_ok += ok
        tot_bad += bad
    # the mutations must land on both sides of the validation
    assert tot_ok > 500 and tot_bad > 500, (tot_ok, tot_bad)


CHILD_EXPR = textwrap.dedent("""
    import ctypes as C, sys
    import numpy as np
    sys.path.insert(0, %r)
    from heavydb_amd import capi
    from heavydb_amd.executor import Expr, ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit, TargetExpr
    from tests.test_expr import _random_bool, _random_expr, _stack_depth
    lib = capi.load_library()
    rng = np.random.default_rng(int(sys.argv[1]))
    types = [capi.INT8, capi.INT16, capi.INT32, capi.INT64, capi.DOUBLE, capi.FLOAT]
    descs = [InputColDescriptor(t, n, ExpressionRange(True, -100, 100, n, -100.0, 100.0)) for t in types for n in (True, False)]
    HOSTILE32 = [-1, -2, 0, 3, 4, 5, 8, 12, 13, 15, 16, 17, 21, 22, 31, 32, 64, 255, 1 << 20, -(1 << 31), (1 << 31) - 1]
    def h32():
        return int(HOSTILE32[rng.integers(0, len(HOSTILE32))])
    ok = bad = 0
    for it in range(int(sys.argv[2])):
        xs = []
        for k in range(int(rng.integers(1, min(capi.MAX_EXPRS, capi.MAX_COLS - len(descs)) + 1))):
            for attempt in range(20):
                e = _random_bool(rng, descs, int(rng.integers(1, 4))) if rng.integers(0, 3) == 0 else \\
                    _random_expr(rng, descs, int(rng.choice(types)), int(rng.integers(1, 4)))
                if e is not None and len(e.nodes) <= capi.MAX_EXPR_NODES and _stack_depth(e) <= capi.MAX_EXPR_STACK:
                    break
            else:
                e = Expr.col(0)
            if xs and rng.integers(0, 3) == 0:
                # the value of an EARLIER expression (a node's hostile `arg` below also names this one or a later one: refused)
                j = int(rng.integers(0, len(xs)))
                e = Expr.col(len(descs) + j).is_null()
            xs.append(e.with_range(ExpressionRange(True, -1000, 1000, True, -1000.0, 1000.0)))
        nc = len(descs)
        ra = RelAlgExecutionUnit(descs, [TargetExpr(capi.COUNT), TargetExpr(capi.MIN, nc + int(rng.integers(0, len(xs))))],
                                 [Qual(nc + int(rng.integers(0, len(xs))), capi.IS_NOT_NULL, 0)], [], exprs=xs)
        p = ra.to_plan()
        valid = lib.mi355q_qmd_init(C.byref(p), C.byref(capi.QMD())) == 0
        n0 = p.n_exprs
        for _ in range(int(rng.integers(0, 3))):
            k = int(rng.integers(0, n0))
            x = p.exprs[k]
            i = int(rng.integers(0, capi.MAX_EXPR_NODES))
            m = int(rng.integers(0, 7))
            if m == 0: x.nodes[i].op = h32()
            elif m == 1: x.nodes[i].type = h32()
            elif m == 2: x.nodes[i].arg = h32()
            elif m == 3: x.nodes[i].reserved = h32()
            elif m == 4: x.n_nodes = h32()
            elif m == 5: p.n_exprs = h32()
            else: p.quals[0].col = h32()
        q = capi.QMD()
        rc = lib.mi355q_qmd_init(C.byref(p), C.byref(q))
        if rc == 0:
            ok += 1
            assert 0 < q.row_size <= 8 * (capi.MAX_GROUP_COLS + capi.MAX_SLOTS) and q.n_targets == 2
        else:
            bad += 1
            assert rc in (capi.ERR_INVALID_PLAN, capi.ERR_UNSUPPORTED), rc
    print(ok, bad)
""") % ROOT


def test_malformed_expression_programs_are_rejected_not_crashed():
    """The same for the projected-expression programs: random well-typed programs over every micro-op (values of earlier
    expressions included), then 0-2 hostile mutations of a node's op / type / arg / reserved, of a program's length, of the
    expression count or of the qual's column, through mi355q_qmd_init (which lowers the programs) in a child process."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    tot_ok = tot_bad = 0
    for seed in (11, 12):
        r = subprocess.run([sys.executable, "-c", CHILD_EXPR, str(seed), "3000"], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (seed, r.returncode, r.stdout[-500:], r.stderr[-2000:])
        ok, bad = (int(x) for x in r.stdout.split())
        tot_ok += ok
        tot_bad += bad
    assert tot_ok > 1000 and tot_bad > 1000, (tot_ok, tot_bad)
