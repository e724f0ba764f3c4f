#!/usr/bin/env python
"""Filters that are BOOLEAN expressions, at benchmark size — one JSON line per shape:

    SELECT g, COUNT(*), SUM(v) FROM t WHERE <filter> GROUP BY g        (g: 1 000 groups, v INT32, a / b INT32, c nullable INT32)

    plain           a < K AND b > L                          two plan quals, no projection: the baseline of the comparison
    and_in_or       (a < K AND b > L) OR c IS NULL           one BOOLEAN expression `= 1` (k_project writes an INT8 column)
    not_or          NOT (a < K OR b > L)
    guarded_div     b <> 0 AND a / b > 3                     the deferred qual: a short-circuit AND inside the expression
    composed        three bands ORed, as three expressions    a root that reads the values of two earlier expressions
    sum_gt          a + b > 1 000 000                        round 6: program atoms (regprog.h) — arithmetic over two columns,
    col_lt_col      a < b                                    a comparison of two columns,
    affine          a * 3 - 7 <= 1 500 000                   a chain over one column

over --rows rows (device-generated, 32 M-row fragments).  Fraction = the columns a step reads (4 B each) / time / 8 TB/s; the
expression steps additionally write and re-read 1 B/row.  NOT RUN on the device yet: written in the last (GPU-less) session of
round 4 so that the next round's first GPU call can put a number on k_project for these shapes
(python tools/bool_filter_bench.py --rows 1e9 > gpurun_out/bool_filter_1b.jsonl)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shapes(capi, Expr, Qual):
    I32 = capi.INT32
    C, L = Expr.col, lambda v: Expr.lit(I32, v)
    a_lt = C(2).cmp(capi.EX_LT, L(500_000))
    b_gt = C(3).cmp(capi.EX_GT, L(250_000))
    band = lambda c, lo, hi: C(c).cmp(capi.EX_GT, L(lo)).logical(capi.EX_AND, C(c).cmp(capi.EX_LT, L(hi)))
    nc = 5
    return [
        ("plain", [], [Qual(2, capi.LT, 500_000), Qual(3, capi.GT, 250_000)], [0, 1, 2, 3]),
        ("and_in_or", [a_lt.logical(capi.EX_AND, b_gt).logical(capi.EX_OR, C(4).is_null())], [Qual(nc, capi.EQ, 1)], [0, 1, 2, 3, 4]),
        ("not_or", [a_lt.logical(capi.EX_OR, b_gt).logical_not()], [Qual(nc, capi.EQ, 1)], [0, 1, 2, 3]),
        ("guarded_div", [C(3).cmp(capi.EX_NE, L(0)).logical(capi.EX_AND, C(2).div(C(3), I32).cmp(capi.EX_GT, L(3)), True)],
         [Qual(nc, capi.EQ, 1)], [0, 1, 2, 3]),
        ("composed", [band(2, 100_000, 200_000), band(3, 300_000, 400_000),
                      band(4, 500_000, 600_000).logical(capi.EX_OR, C(nc)).logical(capi.EX_OR, C(nc + 1))], [Qual(nc + 2, capi.EQ, 1)],
         [0, 1, 2, 3, 4]),
        # round 6: leaves with arithmetic / two columns (program atoms, regprog.h)
        ("sum_gt", [C(2).add(C(3), I32).cmp(capi.EX_GT, L(1_000_000))], [Qual(nc, capi.EQ, 1)], [0, 1, 2, 3]),
        ("col_lt_col", [C(2).cmp(capi.EX_LT, C(3))], [Qual(nc, capi.EQ, 1)], [0, 1, 2, 3]),
        ("affine", [C(2).mul(L(3), I32).sub(L(7), I32).cmp(capi.EX_LE, L(1_500_000))], [Qual(nc, capi.EQ, 1)], [0, 1, 2]),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--verify-rows", type=float, default=0, help="also check every shape against the oracle on a table of this many rows")
    ap.add_argument("--count-only", action="store_true", help="SELECT g, COUNT(*) ...: no value column (the NV = 0 typed members)")
    ap.add_argument("--generic-member", action="store_true", help="MI355Q_OPT_LDS_GENERIC_MEMBER: the run-time-role member of k_groupby_lds")
    ap.add_argument("--only", default="", help="comma-separated shape names")
    ap.add_argument("--big-key", action="store_true", help="g is an INT64 key with 10 M values (baseline hash: the partitioned family, the headline's) instead of 1 000 INT32 groups")
    ap.add_argument("--prepass", action="store_true", help="MI355Q_OPT_FILTER_PREPASS: program atoms through the row-mask pre-pass even where the typed member evaluates them itself")
    ap.add_argument("--interpreted", action="store_true", help="MI355Q_OPT_NO_COMPILED_FILTER: every expression through k_project")
    args = ap.parse_args()
    import numpy as np
    import torch
    from heavydb_amd import capi
    from heavydb_amd.executor import (Executor, Expr, ExpressionRange, FetchResult, InputColDescriptor, Qual, RelAlgExecutionUnit,
                                      TargetExpr, generate_column)
    capi.load_library()
    n, frag = int(args.rows), 32_000_000
    # 0 g: 1 000 groups; 1 v; 2 a, 3 b: uniform [0, 1 M); 4 c: uniform [0, 1 M), stated nullable (no NULLs generated)
    gens = [(capi.GEN_I32_MOD, 21, 1000), (capi.GEN_I32_MOD, 22, 1_000_000), (capi.GEN_I32_MOD, 23, 1_000_000),
            (capi.GEN_I32_MOD, 24, 1_000_000), (capi.GEN_I32_MOD, 25, 1_000_000)]
    if args.big_key:
        gens[0] = (capi.GEN_I64_MOD_MUL, 21, 10_000_000)
    cols = [torch.empty(n, dtype=torch.int64 if (args.big_key and i == 0) else torch.int32, device="cuda:0") for i, _ in enumerate(gens)]
    bufs, rows, off = [], [], 0
    while off < n:
        k = min(frag, n - off)
        for t, (kind, seed, mod) in zip(cols, gens):
            generate_column(int(t.data_ptr()) + off * t.element_size(), k, kind, seed, mod, 1_000_003 if kind == capi.GEN_I64_MOD_MUL else 0,
                            7 if kind == capi.GEN_I64_MOD_MUL else 0, 0.0, 0, off, 0)
        bufs.append([int(t.data_ptr()) + off * t.element_size() for t in cols])
        rows.append(k)
        off += k
    torch.cuda.synchronize()
    descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 7, 9_999_999 * 1_000_003 + 7)) if args.big_key else
             InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, 999))] + \
            [InputColDescriptor(capi.INT32, i == 4, ExpressionRange(True, 0, 999_999, False)) for i in range(1, 5)]
    fr = FetchResult(bufs, rows, keepalive=cols)
    ex = Executor(0)
    flags = (capi.OPT_NO_COMPILED_FILTER if args.interpreted else 0) | (capi.OPT_FILTER_PREPASS if args.prepass else 0) | (capi.OPT_LDS_GENERIC_MEMBER if args.generic_member else 0)
    for name, exprs, quals, reads in shapes(capi, Expr, Qual):
        if args.only and name not in args.only.split(','):
            continue
        xs = [e.with_range(ExpressionRange(True, 0, 1, True)) for e in exprs]
        targets = [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT)] + ([] if args.count_only else [TargetExpr(capi.SUM, 1)])
        ra = RelAlgExecutionUnit(descs, targets, quals, [0],
                                 exprs=xs, num_tuples=n, **({"max_groups_buffer_entry_guess": 20_000_000} if args.big_key else {}))
        best, rs = None, None
        for _ in range(args.steps):
            rs = ex.executeWorkUnit(ra, fr, allow_retry=False, flags=flags)
            best = rs.report.total_ms if best is None else min(best, rs.report.total_ms)
        if args.count_only:
            reads = [c for c in reads if c != 1]   # (no target reads the value column)
        line = {"shape": name, "rows": n, "interpreted": bool(args.interpreted), "prepass": bool(args.prepass), "count_only": bool(args.count_only), "generic_member": bool(args.generic_member), "route": ex.explain(ra, rows, flags=flags), "kernel": rs.report.kernel_name.decode(), "ms": round(best, 3),
                "bytes_per_row": 4 * len(reads) + (4 if args.big_key else 0), "big_key": bool(args.big_key),
                "whole_step_frac": round((4 * len(reads) + (4 if args.big_key else 0)) * n / (best * 1e-3) / 8e12, 4),
                "groups": rs.rowCount()}
        if args.verify_rows:
            from oracle import oracle as orc
            from tests.helpers import compare_buffers
            m = min(int(args.verify_rows), rows[0])
            host = [t[:m].cpu().numpy() for t in cols]
            small = FetchResult([[int(t.data_ptr()) for t in cols]], [m], keepalive=cols)
            got = ex.executeWorkUnit(ra, small, allow_retry=False, flags=flags)
            q, want, code = orc.execute(ra.to_plan(), [host], n_threads=8)
            assert code == 0
            compare_buffers(q, want, got.getStorage())
            line["verified_rows"] = m
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
