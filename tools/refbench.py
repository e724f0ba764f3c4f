#!/usr/bin/env python
"""refbench.py — the reference's OWN synthetic benchmark through the library.

Schema and data: Benchmarks/synthetic_benchmark/create_table.py:116-137 — eleven INT columns x10 .. x10m
(uniform in [1, N], declared nullable, no NULL generated) and three BIGINT columns x10k_s10k .. x1m_s10k (uniform in
[1, N] times 10000), default 4 fragments of 32 M rows.  Queries: every file under
Benchmarks/synthetic_benchmark/queries/{NonGroupedAgg,PerfectHashSingleCol,PerfectHashMultiCol,BaselineHash,Sort,
MultiStep}/*.sql, restated as the execution unit of the query's (first) aggregation step the way
RelAlgExecutor hands it to Executor::executeWorkUnit: group-by expressions, aggregate targets (the post-aggregation
arithmetic of the MultiStep queries — max(a) + max(b), sum(a) / sum(b), HAVING — runs on the handful of result rows
and is not part of the step), cast(x as double / float) keys and `x10 + 1` arguments as projected expressions, Sort
as the step plus ORDER BY cnt LIMIT 100 on the device (mi355q_result_topk).

One JSON line per query: kernel family, whole-step ms, rows/s, algorithmic bytes per row (the distinct columns the
query reads), whole-step fraction of the 8 TB/s roofline.

  python tools/refbench.py                  # 4 x 32 M rows (the reference's default size)
  python tools/refbench.py --rows 1e9 --steps 3 --only PHS,BH
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

INT_COLS = [("x10", 10), ("y10", 10), ("z10", 10), ("x100", 100), ("y100", 100), ("z100", 100), ("x1k", 1000),
            ("x10k", 10_000), ("x100k", 100_000), ("x1m", 1_000_000), ("x10m", 10_000_000)]
BIG_COLS = [("x10k_s10k", 10_000), ("x100k_s10k", 100_000), ("x1m_s10k", 1_000_000)]
STEP = 10_000
FRAGMENT_ROWS = 32_000_000


def schema(card_cap: int = 0):
    """card_cap > 0 bounds every column's cardinality (the small tables of the CPU tests)."""
    from heavydb_amd import capi
    from heavydb_amd.executor import ExpressionRange, InputColDescriptor
    names, descs, gens = [], [], []
    cap = lambda n: min(n, card_cap) if card_cap else n  # noqa: E731
    for i, (nm, n) in enumerate(INT_COLS):
        n = cap(n)
        names.append(nm)
        # declared nullable (create_table.py:42 nullable=True), chunk metadata: no NULLs, min 1, max N
        descs.append(InputColDescriptor(capi.INT32, True, ExpressionRange(True, 1, n, False)))
        gens.append((capi.GEN_I32_MOD, 0xBE9C0000 + i, n, 1, 0, 0.0))
    for i, (nm, n) in enumerate(BIG_COLS):
        names.append(nm)
        # a capped table draws fewer distinct values but keeps the column's declared range (chunk metadata min / max of the
        # real table): the stride-10 000 columns must stay baseline-hash keys, as in the benchmark — a capped RANGE of
        # 6 - 30 M would turn them into perfect-hash tables of that many entries
        descs.append(InputColDescriptor(capi.INT64, True, ExpressionRange(True, STEP, n * STEP, False)))
        gens.append((capi.GEN_I64_MOD_MUL, 0xBE9C0100 + i, cap(n), STEP, STEP, 0.0))
    return names, descs, gens


def queries():
    """name -> dict(group=[...], targets=[(agg, arg)], sort=bool); arguments / keys are column names,
    ("cast", col, "double" | "float") or ("add", col, literal)."""
    q = {}
    six = ["x10", "y10", "z10", "x100", "y100", "z100"]
    q["NGA01"] = dict(group=[], targets=[("count", None)] + [("count", c) for c in six])
    for i, agg in enumerate(["sum", "max", "min", "avg"]):
        q[f"NGA0{i + 2}"] = dict(group=[], targets=[(agg, c) for c in six])
    five = lambda c: [("count", c), ("sum", c), ("max", c), ("min", c), ("avg", c)]  # noqa: E731
    for i, k in enumerate(["x10", "x100", "x1k", "x10k", "x100k", "x1m", "x10m"]):
        q[f"PHS00{i + 1}"] = dict(group=[k], targets=[("key", 0)] + five("y10"))
    for i, k in enumerate(["x10", "x100", "x1k", "x10k", "x100k", "x1m"]):
        q[f"PHM00{i + 1}"] = dict(group=[k, "y10"], targets=[("key", 0), ("key", 1)] + five("z10"))
    for i, k in enumerate(["x10", "x100", "x1k", "x10k", "x100k", "x1m"]):
        q[f"BH00{i + 1}"] = dict(group=[("cast", k, "double")], targets=[("key", 0)] + five("y10"))
    for i, k in enumerate(["x10k_s10k", "x100k_s10k", "x1m_s10k"]):
        q[f"BH00{i + 7}"] = dict(group=[k], targets=[("key", 0), ("count", None), ("sum", "y10")])
    q["BH010"] = dict(group=["x10k_s10k", "x100"], targets=[("key", 0), ("key", 1), ("count", None), ("sum", "y10")])
    for i, k in enumerate(["x100k", "x1m", "x10m"]):
        q[f"S00{i + 1}"] = dict(group=[k], targets=[("key", 0), ("count", None)], sort=True)
    ms = lambda a, b: [("count", None), ("max", a), ("max", b), ("max", ("add", b, 1)), ("sum", a), ("sum", ("add", b, 1))]  # noqa: E731
    for i, k in enumerate(["x1k", "x10k", "x100k", "x1m", "x10m"]):
        q[f"MSBS00{i + 1}"] = dict(group=[("cast", k, "float")], targets=[("key", 0)] + ms("x100", "x10"))
    sxy = [("count", None), ("sum", "x10"), ("sum", "y10")]
    q["MSBS006"] = dict(group=["x10k_s10k", "x100"], targets=[("key", 0), ("key", 1)] + sxy)
    q["MSBS007"] = dict(group=["x1m_s10k", "y10"], targets=[("key", 0), ("key", 1)] + sxy)
    q["MSPHM001"] = dict(group=["x100", "y10"], targets=[("key", 0), ("key", 1)] + ms("y100", "x10"))
    for i, k in enumerate(["x1k", "x10k", "x100k", "x1m"]):
        q[f"MSPHM00{i + 2}"] = dict(group=[k, "y10"], targets=[("key", 0), ("key", 1)] + ms("x100", "x10"))
    q["MSPHM006"] = dict(group=["x1k", "x100", "z10"], targets=[("key", 0), ("key", 1), ("key", 2)] + sxy)
    q["MSPHM007"] = dict(group=["x10k", "x100", "z10"], targets=[("key", 0), ("key", 1), ("key", 2), ("count", None),
                                                                 ("avg", "x10"), ("avg", "y10")])
    for i, k in enumerate(["x1k", "x10k", "x100k", "x1m", "x10m"]):
        q[f"MSPHS00{i + 1}"] = dict(group=[k], targets=[("key", 0)] + ms("x100", "x10"))
    q["MSPHS006"] = dict(group=["x1m"], targets=[("key", 0)] + sxy)
    q["MSPHS007"] = dict(group=["x10m"], targets=[("key", 0), ("count", None), ("avg", "x10"), ("avg", "y10")])
    for i, k in enumerate(["x1m", "x10m"]):
        q[f"MSPHS00{i + 8}"] = dict(group=[k], targets=[("key", 0), ("count", None), ("sum", "x100"), ("sum", "x10")])
    for i, k in enumerate(["x100k", "x1m", "x10m"]):
        q[f"MSPHS0{i + 10}"] = dict(group=[k], targets=[("key", 0), ("count", None), ("sum", "x100"), ("max", ("add", "x10", 1))])
    return q


def build_unit(spec, names, descs, n_rows):
    """RelAlgExecutionUnit of one query over the schema; returns (ra, used column names)."""
    from heavydb_amd import capi
    from heavydb_amd.executor import Expr, ExpressionRange, RelAlgExecutionUnit, TargetExpr
    idx = {n: i for i, n in enumerate(names)}
    exprs, used = [], set()

    def col_of(a):
        if isinstance(a, str):
            used.add(a)
            return idx[a]
        kind, c, arg = a
        used.add(c)
        d = descs[idx[c]]
        if kind == "cast":
            t = capi.DOUBLE if arg == "double" else capi.FLOAT
            # getExpressionRange of a cast to floating point: the operand's bounds as doubles
            r = ExpressionRange(True, 0, 0, d.range.has_nulls, float(d.range.min), float(d.range.max))
            e = Expr.col(idx[c]).cast(t).with_range(r)
        else:   # ("add", col, literal): INT + INT
            r = ExpressionRange(True, d.range.min + arg, d.range.max + arg, d.range.has_nulls)
            e = Expr.col(idx[c]).add(Expr.lit(d.type, arg), d.type).with_range(r)
        for k, (a0, _) in enumerate(exprs):
            if a0 == a:
                return len(names) + k
        exprs.append((a, e))
        return len(names) + len(exprs) - 1
    group = [col_of(g) for g in spec["group"]]
    AGG = {"count": capi.COUNT, "sum": capi.SUM, "max": capi.MAX, "min": capi.MIN, "avg": capi.AVG}
    targets = []
    for agg, arg in spec["targets"]:
        if agg == "key":
            targets.append(TargetExpr(capi.PROJECT_KEY, arg))
        elif arg is None:
            targets.append(TargetExpr(capi.COUNT))
        else:
            targets.append(TargetExpr(AGG[agg], col_of(arg)))
    # baseline tables: 2 x an NDV estimate (RelAlgExecutor.cpp:4213-4218): the product of the key cardinalities,
    # bounded by the row count
    ndv = 1
    for g in spec["group"]:
        c = g if isinstance(g, str) else g[1]
        d = descs[idx[c]]
        ndv *= (d.range.max - d.range.min) // (STEP if d.type == capi.INT64 else 1) + 1
    guess = max(2 * min(ndv, n_rows), 16384)
    ra = RelAlgExecutionUnit(list(descs), targets, [], group, max_groups_buffer_entry_guess=guess,
                             exprs=[e for _, e in exprs], num_tuples=n_rows)
    return ra, used


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=4 * FRAGMENT_ROWS)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--only", default="", help="comma-separated name prefixes")
    ap.add_argument("--verify-rows", type=float, default=0, help="also check every query against the oracle on a table "
                    "of this many rows (test infrastructure; 0 = skip)")
    ap.add_argument("--out", default="")
    ap.add_argument("--unprepared", action="store_true", help="time Executor.executeWorkUnit (plan structs rebuilt by the Python mirror every "
                    "step, as the tables of rounds 3 and 4 up to r04_refbench_1b_final did) instead of one mi355q_execute on prepared structs")
    ap.add_argument("--budget-ms", type=float, default=1500.0, help="a query whose step on the first fragment "
                    "extrapolates beyond this at the full size is not run at the full size (the extrapolation is "
                    "reported instead): a shape that still takes the row kernel with a handful of groups serialises "
                    "the device on a few cache lines for minutes")
    ap.add_argument("--flags", type=int, default=0, help="MI355Q_OPT_* bits for every step (1024 = MI355Q_OPT_NO_IDX_PACK: the "
                    "index-partitioned family's plain records instead of the packed word)")
    args = ap.parse_args()
    import torch
    from heavydb_amd import capi
    from heavydb_amd.executor import Executor, FetchResult, generate_column
    from heavydb_amd.multi_gpu import HipShard
    capi.load_library()
    n_rows = int(args.rows)
    names, descs, gens = schema()
    # the table, resident in HBM: 14 columns, 68 B/row
    cols, bufs, rows = [], [], []
    for d in descs:
        cols.append(torch.empty(n_rows, dtype=torch.int32 if d.type == capi.INT32 else torch.int64, device="cuda:0"))
    off = 0
    while off < n_rows:
        n = min(FRAGMENT_ROWS, n_rows - off)
        ptrs = []
        for ci, (t, g) in enumerate(zip(cols, gens)):
            ptr = int(t.data_ptr()) + off * t.element_size()
            generate_column(ptr, n, g[0], g[1], g[2], g[3], g[4], g[5], 0, off, 0)
            ptrs.append(ptr)
        bufs.append(ptrs)
        rows.append(n)
        off += n
    torch.cuda.synchronize()
    fr = FetchResult(bufs, rows, keepalive=cols)
    ex = Executor(0)
    only = [p for p in args.only.split(",") if p]
    out_lines = []
    for name, spec in queries().items():
        if only and not any(name.startswith(p) for p in only):
            continue
        ra, used = build_unit(spec, names, descs, n_rows)
        bpr = sum(4 if dict(INT_COLS).get(c) else 8 for c in used)
        line = {"query": name, "rows": n_rows, "bytes_per_row": bpr}
        try:
            line["route"] = ex.explain(ra, rows)     # mi355q_explain: derived-plan stages + kernel family at this size
        except Exception as e:                       # (never in the way of the measurement)
            line["route"] = "explain failed: " + str(e)
        try:
            # probe on the first fragment: kernel choice and a rate
            fr1 = FetchResult(bufs[:1], rows[:1], keepalive=cols)
            ex.executeWorkUnit(ra, fr1, flags=args.flags)
            torch.cuda.synchronize()
            probe_ms = None
            for _ in range(2):   # (the faster of two: one call of a derived-plan route now and then stalls ~0.2 s in the driver's
                t0 = time.perf_counter()   # allocator — 1.4 GB of twin / result tables per call — and would fake a skip)
                rs1 = ex.executeWorkUnit(ra, fr1, flags=args.flags)
                torch.cuda.synchronize()
                ms1 = (time.perf_counter() - t0) * 1e3
                dev_ms = float(rs1.report.total_ms)     # HIP-event time of the call on its stream: no allocator stall in it
                if dev_ms > 0:
                    ms1 = min(ms1, dev_ms)
                probe_ms = ms1 if probe_ms is None else min(probe_ms, ms1)
            est = probe_ms * n_rows / rows[0]
            if est > args.budget_ms:
                line.update(kernel=rs1.report.kernel_name.decode(), skipped=True, probe_rows=rows[0], probe_ms=round(probe_ms, 3),
                            extrapolated_ms=round(est, 1), whole_step_frac=round(n_rows * bpr / (est * 1e-3) / 8e12, 5))
                print(json.dumps(line), flush=True)
                out_lines.append(line)
                continue
            rs = ex.executeWorkUnit(ra, fr, flags=args.flags)      # warm-up (workspace, retry ladder of the entry guess)
            torch.cuda.synchronize()
            # the plan / input structs of the C-ABI are built once, as bench.py does (the reference compiles a step once too):
            # a step = one mi355q_execute on them + the result storage it writes
            prep = None
            if not args.unprepared:
                try:
                    prep = HipShard.prepare(ex, ra, fr, flags=args.flags)
                    HipShard.execute_prepared(torch, prep)
                except capi.Mi355qError:
                    prep = None              # (a table the retry ladder has to grow: the plain entry point)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                if prep is not None:
                    sh = HipShard.execute_prepared(torch, prep)
                    rs = sh.result_set()
                    rs.report = sh.report
                else:
                    rs = ex.executeWorkUnit(ra, fr, flags=args.flags)
                if spec.get("sort"):
                    outbuf = torch.empty(100 * rs.getQueryMemDesc().row_size // 8, dtype=torch.int64, device="cuda:0")
                    rs.sort(len(spec["targets"]) - 1, 100, int(outbuf.data_ptr()), desc=False)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / args.steps
            line["prepared"] = prep is not None
            line.update(kernel=rs.report.kernel_name.decode(), variant=int(rs.report.variant), ms=round(ms, 3),
                        rows_per_s=n_rows / ms * 1e3, groups=int(rs.rowCount()),
                        whole_step_frac=round(n_rows * bpr / (ms * 1e-3) / 8e12, 4))
        except capi.Mi355qError as e:
            line.update(error=int(e.code), message=str(e))
        print(json.dumps(line), flush=True)
        out_lines.append(line)
    if args.out:
        with open(args.out, "w") as f:
            for ln in out_lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
