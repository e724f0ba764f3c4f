"""Timings of the SURVEY §8(f) rows on one MI355X (multi-column keys, one-to-many / composite /
LEFT joins, encoded columns): rows/s of the whole step and the dominant kernel's name, one JSON
line per shape.  Data is generated on the device with the library's splitmix64 generator.

  python tools/feature_bench.py [--rows 1e9] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    import torch
    from heavydb_amd import capi
    from heavydb_amd.capi import (AVG, COUNT, DOUBLE, GEN_F64_UNIT, GEN_I32_MOD, GEN_I64_MOD, GEN_I64_MOD_MUL, INT32,
                                  INT64, PROJECT_KEY, SUM)
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, HashJoin, InputColDescriptor,
                                      RelAlgExecutionUnit, TargetExpr)
    from heavydb_amd.synth import ColSpec, generate_table, my_fragments
    capi.load_library()
    n = int(args.rows)
    frags = my_fragments(n, 0, 1)
    ex = Executor(0)
    R = ExpressionRange

    def table(specs):
        cols, bufs, rows = generate_table(torch, specs, frags, 0)
        return cols, FetchResult(bufs, rows, keepalive=cols)

    def run(name, ra, fr, bytes_per_row, extra=None):
        if args.only and args.only not in name:
            return
        rs = ex.executeWorkUnit(ra, fr, allow_retry=False)  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rs = ex.executeWorkUnit(ra, fr, allow_retry=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out = {"shape": name, "rows": n, "ms_per_step": dt * 1e3, "rows_per_s": n / dt,
               "algorithmic_gbs": n * bytes_per_row / dt / 1e9, "frac_of_8tbs": n * bytes_per_row / dt / 8e12,
               "kernel": rs.report.kernel_name.decode(), "groups": rs.rowCount()}
        if extra:
            out.update(extra)
        print(json.dumps(out), flush=True)

    # ---- f1: two group-by columns
    # perfect hash: a (0..999) x b (0..99) = 100 K entries; SUM(v)
    specs = [ColSpec(INT32, GEN_I32_MOD, a=1000, range=R(True, 0, 999)),
             ColSpec(INT32, GEN_I32_MOD, a=100, range=R(True, 0, 99)),
             ColSpec(INT64, GEN_I64_MOD, a=1_000_001, b=-500_000, range=R(True, -500_000, 500_000)),
             ColSpec(INT32, GEN_I32_MOD, a=2, range=R(True, 0, 1))]
    cols, fr = table(specs)
    d = [InputColDescriptor(s.type, False, s.range) for s in specs]
    run("f1 multi-col perfect 1000x100, SUM(i64)",
        RelAlgExecutionUnit(d, [TargetExpr(PROJECT_KEY, 0), TargetExpr(PROJECT_KEY, 1), TargetExpr(SUM, 2)],
                            groupby_exprs=[0, 1]), fr, 16)
    run("f1 multi-col perfect 1000x2 (48 KB table: per-workgroup LDS copy), COUNT + SUM",
        RelAlgExecutionUnit(d, [TargetExpr(COUNT), TargetExpr(SUM, 2)], groupby_exprs=[0, 3]), fr, 16)
    del cols, fr
    # baseline hash: (sparse int64 key with 1 M values, int32 0..9) = 10 M groups
    specs = [ColSpec(INT64, GEN_I64_MOD_MUL, a=1_000_000, b=1_000_003, c=7, range=R(True, 7, 999_999 * 1_000_003 + 7)),
             ColSpec(INT32, GEN_I32_MOD, a=10, range=R(True, 0, 9)),
             ColSpec(DOUBLE, GEN_F64_UNIT, a_f=1000.0, range=R(True, 0, 0, False, 0.0, 1000.0))]
    cols, fr = table(specs)
    d = [InputColDescriptor(s.type, False, s.range) for s in specs]
    run("f1 multi-col baseline 10 M groups (int64 x int32), COUNT + AVG(f64)",
        RelAlgExecutionUnit(d, [TargetExpr(PROJECT_KEY, 0), TargetExpr(PROJECT_KEY, 1), TargetExpr(COUNT),
                                TargetExpr(AVG, 2)], groupby_exprs=[0, 1], max_groups_buffer_entry_guess=20_000_000),
        fr, 20)
    # f4: the same sparse key read through kENCODING_FIXED (int32 stored, BIGINT logical) -> generic family
    del cols, fr
    specs = [ColSpec(INT32, GEN_I32_MOD, a=1000, range=R(True, 0, 999)),
             ColSpec(INT32, GEN_I32_MOD, a=1_000_001, b=-500_000, range=R(True, -500_000, 500_000))]
    cols, fr = table(specs)
    d = [InputColDescriptor(INT32, False, specs[0].range, capi.ENC_FIXED, INT64),
         InputColDescriptor(INT32, False, specs[1].range, capi.ENC_FIXED, INT64)]
    run("f4 kENCODING_FIXED(32) BIGINT key and value, perfect 1000 groups, SUM",
        RelAlgExecutionUnit(d, [TargetExpr(PROJECT_KEY), TargetExpr(SUM, 1)], groupby_exprs=[0]), fr, 8)
    del cols, fr

    # ---- f2: joins.  dim: 10 M distinct keys x 2 rows each (one-to-many), fact keys over 12.5 M (80 % match)
    m = 20_000_000
    dk = torch.arange(m, device="cuda", dtype=torch.int64) // 2
    dw = torch.arange(m, device="cuda", dtype=torch.int64) % 1000
    specs = [ColSpec(INT64, GEN_I64_MOD, a=12_500_000, range=R(True, 0, 12_499_999)),
             ColSpec(INT64, GEN_I64_MOD, a=1000, range=R(True, 0, 999))]
    cols, fr = table(specs)
    fr.inner_col_buffers = [int(dk.data_ptr()), int(dw.data_ptr())]
    fr.inner_num_rows = m
    fr.keepalive += [dk, dw]
    d = [InputColDescriptor(s.type, False, s.range) for s in specs]
    inner = [InputColDescriptor(INT64, False, R(True, 0, m // 2 - 1)), InputColDescriptor(INT64, False, R(True, 0, 999))]
    for pb, tag in [(False, "perfect"), (True, "keyed")]:
        t0 = time.perf_counter()
        hj = HashJoin.getInstance(int(dk.data_ptr()), m, INT64, R(True, 0, m // 2 - 1), prefer_baseline=pb,
                                  one_to_many=1)
        build_ms = (time.perf_counter() - t0) * 1e3
        for kind, ktag in [(capi.JOIN_INNER, "INNER"), (capi.JOIN_LEFT, "LEFT")]:
            ra = RelAlgExecutionUnit(d, [TargetExpr(COUNT), TargetExpr(SUM, 1), TargetExpr(SUM, 1, 1)],
                                     inner_col_descs=inner, join_outer_col=0, join_table=hj, join_kind=kind)
            run(f"f2 {ktag} join, one-to-many {tag} table (20 M inner rows, 2 per key), COUNT + SUM(fact) + SUM(dim)",
                ra, fr, 16, {"build_ms_incl_retry": build_ms, "table": hj.info()["hash_type"],
                             "table_bytes": hj.info()["bytes"]})
        hj.free()
    # composite key (int64, int64): dim 10 M distinct pairs, one-to-one
    da = torch.arange(10_000_000, device="cuda", dtype=torch.int64)
    db = da % 7
    hj = HashJoin.getInstance([int(da.data_ptr()), int(db.data_ptr())], 10_000_000, [INT64, INT64], R(),
                              one_to_many=0)
    ra = RelAlgExecutionUnit(d, [TargetExpr(COUNT), TargetExpr(SUM, 1)], inner_col_descs=inner[:1],
                             join_outer_col=[0, 1], join_table=hj)
    fr.inner_col_buffers = [int(da.data_ptr())]
    fr.inner_num_rows = 10_000_000
    fr.keepalive += [da, db]
    run("f2 INNER join on a composite (int64, int64) key, keyed one-to-one table (10 M rows), COUNT + SUM(fact)",
        ra, fr, 16, {"build_ms": hj.info()["build_ms"], "table_bytes": hj.info()["bytes"]})


if __name__ == "__main__":
    main()
