#!/usr/bin/env python
"""A grouped join at benchmark size, one JSON line per variant:

    SELECT f.g, COUNT(*), SUM(d.w), MAX(d.x), SUM(f.v) FROM f [LEFT] JOIN d ON f.k = d.k GROUP BY f.g

over --rows outer rows (device-generated, 32 M-row fragments; f.k uniform over the 10 M dim keys plus 10 % misses, f.g 100 or
10 000 groups), a one-to-one perfect join table: the planner's route (k_join_gather + the step without a join) against the row
kernel (kernel_variant 1).  Fraction = columns read (k 8 + g 4 + v 8 B/row) / time / 8 TB/s."""
import argparse
import json
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch
    from heavydb_amd import capi
    from heavydb_amd.executor import (Executor, ExpressionRange, FetchResult, HashJoin, InputColDescriptor, RelAlgExecutionUnit,
                                      TargetExpr, generate_column)
    capi.load_library()
    n, m, frag = int(args.rows), 10_000_000, 32_000_000
    rng = np.random.default_rng(1)
    dim_k = torch.from_numpy(rng.permutation(m).astype(np.int64)).cuda()
    dim_w = torch.from_numpy(rng.integers(-1000, 1000, m).astype(np.int64)).cuda()
    dim_x = torch.from_numpy(rng.integers(-5000, 5000, m).astype(np.int32)).cuda()
    hj = HashJoin.getInstance(int(dim_k.data_ptr()), m, capi.INT64, ExpressionRange(True, 0, m - 1))
    fk = torch.empty(n, dtype=torch.int64, device="cuda:0")
    fv = torch.empty(n, dtype=torch.int64, device="cuda:0")
    for groups in (100, 10_000):
        fg = torch.empty(n, dtype=torch.int32, device="cuda:0")
        bufs, rows, off = [], [], 0
        while off < n:
            k = min(frag, n - off)
            generate_column(int(fk.data_ptr()) + off * 8, k, capi.GEN_I64_MOD, 11, m + m // 10, 0, 0, 0.0, 0, off, 0)
            generate_column(int(fg.data_ptr()) + off * 4, k, capi.GEN_I32_MOD, 12, groups, 0, 0, 0.0, 0, off, 0)
            generate_column(int(fv.data_ptr()) + off * 8, k, capi.GEN_I64_MOD, 13, 2_000_001, -1_000_000, 0, 0.0, 0, off, 0)
            bufs.append([int(fk.data_ptr()) + off * 8, int(fg.data_ptr()) + off * 4, int(fv.data_ptr()) + off * 8])
            rows.append(k)
            off += k
        torch.cuda.synchronize()
        fdescs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, m + m // 10 - 1)),
                  InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, groups - 1)),
                  InputColDescriptor(capi.INT64, False, ExpressionRange(True, -10**6, 10**6))]
        idescs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, m - 1)),
                  InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, 999)),
                  InputColDescriptor(capi.INT32, False, ExpressionRange(True, -5000, 4999))]
        fr = FetchResult(bufs, rows, [int(dim_k.data_ptr()), int(dim_w.data_ptr()), int(dim_x.data_ptr())], m,
                         keepalive=[fk, fg, fv, dim_k, dim_w, dim_x])
        ex = Executor(0)
        for left in (False, True):
            ra = RelAlgExecutionUnit(fdescs, [TargetExpr(capi.PROJECT_KEY), TargetExpr(capi.COUNT), TargetExpr(capi.SUM, 1, 1),
                                              TargetExpr(capi.MAX, 2, 1), TargetExpr(capi.SUM, 2)], [], [1], inner_col_descs=idescs,
                                     join_outer_col=0, join_table=hj, join_kind=capi.JOIN_LEFT if left else capi.JOIN_INNER)
            res = {}
            for name, variant in (("planned", 0), ("row_kernel", 1)):
                best, rs = None, None
                for _ in range(args.steps if variant == 0 else 1):
                    rs = ex.executeWorkUnit(ra, fr, allow_retry=False, kernel_variant=variant)
                    best = rs.report.total_ms if best is None else min(best, rs.report.total_ms)
                res[name] = (best, rs)
            a, b = res["planned"][1].getStorage(), res["row_kernel"][1].getStorage()
            line = {"query": f"grouped join, {groups} groups, {'LEFT' if left else 'INNER'}", "rows": n, "bytes_per_row": 20,
                    "route": ex.explain(ra, rows), "kernel": res["planned"][1].report.kernel_name.decode(),
                    "ms": round(res["planned"][0], 3), "row_kernel_ms": round(res["row_kernel"][0], 3),
                    "whole_step_frac": round(20 * n / (res["planned"][0] * 1e-3) / 8e12, 4),
                    "same_as_row_kernel": bool(np.array_equal(a, b))}
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
