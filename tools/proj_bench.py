#!/usr/bin/env python
"""The PROJECTION family at benchmark size — one JSON line per shape:

    SELECT v0, ..., v{k-1} FROM t WHERE i32 < K        (i32 uniform in [0, 2^31); v: INT64 / DOUBLE alternating)

over --rows rows (device-generated, 32 M-row fragments) for selectivity 1 % / 50 % / 99 % and 1 / 3 / 6 projected columns,
row-wise (and the 3-column shape columnar).  Algorithmic bytes of a step = the columns the plan reads (4 + 8 k bytes per
input row) + the entries it writes ((8 + 8 k) bytes per MATCHING row: the row offset and the values); `frac` = those
bytes / the step's time / 8 TB/s, `kernel_frac` the same over the compaction kernel's own HIP-event time.  The output
buffer is caller-owned and sized like the reference sizes it after its COUNT(*) pre-flight (matches + 1 %)."""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--sel", default="0.01,0.5,0.99")
    ap.add_argument("--cols", default="1,3,6")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--variant", default="plain", choices=["plain", "general", "expr", "exprfilter", "join", "join1n"],
                    help="general: the same shapes through the general member (pass_rows = -1); expr: the second target is v1 * 2.0 "
                         "and a third one v0 + 1 (expressions in registers); join: SELECT v.., d.w FROM t JOIN d ON t.fk = d.k "
                         "(d: 1 M rows, dense keys; fk = i32 >> 11)")
    ap.add_argument("--route", default="auto", choices=["auto", "fused", "split"], help="the fast member as ONE launch (ticket + look-back; pass_rows -2) or as k_proj_mask + k_proj_scan_tiles + pass B (pass_rows -3); auto: split from 4 096 tiles")
    ap.add_argument("--interpreted", action="store_true", help="MI355Q_OPT_NO_COMPILED_FILTER: an expression filter through the general member's interpreter instead of the row-mask pre-pass")
    ap.add_argument("--generic-member", action="store_true", help="MI355Q_OPT_LDS_GENERIC_MEMBER: expression targets through the general member's interpreter even where they are forms of the fast member")
    args = ap.parse_args()
    import torch
    from heavydb_amd import capi, synth
    from heavydb_amd.executor import Executor
    lib = capi.load_library()
    n = int(args.rows)
    cache = {}
    ex = Executor(0)
    synth.projection(torch, n, 6, 0.5, cols_cache=cache)
    for sel in [float(x) for x in args.sel.split(",")]:
        for n_out in [int(x) for x in args.cols.split(",")]:
            for columnar in ([False, True] if n_out == 3 else [False]):
                ra, fr, info = synth.projection(torch, n, n_out, sel, columnar=columnar, cols_cache=cache)
                opts = {"flags": (capi.OPT_LDS_GENERIC_MEMBER if args.generic_member else 0) | (capi.OPT_NO_COMPILED_FILTER if args.interpreted else 0)}
                if args.route != "auto":
                    opts["pass_rows"] = {"fused": -2, "split": -3}[args.route]
                if args.variant == "general":
                    opts["pass_rows"] = -1
                elif args.variant == "expr":
                    from heavydb_amd.executor import Expr, ExpressionRange, TargetExpr
                    nc = len(ra.input_col_descs)
                    xs = [Expr.col(1).add(Expr.lit(capi.INT64, 1), capi.INT64).with_range(ExpressionRange(True, -500_000_000, 500_000_007))]
                    tg = [TargetExpr(capi.PROJECT, nc)]
                    if n_out >= 2:
                        xs.append(Expr.col(2).mul(Expr.lit(capi.DOUBLE, 2.0), capi.DOUBLE).with_range(ExpressionRange()))
                        tg.append(TargetExpr(capi.PROJECT, nc + 1))
                    tg += [TargetExpr(capi.PROJECT, 1 + i) for i in range(2, n_out)]
                    ra.exprs, ra.target_exprs = xs, tg
                elif args.variant == "exprfilter":   # WHERE i32 + 5 < k + 5: the same rows through an expression (a lean program atom)
                    from heavydb_amd.executor import Expr, ExpressionRange, Qual
                    nc = len(ra.input_col_descs)
                    k_lit = int(ra.simple_quals[0].literal)
                    ra.exprs = [Expr.col(0).cast(capi.INT64).add(Expr.lit(capi.INT64, 5), capi.INT64).cmp(capi.EX_LT, Expr.lit(capi.INT64, k_lit + 5))
                                .with_range(ExpressionRange(True, 0, 1, False))]
                    ra.simple_quals = [Qual(nc, capi.EQ, 1)]
                elif args.variant in ("join", "join1n"):
                    from heavydb_amd.executor import ExpressionRange, FetchResult, HashJoin, InputColDescriptor, TargetExpr
                    m = 1 << 20
                    one_n = args.variant == "join1n"   # every key of the dimension twice: two entries per matching outer row
                    if "join" not in cache:
                        dk = torch.randperm(m, device="cuda").to(torch.int64)
                        if one_n:
                            dk = dk % (m // 2)
                        dw = torch.randint(-1000, 1000, (m,), device="cuda", dtype=torch.int64)
                        fks = [(c.view(torch.int32) >> 11).contiguous() for c in [info["cols"][0]]]
                        cache["join"] = (dk, dw, fks, HashJoin.getInstance(int(dk.data_ptr()), m, capi.INT64, ExpressionRange(True, 0, (m // 2 if one_n else m) - 1),
                                                                           one_to_many=2 if one_n else 0))
                    dk, dw, fks, hj = cache["join"]
                    nc = len(ra.input_col_descs)
                    off, bufs = 0, []
                    for b, r in zip(fr.col_buffers, fr.num_rows):
                        bufs.append(list(b) + [int(fks[0].data_ptr()) + off * 4])
                        off += r
                    ra.input_col_descs = list(ra.input_col_descs) + [InputColDescriptor(capi.INT32, False, ExpressionRange(True, 0, m - 1))]
                    ra.inner_col_descs = [InputColDescriptor(capi.INT64, False, ExpressionRange(True, 0, m - 1)),
                                          InputColDescriptor(capi.INT64, False, ExpressionRange(True, -1000, 999))]
                    ra.join_outer_col, ra.join_table = nc, hj
                    ra.target_exprs = list(ra.target_exprs) + [TargetExpr(capi.PROJECT, 1, 1)]
                    fr = FetchResult(bufs, fr.num_rows, [int(dk.data_ptr()), int(dw.data_ptr())], m, keepalive=[fr, dk, dw, fks])
                    info = dict(info, bytes_per_row=info["bytes_per_row"] + 4, out_bytes_per_row=info["out_bytes_per_row"] + 8)
                    if one_n:
                        ra.max_groups_buffer_entry_guess = 2 * ra.max_groups_buffer_entry_guess
                q = capi.QMD()
                assert lib.mi355q_qmd_init(ctypes.byref(ra.to_plan()), ctypes.byref(q)) == 0
                out = torch.empty(lib.mi355q_qmd_buffer_bytes(ctypes.byref(q)) // 8, dtype=torch.int64, device="cuda")
                best = kbest = None
                for _ in range(args.steps + 1):  # (the first step allocates the family's workspace)
                    rs = ex.executeWorkUnit(ra, fr, allow_retry=False, out_buffer=int(out.data_ptr()), tune_blocks_per_cu=args.blocks_per_cu, **opts)
                    if best is not None or args.steps == 0:
                        best = rs.report.total_ms if best is None else min(best, rs.report.total_ms)
                        kbest = rs.report.kernel_ms if kbest is None else min(kbest, rs.report.kernel_ms)
                    else:
                        best, kbest = float("inf"), float("inf")
                matched = rs.totalMatched()
                bytes_ = info["bytes_per_row"] * n + info["out_bytes_per_row"] * matched
                # what a step must MOVE: the filter column whole, the projected columns' matching rows only, the entries written
                # (the projected columns are read lazily: at 1 % the algorithmic figure counts bytes no kernel reads)
                must = 4 * n + (info["bytes_per_row"] - 4) * matched + info["out_bytes_per_row"] * matched
                print(json.dumps({"shape": f"sel{sel}_cols{n_out}_{'columnar' if columnar else 'rowwise'}", "variant": args.variant, "route": "split" if rs.report.variant == 16 else "fused" if rs.report.variant == 0 else "general", "rows": n, "matched": matched,
                                  "kernel": rs.report.kernel_name.decode(), "ms": round(best, 3), "kernel_ms": round(kbest, 3),
                                  "read_bytes_per_row": info["bytes_per_row"], "written_bytes_per_match": info["out_bytes_per_row"],
                                  "algorithmic_gb": round(bytes_ / 1e9, 3), "gbps": round(bytes_ / (best * 1e-3) / 1e9, 1),
                                  "frac": round(bytes_ / (best * 1e-3) / 8e12, 4), "kernel_frac": round(bytes_ / (kbest * 1e-3) / 8e12, 4),
                                  "must_move_gb": round(must / 1e9, 3), "must_move_frac": round(must / (best * 1e-3) / 8e12, 4),
                                  "rows_per_s": round(n / (best * 1e-3), 0)}), flush=True)
                del out


if __name__ == "__main__":
    main()
