#!/usr/bin/env python
"""NonGroupedAgg shapes (six nullable INT32 columns, one aggregate kind) at --rows rows over workgroups-per-CU of the
scan-aggregate family: one JSON line per (kind, blocks per CU)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e9)
    ap.add_argument("--bpc", default="1,2,3,4,6")
    args = ap.parse_args()
    import torch
    from heavydb_amd import capi
    from heavydb_amd.executor import Executor, ExpressionRange, FetchResult, InputColDescriptor, RelAlgExecutionUnit, TargetExpr, generate_column
    n, frag = int(args.rows), 32_000_000
    cols = [torch.empty(n, dtype=torch.int32, device="cuda:0") for _ in range(6)]
    bufs, rows, off = [], [], 0
    while off < n:
        k = min(frag, n - off)
        for i, t in enumerate(cols):
            generate_column(int(t.data_ptr()) + off * 4, k, capi.GEN_I32_MOD, 40 + i, 10 if i < 3 else 100, 0, 0, 0.0, 0, off, 0)
        bufs.append([int(t.data_ptr()) + off * 4 for t in cols])
        rows.append(k)
        off += k
    torch.cuda.synchronize()
    descs = [InputColDescriptor(capi.INT32, True, ExpressionRange(True, 0, 99, False)) for _ in range(6)]
    fr = FetchResult(bufs, rows, keepalive=cols)
    ex = Executor(0)
    for kind, agg in (("count", capi.COUNT), ("sum", capi.SUM), ("max", capi.MAX), ("avg", capi.AVG)):
        ra = RelAlgExecutionUnit(descs, [TargetExpr(agg, c) for c in range(6)], num_tuples=n)
        for bpc in [int(x) for x in args.bpc.split(",")]:
            best = None
            for _ in range(4):
                rs = ex.executeWorkUnit(ra, fr, allow_retry=False, tune_blocks_per_cu=bpc)
                best = rs.report.kernel_ms if best is None else min(best, rs.report.kernel_ms)
            print(json.dumps({"kind": kind, "blocks_per_cu": bpc, "kernel_ms": round(best, 3), "variant": rs.report.variant,
                              "frac": round(24 * n / (best * 1e-3) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    main()
