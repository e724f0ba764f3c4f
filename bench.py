#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native query-step executor.

Metric (BASELINE.json): rows/sec + achieved HBM GB/s of a filtered GROUP BY on 10 B rows at
1/2/4/8 GPUs.  A "step" is one execution of the query step over the synthetic table already
resident in HBM (SURVEY.md section 8(d), cfg3-filtered by default):

    SELECT key, COUNT(*), AVG(f64) FROM t WHERE i32 < 2^30 GROUP BY key
    10 B rows, 10 M int64 keys (baseline hash layout, 20 M entries x 32 B), 20 B/row

  python bench.py                       # N=1, cfg3f at the largest row count that fits
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: fragments are dealt round-robin to the ranks (total rows fixed -> "strong"
scaling), each rank runs the step on its shard and the partial tables are merged over RCCL
(heavydb_amd/multi_gpu.py) inside the timed region.

One JSON line on rank 0: metric/value/unit/... plus
  "roofline":     dominant kernel's algorithmic bytes per launch / its HIP-event duration,
                  against the 8 TB/s HBM3E peak (guide: ~6.3 TB/s is the copy ceiling)
  "cpu_baseline": the oracle (CPU restatement of HeavyDB's CPU executor; kind "port") timed on
                  this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg3f", choices=["cfg1", "cfg2", "cfg3", "cfg3f", "cfg4"])
    ap.add_argument("--rows", type=float, default=0, help="total rows (default: the config's size)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = plan-time choice)")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline time")
    ap.add_argument("--verify", action="store_true", help="size-independent property checks")
    ap.add_argument("--prepartitioned", action="store_true",
                    help="cfg3/cfg3f, N > 1: the table arrives hash-partitioned by key (every key on one "
                         "rank), so the final merge is skipped (SURVEY 8e's second mode)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic.json"),
                    help="per-kernel HBM bytes/launch from a rocprofv3 --pmc pass (optional)")
    return ap.parse_args()


def fit_rows(cfg: str, want_rows: int, world: int, free_bytes: int, bytes_per_row: int) -> int:
    """Largest row count (multiple of the fragment size) whose shard fits this GPU with
    headroom for the output table and the partition scratch."""
    from heavydb_amd.synth import FRAGMENT_ROWS
    budget = int(free_bytes * 0.90) - (24 << 30)  # table, scratch, allocator slack
    per_rank = max(budget // bytes_per_row, FRAGMENT_ROWS)
    total = min(want_rows, per_rank * world)
    if total < want_rows:
        total = max(total // (FRAGMENT_ROWS * world), 1) * FRAGMENT_ROWS * world
    return int(total)


def cpu_baseline(cfg: str, info: dict, target_s: float) -> dict:
    """Time the oracle on a bounded sample of the same workload on the host cores."""
    import numpy as np
    from heavydb_amd import capi
    from heavydb_amd.executor import (ExpressionRange, InputColDescriptor, Qual, RelAlgExecutionUnit,
                                      TargetExpr)
    from oracle import oracle as orc
    threads = os.cpu_count() or 1
    seed0 = 0xC0FFEE00

    def sample(rows_per_frag: int, n_frags: int):
        frags = []
        for f in range(n_frags):
            off = f * rows_per_frag
            if cfg == "cfg1":
                cols = [orc.generate_column(rows_per_frag, capi.GEN_I32_UNIFORM31, seed0, row_offset=off)]
            elif cfg == "cfg2":
                cols = [orc.generate_column(rows_per_frag, capi.GEN_I32_MOD, seed0, 1000, 0, row_offset=off),
                        orc.generate_column(rows_per_frag, capi.GEN_I64_MOD, seed0 + 1, 1000001, -500000, row_offset=off)]
            elif cfg in ("cfg3", "cfg3f"):
                cols = [orc.generate_column(rows_per_frag, capi.GEN_I64_MOD_MUL, seed0, info["n_keys"], 1000003, 7, row_offset=off),
                        orc.generate_column(rows_per_frag, capi.GEN_F64_UNIT, seed0 + 1, a_f=1000.0, row_offset=off)]
                if cfg == "cfg3f":
                    cols.append(orc.generate_column(rows_per_frag, capi.GEN_I32_UNIFORM31, seed0 + 2, row_offset=off))
            else:
                m = info["dim_rows"]
                cols = [orc.generate_column(rows_per_frag, capi.GEN_I64_MOD, seed0, m, 0, row_offset=off),
                        orc.generate_column(rows_per_frag, capi.GEN_I64_MOD, seed0 + 1, 2000001, -1000000, row_offset=off)]
            frags.append(cols)
        return frags

    plan = info["ra"].to_plan()
    join = None
    inner = []
    if cfg == "cfg4":
        m = info["dim_rows"]
        dim_k = np.arange(m, dtype=np.int64)
        dim_w = orc.generate_column(m, capi.GEN_I64_MOD, seed0 + 100, 2001, -1000)
        join = orc.OracleJoin(dim_k, capi.INT64, 0, m - 1)
        inner = [dim_k, dim_w]
        plan.join_table = None

    def run(frags):
        t0 = time.perf_counter()
        q, buf, code = orc.execute(plan, frags, inner, join, n_threads=threads)
        dt = time.perf_counter() - t0
        assert code == 0, code
        return dt

    # Sample shape: the reference dispatches one CPU kernel per fragment, each with a private
    # output buffer (Execute.cpp:3121-3153), then reduces the buffers pairwise.  For the
    # big-table configs a kernel's buffer is the whole 640 MB baseline table, so the sample
    # bounds the number of concurrent kernels (16 threads = 10 GB of tables) and the rows per
    # kernel so that the whole leg stays within ~20-30 s of CPU work.
    if cfg in ("cfg3", "cfg3f"):
        threads = max(1, min(threads, 16))
        # ~1 us per row per thread (cache- and TLB-missing probes of a 640 MB table) plus a
        # sequential pairwise reduce of the per-kernel tables: 4 M rows per kernel ~ 20 s
        rows_per_frag = int(4_000_000 * min(max(target_s, 1.0), 12.0) / 12.0)
    else:
        # calibrate on a small sample, then size the timed sample for ~target_s
        cal_rows = 1_000_000
        dt = run(sample(cal_rows, threads))
        rate = cal_rows * threads / max(dt, 1e-6)
        rows_per_frag = int(min(max(rate * target_s / threads, cal_rows), 48_000_000))
    frags = sample(rows_per_frag, threads)
    dt = run(frags)
    total = rows_per_frag * threads
    return {"value": total / dt, "unit": "rows/s", "cores": threads, "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": f"{total} rows of {cfg} ({threads} fragments x {rows_per_frag} rows, one "
                      f"kernel per fragment per host thread + pairwise reduce), {dt:.2f} s"}


def main():
    args = parse_args()
    import torch
    from heavydb_amd import capi, synth
    from heavydb_amd.executor import Executor
    from heavydb_amd.multi_gpu import HipShard, merge

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run (WORLD_SIZE unset)",
              file=sys.stderr)
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    lib = capi.load_library()  # fails loudly if the HIP extension is missing
    import ctypes as C
    name = C.create_string_buffer(256)
    cus, tot, free, clk, bus = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
    capi.check(lib.mi355q_device_info(local_rank, name, C.byref(cus), C.byref(tot), C.byref(free),
                                      C.byref(clk), C.byref(bus)))

    cfg = args.config
    want = int(args.rows) if args.rows else synth.DEFAULT_ROWS[cfg]
    bpr = {"cfg1": 4, "cfg2": 12, "cfg3": 16, "cfg3f": 20, "cfg4": 16}[cfg]
    total_rows = fit_rows(cfg, want, world, free.value, bpr)
    if world > 1:  # all ranks must agree
        t = torch.tensor([total_rows], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        total_rows = int(t.item())

    prepart = bool(args.prepartitioned and cfg in ("cfg3", "cfg3f") and world > 1)
    extra = {"prepartitioned": True} if prepart else {}
    ra, fr, info = synth.CONFIGS[cfg](torch, total_rows, rank, world, local_rank, **extra)
    info["ra"] = ra
    if cfg == "cfg4":
        info["dim_rows"] = fr.inner_num_rows
    ex = Executor(local_rank)
    local_rows = sum(fr.num_rows)

    def step():
        sh = HipShard.execute(torch, ex, ra, fr, kernel_variant=args.variant,
                              force_generic=args.force_generic)
        rep = sh.report
        if world > 1:
            sh = merge(sh, dist, torch, prepartitioned=prepart)
        return sh, rep

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last = None
    for _ in range(args.warmup):
        last = step()
    sync()
    reports = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
        reports.append(last[1])
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    rows_per_s = total_rows * args.steps / elapsed

    # dominant kernel roofline (rank 0's launches; HIP events on the launch stream)
    k_ms = sum(r.kernel_ms for r in reports)
    k_n = sum(max(r.n_launches, 1) for r in reports)
    avg_ms = k_ms / max(k_n, 1)
    alg_bytes_launch = sum(r.algorithmic_bytes for r in reports) / max(k_n, 1)
    achieved = alg_bytes_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    kname = reports[-1].kernel_name.decode() if reports else ""
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json:
    # FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE, per input row of that kernel)
    traffic = None
    try:
        with open(args.traffic_json) as f:
            tj = json.load(f).get(kname)
        # the committed PMC passes were taken on the default workload (cfg3f) only
        if tj and k_n and cfg == "cfg3f":
            rows_per_launch = sum(r.rows_scanned for r in reports) / k_n
            traffic = tj["hbm_bytes_per_row"] * rows_per_launch
    except Exception:
        pass

    verify = None
    if args.verify and rank == 0:
        rs = last[0].result_set()
        ival, dval, nul = rs.fetch()
        verify = {"groups": int(ival.shape[0])}
        if cfg in ("cfg3", "cfg3f"):
            verify["sum_count"] = int(ival[:, 1].sum())
        if cfg == "cfg4":
            verify["slots"] = [int(x) for x in ival[0]]

    out = {
        "metric": "rows/sec, filtered GROUP BY (key, COUNT(*), AVG(f64)) on 10 B int64-key rows",
        "value": rows_per_s,
        "unit": "rows/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int64" if cfg != "cfg1" else "int32",
        "data": "synthetic (device-generated splitmix64 columns, 32 M-row fragments)",
        "config": {"workload": f"{cfg}: {total_rows} rows" + ("" if total_rows == want else f" (largest that fits; asked {want})"),
                   "bytes_per_row": bpr, "fragments_per_rank": len(fr.num_rows), "rows_per_rank": local_rows,
                   "prepartitioned_by_key": prepart, "kernel": kname, "variant": int(reports[-1].variant) if reports else 0,
                   "device": name.value.decode(), "cus": cus.value,
                   "hbm_reported_gbs": round(2 * clk.value * 1e3 * bus.value / 8 / 1e9, 1)},
        "achieved_gbs_whole_step": total_rows * bpr * args.steps / elapsed / 1e9,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname,
                     "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg_bytes_launch},
    }
    if verify:
        out["verify"] = verify
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, info, args.cpu_seconds)
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"] = {"value": None, "unit": "rows/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {e!r}"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
