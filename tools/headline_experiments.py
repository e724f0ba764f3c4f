#!/usr/bin/env python
"""Headline step (cfg3-filtered) under a list of settings, one table generation for all of them:

    python tools/headline_experiments.py [--rows 1e10] [--steps 3]

Settings = (MI355Q_PAIR_WINDOW, scratch cap in GB); each prints whole-step ms, the scatter's HIP-event
ms per launch, the number of chunks, and — with MI355Q_TRACE=1 for one extra step — phase 2's
per-phase cycle breakdown (stderr).  Used to decide the pacing window and the chunk size; the numbers
quoted in DESIGN.md section 4 / profiles/README.md come from this script.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=float, default=1e10)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--settings", default="1:0,0:0,2:0,4:0,1:48,1:64,0:64,1:8,1:2")
    ap.add_argument("--trace", default="1:0")
    args = ap.parse_args()
    import torch
    from heavydb_amd import capi, synth
    from heavydb_amd.executor import Executor
    from heavydb_amd.multi_gpu import HipShard
    capi.load_library()
    rows = int(args.rows)
    ra, fr, info = synth.cfg3(torch, rows, filtered=True)
    ex = Executor(0)

    def run(window: int, scratch_gb: float, steps: int):
        os.environ["MI355Q_PAIR_WINDOW"] = str(window)
        sb = int(scratch_gb * 2**30)
        sh = HipShard.execute(torch, ex, ra, fr, scratch_bytes=sb)  # warm (allocates the scratch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = []
        for _ in range(steps):
            sh = HipShard.execute(torch, ex, ra, fr, scratch_bytes=sb)
            reps.append(sh.report)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        r = reps[-1]
        return {"pair_window": window, "scratch_gb": scratch_gb, "ms_per_step": round(ms, 3),
                "total_ms_events": round(float(r.total_ms), 3), "chunks": int(r.n_launches),
                "scatter_ms_per_launch": round(float(r.kernel_ms) / max(int(r.n_launches), 1), 3),
                "whole_step_frac": round(rows * 20 / (ms * 1e-3) / 8e12, 4), "spilled": int(r.spilled_rows)}

    for s in args.settings.split(","):
        w, g = s.split(":")
        try:
            print(json.dumps(run(int(w), float(g), args.steps)), flush=True)
        except Exception as e:
            print(json.dumps({"pair_window": w, "scratch_gb": g, "error": repr(e)}), flush=True)
    if args.trace:
        w, g = args.trace.split(":")
        os.environ["MI355Q_TRACE"] = "1"
        print(json.dumps(run(int(w), float(g), 1)), flush=True)
        os.environ["MI355Q_PAIR_WINDOW"] = "0"
        print(json.dumps(run(0, float(g), 1)), flush=True)


if __name__ == "__main__":
    main()
