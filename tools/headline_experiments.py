#!/usr/bin/env python
"""Headline step (cfg3-filtered) under a list of settings, one table generation for all of them:

    python tools/headline_experiments.py [--rows 1e10] [--steps 3]

Settings = (exec-option flags, scratch cap in GB); each prints whole-step ms, the scatter's HIP-event ms per
launch, the number of chunks, and — with MI355Q_OPT_TRACE for one extra step — phase 2's per-phase cycle breakdown
(stderr).  Used to decide the chunk size and whether the pair rendezvous pays; the numbers quoted in DESIGN.md
section 4 / profiles/README.md come from this script.  (Round 2 also swept the per-wave pair pacing window here; it
was measured slower at every window and has been removed from the kernel, profiles/r02_phase2_pair_pacing_*.)
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
    ap.add_argument("--settings", default="0:0,2:0,0:48,0:32,0:8", help="flags:scratch_gb, ... (flags 2 = no pair rendezvous)")
    ap.add_argument("--trace", default="0:0")
    ap.add_argument("--cus", default="", help="comma-separated CU counts to plan / launch with (tune_cus): how phase 1 "
                    "and phase 2 scale, i.e. whether they could run side by side on disjoint CU sets")
    ap.add_argument("--overlap", default="", help="comma-separated tune_overlap_cus values (phase 1 of chunk i + 1 on that many "
                    "CUs next to phase 2 of chunk i on the rest), optionally cus:scratch_gb")
    args = ap.parse_args()
    import torch
    from heavydb_amd import capi, synth
    from heavydb_amd.executor import Executor
    from heavydb_amd.multi_gpu import HipShard
    capi.load_library()
    rows = int(args.rows)
    ra, fr, info = synth.cfg3(torch, rows, filtered=True)
    ex = Executor(0)

    def run(flags: int, scratch_gb: float, steps: int, cus: int = 0, overlap: int = 0):
        sb = int(scratch_gb * 2**30)
        sh = HipShard.execute(torch, ex, ra, fr, scratch_bytes=sb, flags=flags, tune_cus=cus, tune_overlap_cus=overlap)  # warm (allocates the scratch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = []
        for _ in range(steps):
            sh = HipShard.execute(torch, ex, ra, fr, scratch_bytes=sb, flags=flags, tune_cus=cus, tune_overlap_cus=overlap)
            reps.append(sh.report)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        r = reps[-1]
        return {"flags": flags, "cus": cus, "overlap_cus": overlap, "scratch_gb": scratch_gb, "ms_per_step": round(ms, 3),
                "total_ms_events": round(float(r.total_ms), 3), "chunks": int(r.n_launches),
                "scatter_ms_per_launch": round(float(r.kernel_ms) / max(int(r.n_launches), 1), 3),
                "whole_step_frac": round(rows * 20 / (ms * 1e-3) / 8e12, 4), "spilled": int(r.spilled_rows)}

    for s in args.settings.split(","):
        w, g = s.split(":")
        try:
            print(json.dumps(run(int(w), float(g), args.steps)), flush=True)
        except Exception as e:
            print(json.dumps({"flags": w, "scratch_gb": g, "error": repr(e)}), flush=True)
    for c in [int(x) for x in args.cus.split(",") if x]:
        print(json.dumps(run(0, 0.0, args.steps, c)), flush=True)
    for x in [x for x in args.overlap.split(",") if x]:
        c, _, g = x.partition(":")
        try:
            print(json.dumps(run(0, float(g or 0), args.steps, 0, int(c))), flush=True)
        except Exception as e:
            print(json.dumps({"overlap_cus": c, "scratch_gb": g, "error": repr(e)}), flush=True)
    if args.trace:
        w, g = args.trace.split(":")
        print(json.dumps(run(int(w) | capi.OPT_TRACE, float(g), 1)), flush=True)


if __name__ == "__main__":
    main()
