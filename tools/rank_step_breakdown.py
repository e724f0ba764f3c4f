#!/usr/bin/env python
"""What ONE rank of 8 spends on its own step of the headline query (cfg5: 1.25 B of the 10 B rows, fragment f -> rank f % 8):
wall time per step, the HIP-event time of the whole call and of the scatter launches, so that the part of the step
that does not shrink with the rows (the 10 M-group table, launch tails) can be read off against the 1-GPU step.  Run it
under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from heavydb_amd import capi, synth
    from heavydb_amd.executor import Executor
    from heavydb_amd.multi_gpu import HipShard
    capi.load_library()
    total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000_000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    ex = Executor(0)
    ra, fr, info = synth.cfg3(torch, total, 0, world, 0, filtered=True)
    rows = sum(fr.num_rows)
    HipShard.execute(torch, ex, ra, fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = []
    for _ in range(steps):
        sh = HipShard.execute(torch, ex, ra, fr)
        reps.append(sh.report)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    r = reps[-1]
    print(json.dumps({"world": world, "rows_of_this_rank": rows, "step_ms": round(ms, 3), "events_total_ms": round(float(r.total_ms), 3),
                      "scatter_ms": round(float(r.kernel_ms), 3), "chunks": int(r.n_launches),
                      "proportional_share_of_1gpu_step_ms": None}))


if __name__ == "__main__":
    main()
