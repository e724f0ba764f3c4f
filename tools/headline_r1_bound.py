#!/usr/bin/env python
"""What the headline step costs when phase 2 reads every record ONCE: cfg3-filtered at 10 B rows with fewer groups, so that a
partition's groups fit one LDS table (R = 1 sub-range; the headline's 10 M groups need R = 2, i.e. every record is read
twice).  Everything lever (d) of the round-4 verdict could gain — the second read of the records served by the L2 instead of
HBM — is bounded by this line: with R = 1 the second read does not exist at all.  One JSON line per key count."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from heavydb_amd import capi, synth  # noqa: E402
from heavydb_amd.executor import Executor  # noqa: E402
from heavydb_amd.multi_gpu import HipShard  # noqa: E402

capi.load_library()
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000_000
for n_keys in (10_000_000, 5_000_000, 4_000_000):
    ra, fr, info = synth.cfg3(torch, rows, filtered=True, n_keys=n_keys)
    ex = Executor(0)
    prep = HipShard.prepare(ex, ra, fr)
    best = None
    for _ in range(4):
        sh = HipShard.execute_prepared(torch, prep)
        torch.cuda.synchronize()
        best = sh.report.total_ms if best is None else min(best, sh.report.total_ms)
    route = ex.explain(ra, fr.num_rows)
    print(json.dumps({"workload": "cfg3f", "rows": rows, "n_keys": n_keys, "entry_count": sh.qmd().entry_count, "ms": round(best, 2),
                      "whole_step_frac": round(20 * rows / (best * 1e-3) / 8e12, 4), "kernel_ms_sum": round(sh.report.kernel_ms, 2),
                      "launches": sh.report.n_launches, "route": route, "groups": sh.result_set().rowCount()}), flush=True)
    del sh, prep, fr, ra
    torch.cuda.empty_cache()
