#!/usr/bin/env python
"""cfg1 (COUNT(*) WHERE i32 < k, 100 M rows): where the whole step's time goes — host wall clock per call, the kernel's
HIP-event time, and the library's own trace marks (MI355Q_OPT_TRACE) of one call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heavydb_amd import capi, synth
from heavydb_amd.executor import Executor
from heavydb_amd.multi_gpu import HipShard
ra, fr, info = synth.cfg1(torch, 100_000_000)
ex = Executor(0)
prep = HipShard.prepare(ex, ra, fr)
for _ in range(5):
    HipShard.execute_prepared(torch, prep)
torch.cuda.synchronize()
best = 1e9
kms = 1e9
for _ in range(200):
    t0 = time.perf_counter()
    sh = HipShard.execute_prepared(torch, prep)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
    kms = min(kms, sh.report.kernel_ms)
print(json.dumps({"wall_us": round(best * 1e6, 1), "kernel_us": round(kms * 1e3, 1), "total_ms_report": round(sh.report.total_ms * 1e3, 1)}))
sys.stderr.flush()
prep2 = HipShard.prepare(ex, ra, fr, flags=capi.OPT_TRACE)
for _ in range(3):
    HipShard.execute_prepared(torch, prep2)
