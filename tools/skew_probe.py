#!/usr/bin/env python
"""How the headline query behaves on skewed keys: a fraction `hot` of the rows gets ONE key, the
rest stays uniform over 10 M keys (1 B rows).  Prints time, the family member that produced the
result and the spill count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from heavydb_amd import synth
from heavydb_amd.executor import Executor

rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
for hot in [float(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "0.001", "0.01", "0.1", "0.3"])]:
    ra, fr, info = synth.cfg3(torch, rows, filtered=True)
    key = fr.keepalive[0]
    if hot < 0:  # Zipf(s = 1) ranks: rank = n_keys ** u, u uniform -> P(rank) ~ 1 / rank
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        step = 1 << 26
        for lo in range(0, key.numel(), step):
            u = torch.rand(min(step, key.numel() - lo), device="cuda", generator=g, dtype=torch.float64)
            rank = torch.exp(u * 16.118095650958319).to(torch.int64).clamp_(0, 9_999_999)  # ln(1e7)
            key[lo:lo + step] = rank * 1000003 + 7
    elif hot > 0:
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        step = 1 << 26
        for lo in range(0, key.numel(), step):
            m = torch.rand(min(step, key.numel() - lo), device="cuda", generator=g) < hot
            key[lo:lo + step][m] = 7 + 1000003 * 12345
    torch.cuda.synchronize()
    ex = Executor(0)
    for rep in range(2):
        t0 = time.perf_counter()
        rs = ex.executeWorkUnit(ra, fr, allow_retry=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"hot fraction {hot}: {dt*1e3:8.2f} ms  kernel {rs.report.kernel_name.decode()} variant {rs.report.variant} "
          f"spilled {rs.report.spilled_rows}  groups {rs.rowCount()}", flush=True)
    del rs, ra, fr, key
    torch.cuda.empty_cache()
