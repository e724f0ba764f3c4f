#!/bin/bash
out=gpurun_out/r02n
mkdir -p $out
export TMPDIR=/tmp
cat > /tmp/kexp.py <<'PY'
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch
from heavydb_amd import capi, synth
from heavydb_amd.executor import Executor
from heavydb_amd.multi_gpu import HipShard
capi.load_library()
ex = Executor(0)
def run(tag, ra, fr, **kw):
    sh = HipShard.execute(torch, ex, ra, fr, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        sh = HipShard.execute(torch, ex, ra, fr, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    r = sh.report
    print(json.dumps({"case": tag, "ms_per_step": round(ms, 2), "kernel": r.kernel_name.decode(), "scatter_ms": round(float(r.kernel_ms) / max(int(r.n_launches), 1), 2),
                      "slots": [int(x) for x in sh.result_set().getStorage().reshape(-1)[:2]]}), flush=True)
ra, fr, info = synth.cfg4(torch, 3_200_000_000, sparse=True, sum_dim=True)
for passes in ["1", "2"]:
    for nop in [False, True]:
        os.environ["MI355Q_PROBE_KEYED_R"] = passes
        if nop: os.environ["MI355Q_PROBE_NO_PACING"] = "1"
        else: os.environ.pop("MI355Q_PROBE_NO_PACING", None)
        run(f"sparse passes={passes} pacing={'off' if nop else 'on'}", ra, fr)
del ra, fr
torch.cuda.empty_cache()
os.environ.pop("MI355Q_PROBE_KEYED_R", None)
ra, fr, info = synth.cfg4(torch, 3_200_000_000, sparse=False, sum_dim=True)
for nop in [False, True]:
    if nop: os.environ["MI355Q_PROBE_NO_PACING"] = "1"
    else: os.environ.pop("MI355Q_PROBE_NO_PACING", None)
    run(f"dense sum_dim pacing={'off' if nop else 'on'}", ra, fr)
PY
timeout 900 python /tmp/kexp.py > $out/kexp.jsonl 2> $out/kexp.err; cat $out/kexp.jsonl; tail -2 $out/kexp.err
