#!/bin/bash
out=gpurun_out/r02m
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
ra, fr, info = synth.cfg4(torch, 3_200_000_000, sparse=True, sum_dim=True)
ex = Executor(0)
for var in ["1", "2", "4", "direct"]:
    kw = {}
    if var == "direct":
        kw["kernel_variant"] = 1
    else:
        os.environ["MI355Q_PROBE_KEYED_R"] = var
    sh = HipShard.execute(torch, ex, ra, fr, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        sh = HipShard.execute(torch, ex, ra, fr, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    r = sh.report
    print(json.dumps({"passes": var, "ms_per_step": round(ms, 2), "kernel": r.kernel_name.decode(), "scatter_ms": round(float(r.kernel_ms) / max(int(r.n_launches), 1), 2),
                      "slots": [int(x) for x in sh.result_set().getStorage().reshape(-1)[:2]]}), flush=True)
PY
timeout 900 python /tmp/kexp.py > $out/kexp.jsonl 2> $out/kexp.err; cat $out/kexp.jsonl; tail -2 $out/kexp.err
export MI355Q_PROBE_KEYED_R=2
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  rocprofv3 --kernel-trace --pmc $grp -d $out/pmc -o pmc -- python bench.py --config cfg4 --sparse --rows 3.2e9 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc.log 2>&1
  python tools/rocpd_stats.py $out/pmc/pmc_results.db | sed -n '/PMC/,$p' | grep -E "k_part_probe" | cut -c1-200
  rm -rf $out/pmc
done
