#!/bin/bash
out=gpurun_out/r02z
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log | cut -c1-300
timeout 600 python tools/feature_bench.py --rows 1e9 --steps 3 > $out/feature_bench.jsonl 2> $out/feature_bench.err
python - <<PY
import json
for l in open("$out/feature_bench.jsonl"):
    try:
        d=json.loads(l); print(d.get("shape","?")[:100], round(d.get("ms_per_step"),2), d.get("kernel"))
    except Exception: pass
PY
echo finished
