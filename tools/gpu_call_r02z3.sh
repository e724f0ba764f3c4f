#!/bin/bash
out=gpurun_out/r02z3
mkdir -p $out
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
timeout 1500 python -u -m pytest tests -m gpu -v -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -n "PASSED\|FAILED" $out/pytest_gpu.log | tail -3 | cut -c1-300; grep -n "Fatal\|fault" $out/pytest_gpu.log | head -3
timeout 600 python -u tools/feature_bench.py --rows 1e9 --steps 3 > $out/feature_bench.jsonl 2> $out/feature_bench.err
python - <<PY
import json
for l in open("$out/feature_bench.jsonl"):
    try:
        d=json.loads(l); print(d.get("shape","?")[:100], round(d.get("ms_per_step"),2), d.get("kernel"))
    except Exception: pass
PY
tail -3 $out/feature_bench.err
echo finished
