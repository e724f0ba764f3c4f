#!/bin/bash
# Round 3, second GPU call: projected expressions on the device (matrix cases, the reference-function vectors),
# the reference's synthetic benchmark queries vs the oracle, Arrow vs oracle, and a first refbench table
# (before the fast families are widened).
out=gpurun_out/r03b
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_arrow.py tests/test_zz_gpu_refbench.py -m gpu -q -x -p no:cacheprovider \
   -k "expr or arrow or refbench" > $out/pytest_expr.log 2>&1
echo "pytest exit $?"; tail -15 $out/pytest_expr.log
timeout 600 python tools/refbench.py --steps 3 --out $out/refbench_128m_before.jsonl > $out/refbench_128m.log 2>&1
echo "refbench 128M exit $?"; python - <<PY
import json
for l in open("$out/refbench_128m_before.jsonl"):
    d = json.loads(l)
    print(d["query"], d.get("kernel"), d.get("ms"), d.get("whole_step_frac"), d.get("error", ""))
PY
timeout 900 python tools/refbench.py --rows 1e9 --steps 3 --out $out/refbench_1b_before.jsonl > $out/refbench_1b.log 2>&1
echo "refbench 1B exit $?"; python - <<PY
import json
for l in open("$out/refbench_1b_before.jsonl"):
    d = json.loads(l)
    print(d["query"], d.get("kernel"), d.get("ms"), d.get("whole_step_frac"), d.get("error", ""))
PY
