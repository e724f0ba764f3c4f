#!/bin/bash
# Round 3, fourth GPU call: the widened families (LDS baseline attempts, DOUBLE / 4-byte-width keys through the
# partitioned family, several value columns zipped) against the oracle, then the refbench tables.
out=gpurun_out/r03d
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_refbench.py tests/test_gpu_parity.py tests/test_zz_gpu_sqlite_scale.py tests/test_zz_gpu_boundary.py -m gpu -q -x -p no:cacheprovider > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -8 $out/pytest.log
timeout 400 python tools/refbench.py --steps 3 --out $out/refbench_128m.jsonl > $out/refbench_128m.log 2>&1
echo "refbench 128M exit $?"
timeout 600 python tools/refbench.py --rows 1e9 --steps 3 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1
echo "refbench 1B exit $?"
python - <<PY
import json
for tag in ("128m", "1b"):
    print("==", tag)
    try:
        for l in open("$out/refbench_%s.jsonl" % tag):
            d = json.loads(l)
            print(d["query"], d.get("kernel"), d.get("ms", d.get("extrapolated_ms")), d.get("whole_step_frac"), "SKIPPED" if d.get("skipped") else "", d.get("error", ""))
    except Exception as e:
        print("no table:", e)
PY
