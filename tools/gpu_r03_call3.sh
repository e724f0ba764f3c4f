#!/bin/bash
# Round 3, third GPU call: the whole -m gpu suite after the ABI / kernel-family changes (minus the two 10 B-row
# cfg4 oracle runs, green in call 1 and untouched since), the native programs, refbench at both sizes, the headline.
out=gpurun_out/r03c
mkdir -p $out
export TMPDIR=/tmp
timeout 1300 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not cfg4_full_size" > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -12 $out/pytest_gpu.log
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
timeout 300 python bench.py --steps 10 --warmup 2 > $out/bench_default.json 2> $out/bench_default.err; echo "bench exit $?"; cut -c1-600 $out/bench_default.json
