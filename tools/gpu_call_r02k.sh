#!/bin/bash
out=gpurun_out/r02k
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zz_gpu_join_probe.py -m gpu -q -x -p no:cacheprovider > $out/pytest_probe.log 2>&1
echo "pytest probe exit $?"; tail -15 $out/pytest_probe.log | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_zz_gpu_join_probe.py > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log | cut -c1-300
timeout 900 python tools/feature_bench.py --rows 1e9 --steps 3 --only "f2" > $out/feature_bench.jsonl 2> $out/feature_bench.err
python - <<PY
import json
for l in open("$out/feature_bench.jsonl"):
    try:
        d=json.loads(l); print(d.get("shape","?")[:90], round(d.get("ms_per_step"),2), d.get("kernel"))
    except Exception: pass
PY
for extra in "--sparse" "--sparse --sum-dim"; do
  tag=$(echo "cfg4$extra" | tr -d ' ' | tr -- '-' '_')
  timeout 900 python bench.py --config cfg4 $extra --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1])
    print("$tag", d["ms_per_step"], d["roofline"]["whole_step_frac"], d["roofline"]["avg_launch_ms"], d["config"].get("variant"), d.get("verify"))
except Exception as e: print("$tag failed", e, open("$out/bench_$tag.err").read()[-600:])
PY
done
timeout 600 python bench.py --config cfg2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
python - <<PY
import json
d=json.loads(open("$out/bench_cfg2.json").read().strip().splitlines()[-1])
print("cfg2", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
