#!/bin/bash
# round 4, call 14: the perfect-twin and shifted-argument routes on the device (refbench parity at 30 K / 16 M rows), the 57 steps at 1 B rows,
# cfg4 sparse on the consolidated keyed probe
out=${1:-gpurun_out/r04_call14}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 900 python -u -m pytest tests/test_zz_gpu_refbench.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x --durations=5 > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -8 $out/pytest.log
timeout 900 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1
echo "refbench exit $?"
python - <<PY
import json
for l in open("$out/refbench_1b.jsonl"):
    d=json.loads(l); print(d.get("query"), (d.get("route") or "")[:60], d.get("ms"), d.get("whole_step_frac"))
PY
for args in "--sparse" "--sparse --sum-dim"; do
  tag=$(echo "$args" | tr -d ' -' )
  timeout 300 python bench.py --config cfg4 $args --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_cfg4_$tag.json 2> $out/bench_cfg4_$tag.err
  echo "cfg4 $args: exit $? $(python -c "import json,sys; d=json.load(open('$out/bench_cfg4_$tag.json')); print(d['ms_per_step'], d['roofline'].get('whole_step_frac'))" 2>&1)"
done
