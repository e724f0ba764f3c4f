#!/bin/bash
# round 4, call 5: cfg4 on the keyed table (keys-only probe for Query A; one vs two passes for Query B), the join / parity /
# native multi-rank gpu tests after this round's changes, baseline windows from the NDV estimate, cfg1's fixed cost
out=${1:-gpurun_out/r04_call5}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 900 python -u -m pytest tests/test_zz_gpu_join_probe.py tests/test_gpu_parity.py tests/test_native_multi.py tests/test_zz_gpu_async.py tests/test_integration_glue.py -m gpu -q -p no:cacheprovider -x --durations=5 > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -8 $out/pytest.log
for args in "--sparse" "--sparse --sum-dim" "--sparse --sum-dim --probe-passes 2" "--sparse --probe-passes 2"; do
  tag=$(echo "$args" | tr -d ' -' )
  timeout 300 python bench.py --config cfg4 $args --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_cfg4_$tag.json 2> $out/bench_cfg4_$tag.err
  echo "cfg4 $args: exit $? $(python -c "import json,sys; d=json.load(open('$out/bench_cfg4_$tag.json')); print(d['ms_per_step'], d['roofline'].get('whole_step_frac'), d['config'].get('kernel'))" 2>&1)"
done
timeout 120 python bench.py --config cfg1 --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_cfg1.json 2> $out/bench_cfg1.err
python -c "import json; d=json.load(open('$out/bench_cfg1.json')); print('cfg1', d['ms_per_step'], d['roofline'])"
timeout 300 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_sel.jsonl --only BH004,BH007,MSBS002,S001,S002,S003,BH001,PHS007 > $out/refbench_sel.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r04_call5/refbench_sel.jsonl"):
    d=json.loads(l); print(d.get("query"), (d.get("route") or "")[:70], d.get("ms"), d.get("whole_step_frac"))
PY
