#!/bin/bash
# round 4, call 7: the keyed payload probe with its four probe sequences advancing together (windows)
out=${1:-gpurun_out/r04_call7}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 600 python -u -m pytest tests/test_zz_gpu_join_probe.py -m gpu -q -p no:cacheprovider -x > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -3 $out/pytest.log
for args in "--sparse" "--sparse --sum-dim"; do
  tag=$(echo "$args" | tr -d ' -' )
  timeout 300 python bench.py --config cfg4 $args --steps 3 --warmup 1 --no-cpu-baseline --verify > $out/bench_cfg4_$tag.json 2> $out/bench_cfg4_$tag.err
  echo "cfg4 $args: exit $? $(python -c "import json,sys; d=json.load(open('$out/bench_cfg4_$tag.json')); print(d['ms_per_step'], d['roofline'].get('whole_step_frac'), d['roofline'].get('avg_launch_ms'))" 2>&1)"
done
