#!/bin/bash
# round 4, call 10: keyed payload probe, four lanes per probe sequence (one 64-byte line per round): K = 4, 6, 2 sequences per lane group, against the lockstep windows
out=${1:-gpurun_out/r04_call10}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
for sel in 0 3 4 5; do
  for args in "--sparse" "--sparse --sum-dim"; do
    tag=$(echo "$args" | tr -d ' -' )_sel$sel
    timeout 300 python bench.py --config cfg4 $args --blocks-per-cu $sel --steps 2 --warmup 1 --no-cpu-baseline --verify > $out/bench_cfg4_$tag.json 2> $out/bench_cfg4_$tag.err
    echo "sel=$sel cfg4 $args: exit $? $(python -c "import json,sys; d=json.load(open('$out/bench_cfg4_$tag.json')); print(d['ms_per_step'], d['verify'])" 2>&1)"
  done
done
timeout 600 python -u -m pytest tests/test_zz_gpu_join_probe.py -m gpu -q -p no:cacheprovider -x > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -3 $out/pytest.log
