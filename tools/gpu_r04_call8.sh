#!/bin/bash
# round 4, call 8: the keyed payload probe with per-wave record rings (a finished probe sequence is replaced at once)
out=${1:-gpurun_out/r04_call8}
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
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/prof -o sparse -- python $GRAFT_REPO_ROOT/bench.py --config cfg4 --sparse --sum-dim --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof.log 2>&1)
python tools/rocpd_stats.py $out/prof/sparse_results.db > $out/kernel_stats.txt 2>&1 || ls -R $out/prof | head
rm -rf $out/prof
head -12 $out/kernel_stats.txt
