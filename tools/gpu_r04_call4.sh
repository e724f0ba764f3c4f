#!/bin/bash
# round 4, call 4: the index-partitioned family on the device (parity + the reference benchmark at 1 B rows), windows with
# stripe-mates on one XCD, what one rank of 8 spends on its step (kernel split), the full ORDER BY sort timed
out=${1:-gpurun_out/r04_call4}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 900 python -u -m pytest tests/test_zz_gpu_refbench.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x --durations=5 > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -6 $out/pytest.log
timeout 900 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1
echo "refbench exit $?"; python - <<'PY'
import json
for l in open("gpurun_out/r04_call4/refbench_1b.jsonl"):
    d=json.loads(l); print(d.get("query"), (d.get("route") or "")[:60], d.get("ms"), d.get("whole_step_frac"), d.get("skipped",""))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$out/trace_rank -o rank -- python $GRAFT_REPO_ROOT/tools/rank_step_breakdown.py 1e10 8 5 > $GRAFT_REPO_ROOT/$out/rank_step.json 2> $GRAFT_REPO_ROOT/$out/rank_step.err
cd $GRAFT_REPO_ROOT
find $out/trace_rank -name "*kernel_stats.csv" -exec cp {} $out/rank_step_kernel_stats.csv \; ; rm -rf $out/trace_rank
cat $out/rank_step.json; head -12 $out/rank_step_kernel_stats.csv
timeout 300 python tools/topk_time.py > $out/sort_time.txt 2>&1; cat $out/sort_time.txt
