#!/bin/bash
# round 4, last session: the whole -m gpu suite on the final build (ABI 5: comparisons / CASE, the derived-plan routes, the
# grouped-join gather route), smoke(), the 57 benchmark steps at 1 B rows timed on prepared plan structs, two fresh fuzz seeds
out=${1:-gpurun_out/round_r04_final}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 2400 python -u -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 -rf > $out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $out/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b_prepared.jsonl > $out/refbench_1b.log 2>&1; echo "refbench exit $?"
python - <<PY
import json
for l in open("$out/refbench_1b_prepared.jsonl"):
    d=json.loads(l); print(d.get("query"), d.get("ms"), d.get("whole_step_frac"), d.get("prepared"))
PY
for s in 9301 9302; do
  MI355Q_FUZZ_SEED=$s MI355Q_FUZZ_ITERS=200 timeout 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "fuzzed_row_plans or joins_match_oracle_on_random_plans" > $out/fuzz_seed_$s.log 2>&1
  echo "fuzz seed $s exit $? $(tail -1 $out/fuzz_seed_$s.log | cut -c1-100)"
done
