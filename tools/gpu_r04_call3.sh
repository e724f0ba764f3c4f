#!/bin/bash
# round 4, call 3: the reference benchmark at 1 B rows (all 57 steps) after the typed LDS members, the batched probes
# and the vectorised projection; parity of the expression / refbench gpu tests
out=${1:-gpurun_out/r04_call3}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 900 python -u -m pytest tests/test_zz_gpu_refbench.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x --durations=5 > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -6 $out/pytest.log
timeout 900 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_1b.jsonl > $out/refbench_1b.log 2>&1
echo "refbench exit $?"; python - <<'PY'
import json,sys
for l in open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r04_call3/refbench_1b.jsonl"):
    d=json.loads(l); print(d.get("query"), d.get("route"), d.get("ms"), d.get("whole_step_frac"), d.get("skipped",""))
PY
