#!/bin/bash
# round 4, call 18: the count-only member of the index-partitioned family on the device (refbench parity at 30 K / 16 M rows,
# the matrix with the large-input members), S001-003 at 1 B rows
out=${1:-gpurun_out/r04_call18}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 600 python -u -m pytest tests/test_zz_gpu_refbench.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "refbench or compact or count or large_input_members" > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -3 $out/pytest.log
timeout 200 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --only S001,S002,S003,PHS005,PHS007 --out $out/refbench_count_only.jsonl > $out/refbench.log 2>&1; echo "refbench exit $?"
python - <<PY
import json
for l in open("$out/refbench_count_only.jsonl"):
    d=json.loads(l); print(d.get("query"), (d.get("route") or "")[:70], d.get("ms"), d.get("whole_step_frac"))
PY
