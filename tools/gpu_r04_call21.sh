#!/bin/bash
# round 4, call 21: the lattice-key route on the device: refbench parity (30 K / 16 M rows), the matrix with the large-input
# members, BH007-010 / MSBS006-007 (+ the headline shape, which must not take it) at 1 B rows
out=${1:-gpurun_out/r04_call21}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 400 python -u -m pytest tests/test_zz_gpu_refbench.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "refbench or large_input_members or random_plans or fuzzed" > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -3 $out/pytest.log
timeout 200 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --only BH007,BH008,BH009,BH010,MSBS006,MSBS007 --out $out/refbench_lattice.jsonl > $out/refbench.log 2>&1; echo "refbench exit $?"
python - <<PY
import json
for l in open("$out/refbench_lattice.jsonl"):
    d=json.loads(l); print(d.get("query"), d.get("kernel"), d.get("ms"), d.get("whole_step_frac"))
PY
timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_default.json 2>/dev/null; python -c "import json; d=json.load(open('$out/bench_default.json')); print('headline', d['ms_per_step'], d['config']['kernel'])"
