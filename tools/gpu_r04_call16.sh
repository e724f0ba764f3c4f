#!/bin/bash
# round 4, call 16: grouped joins through the gather route on the device (16 M rows vs oracle and vs the row kernel, the
# case matrix with the large-input members), then the same shape at 1 B rows: gather route vs row kernel
out=${1:-gpurun_out/r04_call16}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 1200 python -u -m pytest tests/test_zz_gpu_join_probe.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -s -k "grouped_join or join" > $out/pytest.log 2>&1
echo "pytest exit $?"; grep "grouped join" $out/pytest.log; tail -3 $out/pytest.log
timeout 600 python tools/grouped_join_bench.py --rows 1e9 > $out/grouped_join_1b.jsonl 2> $out/grouped_join_1b.err; echo "bench exit $?"; cat $out/grouped_join_1b.jsonl
