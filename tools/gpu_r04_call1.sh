#!/bin/bash
# round 4, call 1: the two full-size parity tests that were missing, the CU sweep of the headline step and the
# phase-1 / phase-2 overlap sweep (VERDICT r03 next #1, #2c)
out=${1:-gpurun_out/r04_call1}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 900 python tools/headline_experiments.py --steps 4 --settings 0:0 --trace "" --cus 256,224,192,160,128,96 \
  --overlap 224,208,192,176,160,144,128,112,192:100,160:100 > $out/headline_overlap.jsonl 2> $out/headline_overlap.err
echo "experiments exit $?"; cat $out/headline_overlap.jsonl
timeout 1500 python -u -m pytest tests/test_zz_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -x -s \
  -k "unfiltered_full_size or sparse_full_size" --durations=5 > $out/pytest_fullsize.log 2>&1
echo "pytest exit $?"; tail -12 $out/pytest_fullsize.log
