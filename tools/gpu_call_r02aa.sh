#!/bin/bash
out=gpurun_out/r02aa
mkdir -p $out
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
timeout 1500 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_gpu_slice_merge.py tests/test_zz_gpu_baseline_sizes.py -m gpu -q -p no:cacheprovider -x -k "not cfg4 and not cfg2 and not cfg1" > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -3 $out/pytest_gpu.log | cut -c1-300
timeout 900 python tools/headline_experiments.py --steps 3 --settings "0:0,0:0" --trace "0:0" > $out/experiments.jsonl 2> $out/experiments.err
cat $out/experiments.jsonl; grep "phase 2" $out/experiments.err | tail -2
echo finished
