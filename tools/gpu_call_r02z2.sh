#!/bin/bash
out=gpurun_out/r02z2
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "test_hip_matches_oracle" > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log | cut -c1-300
echo finished
