#!/bin/bash
# The whole -m gpu suite, unbuffered and verbose (a crash leaves the name of the test that was running), then smoke()
out=gpurun_out/full_suite
mkdir -p $out
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "this box cannot initialise the GPU"; exit 3; }
timeout 2400 python -u -m pytest tests -m gpu -v -p no:cacheprovider -x > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -c PASSED $out/pytest_gpu.log; grep -n "FAILED\|Fatal\|fault" $out/pytest_gpu.log | head -5; tail -2 $out/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
