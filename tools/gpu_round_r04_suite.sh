#!/bin/bash
# round 4, final session part 1: the whole -m gpu suite on the final build (incl. the 10 B-row oracle tests), then smoke()
out=${1:-gpurun_out/round_r04}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 2700 python -u -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 -rf > $out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $out/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
