#!/bin/bash
# round 4, call 15: comparisons of two values / CASE on the device: the parity matrix (new expr_* cases), the reference-run
# vectors through k_project (7 128 comparison vectors), the SQLite cross-check, the binding's device check
out=${1:-gpurun_out/r04_call15}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 1500 python -u -m pytest tests/test_gpu_parity.py tests/test_integration_glue.py tests/test_zz_gpu_refbench.py -m gpu -q -p no:cacheprovider --durations=5 > $out/pytest.log 2>&1
echo "pytest exit $?"; tail -8 $out/pytest.log
timeout 900 python -u -m pytest tests -m gpu -q -p no:cacheprovider -k "expr or sqlite or abi" > $out/pytest_expr.log 2>&1
echo "pytest expr exit $?"; tail -3 $out/pytest_expr.log
