#!/bin/bash
# round 5: ONE script for every GPU call of the round (VERDICT r04 hygiene: no per-call scripts).  usage: tools/gpu_r05.sh <step> [out-dir]
step=${1:-help}
out=${2:-gpurun_out/r05_$step}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
case $step in
  before)   # the "before" of the expression-fusion work: the five BOOLEAN-filter shapes through the k_project interpreter pass
    timeout 500 python tools/bool_filter_bench.py --rows 1e9 --steps 3 > $out/bool_filter_1b.jsonl 2> $out/bool_filter.err; echo "bool_filter exit $?"
    cat $out/bool_filter_1b.jsonl; tail -5 $out/bool_filter.err ;;
  *) echo "unknown step $step"; exit 2 ;;
esac
