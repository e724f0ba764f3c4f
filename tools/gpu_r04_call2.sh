#!/bin/bash
# round 4, call 2: the typed LDS group-by members on the device (parity through both members, the reference benchmark's
# LDS shapes at 1 B rows), and the write cost of 120-byte / 96-byte staging lines (lever (b) of the headline)
out=${1:-gpurun_out/r04_call2}
mkdir -p $out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -c "import torch; x=torch.zeros(4).cuda(); print('device ok', x.sum().item())" || { echo "no GPU"; exit 3; }
timeout 120 ./tools/microbench/scatter 1000000000 short > $out/microbench_scatter_odd_lines.txt 2>&1; tail -8 $out/microbench_scatter_odd_lines.txt
timeout 900 python -u -m pytest tests/test_zz_gpu_refbench.py -m gpu -q -p no:cacheprovider -x --durations=5 > $out/pytest_refbench.log 2>&1
echo "pytest exit $?"; tail -12 $out/pytest_refbench.log
timeout 600 python tools/refbench.py --rows 1e9 --steps 3 --budget-ms 1500 --out $out/refbench_lds_1b.jsonl \
  --only PHS001,PHS002,PHS003,PHS004,PHM001,PHM002,PHM003,BH001,BH002,BH003,BH004,BH007,MSBS001,MSBS002,MSPHS001,MSPHS002,MSPHM001,MSPHM002 > $out/refbench_lds_1b.log 2>&1
echo "refbench exit $?"; cat $out/refbench_lds_1b.jsonl | cut -c1-400
