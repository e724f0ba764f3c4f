#!/bin/bash
# first GPU session: parity tests, then short benches of every config/variant
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showmeminfo vram > gpurun_out/smi.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
for spec in "cfg1 0 0" "cfg2 0 0" "cfg3f 1e9 1" "cfg3f 1e9 2" "cfg3f 1e9 3" "cfg3 1e9 2" "cfg4 1e9 0"; do
  set -- $spec
  extra=""
  [ "$2" != "0" ] && extra="--rows $2"
  timeout 300 python bench.py --config $1 $extra --variant $3 --steps 3 --warmup 1 --no-cpu-baseline --verify > gpurun_out/bench_$1_v$3.log 2>&1
  echo "== $spec exit $?"; tail -c 1500 gpurun_out/bench_$1_v$3.log
done
